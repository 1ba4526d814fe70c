// overlap_probe.hip - does an HBM-streaming kernel on one stream overlap an f64-VALU kernel on another?
// (design question behind the E-step / M-step overlap of the lock-step EM; not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_probe tools/overlap_probe.hip && /tmp/overlap_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                          \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

// streaming read: every workgroup sums a contiguous slab, 16 B per lane per load, UNROLL loads in flight.
// REGS pads the register allocation (array kept live) to mimic a register-hungry streaming kernel.
template <int UNROLL, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_stream(const double2 *__restrict__ src, size_t n_per_wg,
                                                               double *__restrict__ out) {
    const double2 *p = src + (size_t)blockIdx.x * n_per_wg;
    double acc = 0.0;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < n_per_wg; i += UNROLL * 256) {
        double2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

// f64 VALU work: dependent exp chains, one wavefront per block like k2_estep
__global__ __launch_bounds__(64, 3) void k_valu(double *__restrict__ out, int iters) {
    extern __shared__ double pad_lds[];   // dynamic LDS only caps how many of these blocks share a CU
    if (iters < 0) pad_lds[threadIdx.x] = 1.0;
    double x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = -1.0 - 0.001 * (threadIdx.x + j);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = -exp(x[j]) - 0.5;
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    if (s == 12345.678) out[blockIdx.x] = s;
}

static float time_ms(hipStream_t s, hipEvent_t a, hipEvent_t b) {
    float ms;
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    const size_t bytes = (size_t)12 << 30;
    double2 *src;
    double *out;
    CHK(hipMalloc(&src, bytes));
    CHK(hipMalloc(&out, 1 << 24));
    CHK(hipMemset(src, 0, bytes));
    hipStream_t s1, s2;
    CHK(hipStreamCreate(&s1));
    CHK(hipStreamCreate(&s2));
    hipEvent_t a1, b1, a2, b2;
    CHK(hipEventCreate(&a1));
    CHK(hipEventCreate(&b1));
    CHK(hipEventCreate(&a2));
    CHK(hipEventCreate(&b2));
    const int n_wg = 32768;
    const size_t n_per_wg = bytes / 16 / n_wg;
    const int valu_blocks = 51200, iters = 150;
    size_t valu_lds = 13 * 1024;
    auto run = [&](int variant, bool do_s, bool do_v) {
        if (do_s) {
            CHK(hipEventRecord(a1, s1));
            if (variant == 0) hipLaunchKernelGGL((k_stream<4, 8>), dim3(n_wg), dim3(256), 0, s1, src, n_per_wg, out);
            if (variant == 1) hipLaunchKernelGGL((k_stream<16, 3>), dim3(n_wg), dim3(256), 0, s1, src, n_per_wg, out);
            if (variant == 2) hipLaunchKernelGGL((k_stream<8, 4>), dim3(n_wg), dim3(256), 0, s1, src, n_per_wg, out);
            CHK(hipEventRecord(b1, s1));
        }
        if (do_v) {
            CHK(hipEventRecord(a2, s2));
            hipLaunchKernelGGL(k_valu, dim3(valu_blocks), dim3(64), valu_lds, s2, out, iters);
            CHK(hipEventRecord(b2, s2));
        }
        CHK(hipDeviceSynchronize());
        float ts = do_s ? time_ms(s1, a1, b1) : 0, tv = do_v ? time_ms(s2, a2, b2) : 0;
        return std::pair<float, float>(ts, tv);
    };
    CHK(hipFuncSetAttribute((const void *)k_valu, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (int lds_kb : {13, 20, 40})
    for (int variant = 0; variant < 3; ++variant) {
        valu_lds = (size_t)lds_kb * 1024;
        run(variant, true, true);
        auto s = run(variant, true, false);
        auto v = run(variant, false, true);
        hipEvent_t w0, w1;
        CHK(hipEventCreate(&w0));
        CHK(hipEventCreate(&w1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(w0, s1));
        CHK(hipStreamWaitEvent(s2, w0, 0));
        auto both = run(variant, true, true);
        printf("valu LDS %2d KB/wave, variant %d: stream alone %.2f ms (%.2f TB/s), valu alone %.2f ms, together: stream %.2f ms, valu %.2f ms\n",
               lds_kb, variant, s.first, bytes / s.first / 1e9, v.second, both.first, both.second);
    }
    return 0;
}
