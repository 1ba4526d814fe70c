// f64_rate_probe.hip - what the f64 pipes of a gfx950 SIMD deliver (tools only, not part of the product):
//   v_mfma_f64_16x16x4_f64 in one dependent chain / in four independent chains, v_fma_f64 in eight independent chains,
//   and both kinds side by side from different wavefronts of a SIMD (do they overlap?).
//   hipcc --offload-arch=gfx950 -O3 -o build/f64_rate_probe tools/f64_rate_probe.hip && build/f64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 4: as 1, but every MFMA of the loop body reads A / B register pairs of its own (16 of each), as a kernel whose operands come fresh from LDS
// MODE 0: one MFMA chain; 1: four MFMA chains; 2: eight FMA chains; 3: blocks of 8 wavefronts - 0-3 MFMA (4 chains), 4-7 FMA
// (wavefronts w and w + 4 of a workgroup share a SIMD)
template <int MODE>
__global__ __launch_bounds__(512) void k_rate(double *out, int iters, double x, double y) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE <= 1 || MODE == 4 || (MODE == 3 && wave < 4);
    double res = 0.0;
    if (do_mfma) {
        v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        if (MODE == 4) {
            double xa[16], yb[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                xa[q] = x + 1e-9 * (q + threadIdx.x);
                yb[q] = y - 1e-9 * (q + threadIdx.x);
                asm volatile("" : "+v"(xa[q]), "+v"(yb[q]));
            }
            for (int i = 0; i < iters; i += 4) {
#pragma unroll
                for (int q = 0; q < 16; q += 4) {
                    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[q], yb[q], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[q + 1], yb[q + 1], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[q + 2], yb[q + 2], a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[q + 3], yb[q + 3], a3, 0, 0, 0);
                }
            }
        } else
        for (int i = 0; i < iters; ++i) {
            if (MODE == 0) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
        }
        res = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) c[q] = __builtin_fma(x, y, c[q]);      // 32 v_fma_f64 per iteration
        for (int q = 0; q < 8; ++q) res += c[q];
    }
    if (res == 1234.5) out[blockIdx.x] = res;
}

template <int MODE>
static void run(const char *name, int waves_per_simd, double *out) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const int iters = 20000, threads = MODE == 3 ? 512 : 256;
    const int blocks = 256 * waves_per_simd / (threads / 256);    // 4 wavefronts per 256 threads = one per SIMD of a CU
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 0.9999999);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double waves = (double)blocks * (threads / 64), simds = 1024.0;
    // per SIMD: instructions issued by its wavefronts
    const double mfma_w = (MODE <= 1 || MODE == 4 ? waves : (MODE == 3 ? waves / 2 : 0)), fma_w = (MODE == 2 ? waves : (MODE == 3 ? waves / 2 : 0));
    const double n_mfma = mfma_w * iters * 4 / simds, n_fma = fma_w * iters * 32 / simds;
    printf("%-58s %d wave(s)/SIMD  %.3f ms", name, waves_per_simd, best);
    if (n_mfma > 0) printf("  %.1f ns per MFMA per SIMD (%.1f TFLOP/s)", best * 1e6 / n_mfma, mfma_w * iters * 4 * 2048.0 / best / 1e9);
    if (n_fma > 0) printf("  %.2f ns per FMA per SIMD (%.1f TFLOP/s)", best * 1e6 / n_fma, fma_w * iters * 32 * 128.0 / best / 1e9);
    printf("\n");
}


// one VALU instruction in eight independent chains (32 per iteration), 4 wavefronts per SIMD
#define OPK(NAME, ASM)                                                                                     \
__global__ __launch_bounds__(256) void NAME(double *out, int iters, double x, double y) {                  \
    double c[8];                                                                                           \
    for (int q = 0; q < 8; ++q) c[q] = x + q * 1e-3 + threadIdx.x * 1e-6;                                  \
    int e = (int)(threadIdx.x & 3);                                                                        \
    for (int i = 0; i < iters; ++i)                                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                      \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) asm volatile(ASM : "+v"(c[q]) : "v"(y), "v"(e)); \
    double res = 0;                                                                                        \
    for (int q = 0; q < 8; ++q) res += c[q];                                                               \
    if (res == 1234.5) out[blockIdx.x] = res;                                                              \
}
OPK(k_add, "v_add_f64 %0, %0, %1")
OPK(k_mul, "v_mul_f64 %0, %0, %1")
OPK(k_max, "v_max_f64 %0, %0, %1")
OPK(k_ldexp, "v_ldexp_f64 %0, %0, %2")
OPK(k_rndne, "v_rndne_f64 %0, %0")
OPK(k_rcp, "v_rcp_f64 %0, %0")
OPK(k_frexp, "v_frexp_mant_f64 %0, %0")
OPK(k_mov64, "v_mov_b64 %0, %1")
#define OPK32(NAME, ASM)                                                                                   \
__global__ __launch_bounds__(256) void NAME(double *out, int iters, double x, double y) {                  \
    int c[8];                                                                                              \
    for (int q = 0; q < 8; ++q) c[q] = (int)threadIdx.x + q;                                               \
    int e = (int)(threadIdx.x & 3);                                                                        \
    for (int i = 0; i < iters; ++i)                                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                      \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) asm volatile(ASM : "+v"(c[q]) : "v"(e));         \
    int res = 0;                                                                                           \
    for (int q = 0; q < 8; ++q) res += c[q];                                                               \
    if (res == 12345) out[blockIdx.x] = res;                                                               \
}
OPK32(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc")
OPK32(k_add32, "v_add_u32 %0, %0, %1")
OPK32(k_cnd64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
OPK32(k_cmpcnd, "v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
OPK32(k_cmpcnd64, "v_cmp_gt_u32 s[10:11], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]")
OPK32(k_cmp32, "v_cmp_gt_u32 vcc, %0, %1")
OPK32(k_bfi, "v_bfi_b32 %0, %1, %0, %1")
OPK32(k_and, "v_and_b32 %0, %0, %1")
OPK32(k_cnd_novcc, "v_cndmask_b32_e64 %0, %0, %1, s[12:13]")
OPK(k_cmp, "v_cmp_gt_f64 vcc, %0, %1")
template <typename K>
static void run_op(const char *name, K kern, double *out, int blocks = 1024) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const int iters = 20000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.9999999);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("%-22s %d wave(s)/SIMD  %.3f ms  %.2f ns per instruction per SIMD\n", name, blocks / 256, best, best * 1e6 / ((double)blocks * 4 * iters * 32 / 1024.0));
}

int main() {
    double *out;
    CHK(hipMalloc(&out, 1 << 20));
    for (int w : {1, 2, 4}) run<0>("MFMA f64 16x16x4, one dependent chain", w, out);
    for (int w : {1, 2, 4}) run<1>("MFMA f64 16x16x4, four independent chains", w, out);
    for (int w : {1, 2, 4}) run<4>("MFMA f64 16x16x4, four chains, 16 A / B register pairs", w, out);
    for (int w : {1, 2, 4}) run<2>("v_fma_f64, eight independent chains", w, out);
    for (int w : {2, 4}) run<3>("MFMA wavefronts beside v_fma_f64 wavefronts on the same SIMDs", w, out);
    run_op("v_add_f64", k_add, out);
    run_op("v_mul_f64", k_mul, out);
    run_op("v_max_f64", k_max, out);
    run_op("v_ldexp_f64", k_ldexp, out);
    run_op("v_rndne_f64", k_rndne, out);
    run_op("v_rcp_f64", k_rcp, out);
    run_op("v_frexp_mant_f64", k_frexp, out);
    run_op("v_mov_b64", k_mov64, out);
    run_op("v_cndmask_b32", k_cnd, out);
    run_op("v_add_u32", k_add32, out);
    run_op("v_cndmask_b32 (vcc)", k_cnd, out, 256);
    run_op("v_cndmask_b32 (vcc)", k_cnd, out, 512);
    run_op("v_cndmask_b32_e64 sgpr", k_cnd64, out);
    run_op("v_cndmask_b32_e64 sgpr", k_cnd64, out, 256);
    run_op("v_cmp_gt_u32 vcc", k_cmp32, out);
    run_op("cmp vcc + cndmask vcc", k_cmpcnd, out);
    run_op("cmp sgpr + cndmask sgpr", k_cmpcnd64, out);
    run_op("v_bfi_b32", k_bfi, out);
    run_op("v_and_b32", k_and, out);
    run_op("v_cmp_gt_f64", k_cmp, out);
    return 0;
}
