// stride_probe.hip - HBM read rate of the M-step's access pattern: does it matter that a tile chunk is 64 pieces of
// 256 B at a row pitch of ~8.5 KB instead of one contiguous 16 KB block?  (layout question, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one workgroup per tile: 64 rows x ncols doubles; per step every thread loads one double2: 4 waves x (4 rows x 16 lanes)
// MODE 0: rows at pitch `pitch` doubles (the tensor as it is); MODE 1: the tile stored chunk-major (contiguous 16 KB chunks)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 3) void k_tile(const double *__restrict__ base, size_t tile_stride, int pitch, int ncols,
                                                 double *__restrict__ out) {
    const double *t = base + (size_t)blockIdx.x * tile_stride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, quarter = lane >> 4, ns = 2 * (lane & 15);
    double acc = 0.0;
    const int nchunks = ncols / 32;
    for (int c0 = 0; c0 < nchunks; c0 += DEPTH) {
        double2 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 4 * wave + quarter + 16 * it, c = c0 + d;
                const double *p = (MODE == 0) ? t + (size_t)row * pitch + c * 32 + ns
                                              : t + (size_t)c * (64 * 32) + row * 32 + ns;
                v[d][it] = (c < nchunks) ? *reinterpret_cast<const double2 *>(p) : make_double2(0, 0);
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int it = 0; it < 4; ++it) acc += v[d][it].x + v[d][it].y;
    }
    if (acc == 1234.5) out[blockIdx.x] = acc;
}

int main() {
    const int ncols = 1056, pitch = 1056, n_tiles = 32768;     // 64 x 1056 x 8 B = 540 KB per tile, 17.7 GB in all
    const size_t tile_stride = (size_t)64 * pitch;
    double *buf, *out;
    CHK(hipMalloc(&buf, tile_stride * n_tiles * 8));
    CHK(hipMalloc(&out, n_tiles * 8));
    CHK(hipMemset(buf, 0, tile_stride * n_tiles * 8));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    auto run = [&](int which) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(a));
            if (which == 0) hipLaunchKernelGGL((k_tile<0, 2>), dim3(n_tiles), dim3(256), 0, 0, buf, tile_stride, pitch, ncols, out);
            if (which == 1) hipLaunchKernelGGL((k_tile<1, 2>), dim3(n_tiles), dim3(256), 0, 0, buf, tile_stride, pitch, ncols, out);
            if (which == 2) hipLaunchKernelGGL((k_tile<0, 4>), dim3(n_tiles), dim3(256), 0, 0, buf, tile_stride, pitch, ncols, out);
            if (which == 3) hipLaunchKernelGGL((k_tile<1, 4>), dim3(n_tiles), dim3(256), 0, 0, buf, tile_stride, pitch, ncols, out);
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        const char *names[] = {"row-pitch layout, 2 chunks in flight", "chunk-major layout, 2 chunks in flight",
                               "row-pitch layout, 4 chunks in flight", "chunk-major layout, 4 chunks in flight"};
        printf("%-42s %.3f ms  %.2f TB/s\n", names[which], best, (double)tile_stride * n_tiles * 8 / best / 1e9);
    };
    for (int w = 0; w < 4; ++w) run(w);
    return 0;
}
