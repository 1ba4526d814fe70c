// dma_probe.hip - HBM read rate of LDS-DMA rings in the M-step's tile access pattern (tools only, not part of the product).
// One workgroup (4 wavefronts) per 64-row tile, rows at a pitch of ~8.5 KB, rings of RT slots in LDS, one barrier per slot,
// counted vmcnt.  Question: does the PIECE shape matter - 8 rows x 128 B per wave-instruction (k3_mstep's 16-bin half-chunks)
// against 4 rows x 256 B (32-bin chunks) - and how deep must the ring be at 3 workgroups per CU?
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe tools/dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(unsigned off_bytes, const double *sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off_bytes), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// BINS = 16: slot = 64 rows x 128 B (8 KB), 2 instructions per wavefront and slot (8 rows x 128 B each);
// BINS = 32: slot = 64 rows x 256 B (16 KB), 4 instructions per wavefront and slot (4 rows x 256 B each).
// KEEP = slots whose DMAs may stay in flight across the wait for slot h (at most RT - 2: the wait is in issue order).
template <int BINS, int RT, int KEEP, int WPC>
__global__ __launch_bounds__(256, WPC) void k_ring(const double *__restrict__ base, size_t tile_stride, int pitch, int ncols,
                                                   double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NI = BINS / 8;                 // instructions per wavefront and slot
    constexpr int SLOT = 64 * BINS;              // f64
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds;
    const unsigned long long tb = (unsigned long long)(base + (size_t)blockIdx.x * tile_stride);
    const unsigned tlo = __builtin_amdgcn_readfirstlane((unsigned)tb), thi = __builtin_amdgcn_readfirstlane((unsigned)(tb >> 32));
    const double *t = reinterpret_cast<const double *>(((unsigned long long)thi << 32) | tlo);   // (unsigned: the builtin returns int)
    unsigned off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int lpr = BINS / 2;                                   // lanes per row
        const int r = 16 * wave + (64 / lpr) * i + lane / lpr, u = lane % lpr;
        off[i] = (unsigned)(((size_t)r * pitch + 2 * u) * 8);
    }
    const int nh = ncols / BINS;
    auto issue = [&](int h) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            dma16(off[i], t + (size_t)h * BINS, lds0 + (unsigned)(h % RT) * (SLOT * 8) + (unsigned)(NI * wave + i) * 1024u);
    };
#pragma unroll
    for (int j = 0; j < RT - 1; ++j)
        if (j < nh) issue(j);
    double acc = 0.0;
    for (int h = 0; h < nh; ++h) {
        if (h + RT - 2 < nh) wait_vm<NI *(RT - 2 < KEEP ? RT - 2 : KEEP)>(); else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (h + RT - 1 < nh) issue(h + RT - 1);
        acc += lds[(h % RT) * SLOT + tid];          // one LDS read per slot stands in for the fragment reads
    }
    if (acc == 1234.5) out[blockIdx.x] = acc;
}

template <int BINS, int RT, int KEEP, int WPC>
static void run(const char *name, const double *buf, size_t tile_stride, int pitch, int ncols, int n_tiles, double *out) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const size_t lds = (size_t)RT * 64 * BINS * 8;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring<BINS, RT, KEEP, WPC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((k_ring<BINS, RT, KEEP, WPC>), dim3(n_tiles), dim3(256), lds, 0, buf, tile_stride, pitch, ncols, out);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("%-64s LDS %3zu KB  %.3f ms  %.2f TB/s\n", name, lds >> 10, best, (double)64 * ncols * n_tiles * 8 / best / 1e9);
}

int main() {
    const int ncols = 672, pitch = 1072, n_tiles = 32768;     // a live extent of 672 of 1,072 bins: 344 KB per tile, 11.3 GB in all
    const size_t tile_stride = (size_t)64 * pitch;
    double *buf, *out;
    CHK(hipMalloc(&buf, tile_stride * n_tiles * 8));
    CHK(hipMalloc(&out, n_tiles * 8));
    CHK(hipMemset(buf, 0, tile_stride * n_tiles * 8));
    run<16, 4, 1, 3>("128-B pieces, 4 x 8 KB slots, 1 slot kept in flight, 3 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<16, 4, 2, 3>("128-B pieces, 4 x 8 KB slots, 2 kept, 3 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<16, 5, 1, 3>("128-B pieces, 5 x 8 KB slots, 1 kept, 3 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<16, 5, 3, 3>("128-B pieces, 5 x 8 KB slots, 3 kept, 3 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<16, 6, 4, 3>("128-B pieces, 6 x 8 KB slots, 4 kept, 3 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<32, 3, 1, 3>("256-B pieces, 3 x 16 KB slots, 1 kept, 3 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<32, 4, 2, 2>("256-B pieces, 4 x 16 KB slots, 2 kept, 2 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<16, 8, 6, 2>("128-B pieces, 8 x 8 KB slots, 6 kept, 2 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<16, 4, 2, 4>("128-B pieces, 4 x 8 KB slots, 2 kept, 4 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    run<16, 3, 1, 5>("128-B pieces, 3 x 8 KB slots, 1 kept, 5 WG/CU", buf, tile_stride, pitch, ncols, n_tiles, out);
    return 0;
}
