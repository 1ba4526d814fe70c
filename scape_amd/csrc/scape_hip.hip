// scape_hip.hip - gfx950 (MI355X) kernels + C ABI for SCAPE's infer_pa hot path.
//
// What is computed is specified by the reference's Python (citations: /root/reference/src/scape/):
//   Phase A  point log-likelihoods        taichi_core.py:101-157, apa_core.py:954-957
//   Phase B  marginal tensor over (a, b)  taichi_core.py:160-246
//   EM       em_algo / mstep / elbo / bic apa_core.py:473-573, :702-779
//   labels   get_label                    apa_core.py:873-881
// How it is computed is MI355X-first: every array keeps the bin axis (n) fastest so that the
// 64 lanes of a wavefront read/write consecutive f64; rows are padded to a 16-element pitch so
// rows start 128-B aligned and can be read as double2.  The EM advances all (UTR, K, restart) jobs of a call
// in lock-step rounds (em_lockstep.inc): k2_estep, one wavefront per job (responsibilities, weights, ELBO),
// then k2_mstep, one workgroup per (UTR, 64-row tensor tile), which multiplies the tile with the v vectors of
// every job whose window covers it on the f64 matrix cores and keeps per-(job, tile) first arg-maxes.
// k_em below is the round-1 job-at-a-time kernel (SCAPE_HIP_EM=v1, A/B runs only).
// All arithmetic is f64 (the reference's finite -inf sentinel overflows f32).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <chrono>
#include <vector>

#include "scape_hip.h"

#define SENT SCAPE_SENT
#define PI_REF 3.141592653589793  // taichi_core.py:9
#define PITCH 16                  // row pitch granule (f64 elements) -> 128-B aligned rows
#define TILE_ROWS 64               // tensor rows per M-step tile (= MT_ROWS of em_lockstep.inc)
#define EM_THREADS 256
#define EM_WAVES (EM_THREADS / 64)

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                        ":" + std::to_string(__LINE__) + ")");                               \
    } while (0)

// ------------------------------------------------------------------------------------------
// device-side descriptors
// ------------------------------------------------------------------------------------------
struct UtrDesc {
    int64_t bin_off;    // first bin in x/l/r/pa/cnt
    int64_t theta_off;  // first grid point in theta
    int64_t at_off;     // offset (f64) of AT / V  [T][Np]
    int64_t m_off;      // offset (f64) of M       [T][B][Np]
    int64_t log_off;    // first entry in the log-domain bin list
    int64_t tile_off;   // first entry of this UTR in the per-tile extent table (64-row tiles of M)
    int64_t mlog_off;   // first entry of this UTR in the compact table of log-domain values (phase_b_split.inc)
    int32_t c_lo, c_hi; // uniform theta grid: the interior grid points [c_lo, c_hi] share one window table; c_lo > c_hi: none
    int32_t N, Np, T, n_log;
    double unif_ll, L, min_theta;
};

struct DevParams {
    double mu_f, sigma_f, max_unif_ws;
    int32_t B, S, nround, pad;
    double betas[SCAPE_MAX_BETA];
    double s_dis[SCAPE_MAX_S];
    double pmf_s[SCAPE_MAX_S];
    double inv_s[SCAPE_MAX_S];   // 1 / s_dis[j] (the same IEEE quotient the kernels would compute per bin)
};

// ------------------------------------------------------------------------------------------
// device functions: the reference's @ti.func set (taichi_core.py:24-97)
// ------------------------------------------------------------------------------------------
// log(x) for finite x > 0 (subnormals included) in ~35 f64 instructions instead of the ~85 of the library's
// double-double routine - the tensor build takes one log per entry (13 x N x T per UTR) and is bound by them.
// Algorithm: the classic argument reduction x = 2^k (1+f), sqrt(2)/2 <= 1+f < sqrt(2), s = f/(2+f),
// log(1+f) = f - hfsq + s (hfsq + R(s^2)) with the degree-14 minimax R in s (coefficients Lg1..Lg7, error < 2^-58.45),
// k ln2 added as a hi/lo pair; the quotient comes from v_rcp_f64 plus two Newton steps and a residual correction.
// Worst error found on 2e7 arguments (host replica of exactly this sequence, against long-double log): 0.83 ulp.
__device__ __forceinline__ double d_log_pos(double x) {
#pragma clang fp contract(off)
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);        // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m * 2.0 : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0, d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = __builtin_fma(__builtin_fma(-d, s, f), r, s);
    const double dk = (double)e, z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double inner = __builtin_fma(s, hfsq + R, dk * ln2_lo);
    return dk * ln2_hi - ((hfsq - inner) - f);
}

#include "exp_table.inc"
// exp(x) for x <= 0 in ~18 f64 instructions + one 16-byte LDS read (the library routine takes ~37): the E-step
// takes one exp per (bin, component, round), Phase A thirteen per (bin, theta) - both are bound by them.
// x = (128 e + j) ln2/128 + r, |r| <= ln2/256: exp(x) = 2^e * T[j] * (1 + p(r)), T[j] = 2^(j/128) as a hi/lo pair
// (table in LDS: d_load_exptab), p = expm1 to degree 5 (next term r^6/720 < 2^-62).  Arguments below -750 give 0
// like exp itself.  Worst error on 3e7 arguments in [-700, 0] (host replica, against long-double exp): 0.60 ulp.
__device__ __forceinline__ void d_load_exptab(double2 *tab /* LDS, 128 entries */, int tid, int nthreads) {
    for (int i = tid; i < 128; i += nthreads) tab[i] = g_exp_tab[i];
}
__device__ __forceinline__ double d_exp_nonpos(double x, const double2 *tab) {
#pragma clang fp contract(off)
    x = fmax(x, -750.0);
    const double n = __builtin_rint(x * EXP_INV_L);
    double r = __builtin_fma(n, -EXP_L_HI, x);
    r = __builtin_fma(n, -EXP_L_LO, r);
    const int ni = (int)n;
    const double2 t = tab[ni & 127];
    const double r2 = r * r;
    const double q = __builtin_fma(r, __builtin_fma(r, __builtin_fma(r, 1.0 / 120, 1.0 / 24), 1.0 / 6), 0.5);
    const double p = __builtin_fma(r2, q, r);
    return ldexp(t.x + __builtin_fma(t.x, p, t.y), ni >> 7);
}

// 1/x for x in [2^-20, 2^20] (no scaling or special cases needed): v_rcp_f64 + two Newton steps, within an ulp of the quotient
__device__ __forceinline__ double d_rcp_mid(double x) {
#pragma clang fp contract(off)
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double d_logw(double w) { return (w <= 0.0) ? SENT : log(w); }
__device__ __forceinline__ double d_logpdf_normal(double x, double mu, double sigma) {
    double z = (x - mu) / sigma;
    return -0.5 * (z * z) - log(sigma) - 0.5 * log(2 * PI_REF);
}
__device__ __forceinline__ double d_pdf_normal(double x, double mu, double sigma) {
    double z = (x - mu) / sigma;
    return exp(-0.5 * (z * z)) / sqrt(2 * PI_REF) / sigma;
}

__device__ __forceinline__ double d_pdf_normal_t(double x, double mu, double sigma, const double2 *tab) {
    double z = (x - mu) / sigma;
    return d_exp_nonpos(-0.5 * (z * z), tab) / sqrt(2 * PI_REF) / sigma;
}

// loglik_xlr_t_pa_kernel body (taichi_core.py:101-107)
__device__ __forceinline__ double d_point_pa(double x, double l, double pa, double th, double sigma_f) {
    double u = th - x;
    double ll_l = (l <= u) ? -log(u) : SENT;
    return ll_l + d_logpdf_normal(pa - th, 0.0, sigma_f);
}
// loglik_xlr_t_r_unknown_kernel body (taichi_core.py:141-157); returns v (linear), A through *a
__device__ __forceinline__ double d_point_r_unknown(double x, double l, double th, const double *s,
                                                    const double *pmf, const double *inv_s, int S, double mu_f,
                                                    double sigma_f, double *a, const double2 *tab) {
    double u = th - x;
    double v = 0.0;
    if (l <= u) {  // lik_l_xt == 0 otherwise, every term of the sum is then exactly 0
        // 1/s_j * pdfN(x; th + s_j - mu_f, sigma_f) * lk * pmf_j, summed in j order.  The reference divides four times
        // per term (1/s_j, z = ./sigma, /sqrt(2 pi), /sigma); here the three divisors that do not depend on the bin
        // become reciprocals taken once (each product then differs from the quotient by at most an ulp: 4e-16 relative
        // on v, against the 1e-13 the log-domain comparison with the reference allows) - 52 divisions of ~14
        // instructions per (bin, theta) were three quarters of this kernel
        const double lk = 1 / u, inv_sigma = 1 / sigma_f, inv_norm = 1 / sqrt(2 * PI_REF);
        for (int j = 0; j < S; ++j) {
            const double z = (x - (th + s[j] - mu_f)) * inv_sigma;
            const double pdf = d_exp_nonpos(-0.5 * (z * z), tab) * inv_norm * inv_sigma;
            v += inv_s[j] * pdf * lk * pmf[j];
        }
    }
    if (v < 1e-300) v = 0.0;
    *a = (v <= 0.0) ? SENT : d_log_pos(v);
    return v;
}
// loglik_xlr_t_r_known_kernel body (taichi_core.py:111-132), two-pass logsumexp (:40-54)
__device__ __forceinline__ double d_point_r_known(double x, double l, double r, double th,
                                                  const double *s, const double *pmf, int S,
                                                  double mu_f, double sigma_f) {
    double u = th - x;
    double ll_l = (l <= u) ? -log(u) : SENT;
    double tmpn = 0.0, mx = 0.0;
    for (int j = 0; j < S; ++j) {
        double t;
        if (s[j] < r) {
            t = SENT;
        } else {
            tmpn += pmf[j];
            double lrs = (r <= s[j]) ? -log(s[j]) : SENT;
            t = lrs + d_logpdf_normal(x, th + s[j] - mu_f, sigma_f) + ll_l + log(pmf[j]);
        }
        if (j == 0 || t > mx) mx = t;
    }
    double sum = 0.0;
    for (int j = 0; j < S; ++j) {
        double t;
        if (s[j] < r) {
            t = SENT;
        } else {
            double lrs = (r <= s[j]) ? -log(s[j]) : SENT;
            t = lrs + d_logpdf_normal(x, th + s[j] - mu_f, sigma_f) + ll_l + log(pmf[j]);
        }
        sum += exp(t - mx);
    }
    return log(sum) + mx - log(tmpn);
}

// np.sum over a short contiguous f64 run, in numpy's own order (sequential below 8 elements,
// 8 interleaved accumulators above) so that row normalisations round like the reference's
template <int CMAX>
__device__ __forceinline__ double d_np_sum(const double (&a)[CMAX], int C) {
    if (C < 8) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < (CMAX < 7 ? CMAX : 7); ++i)
            if (i < C) r += a[i];
        return r;
    }
    if constexpr (CMAX >= 8) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        const int lim = C - (C % 8);
#pragma unroll
        for (int i = 8; i + 8 <= CMAX; i += 8)
            if (i < lim) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] += a[i + j];
            }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = 8; i < CMAX; ++i)
            if (i >= lim && i < C) res += a[i];
        return res;
    }
    return 0.0;
}

__device__ __forceinline__ double d_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

// ------------------------------------------------------------------------------------------
// operator-level kernels (the taichi_core seam: one theta, one data subset)
// ------------------------------------------------------------------------------------------
__global__ void k_op_pa(const double *x, const double *l, const double *pa, int n, double theta,
                        double sigma_f, double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = d_point_pa(x[i], l[i], pa[i], theta, sigma_f);
}
__global__ void k_op_r_known(const double *x, const double *l, const double *r, int n, DevParams P,
                             double theta, double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = d_point_r_known(x[i], l[i], r[i], theta, P.s_dis, P.pmf_s, P.S, P.mu_f, P.sigma_f);
}
__global__ void k_op_r_unknown(const double *x, const double *l, int n, DevParams P, double theta,
                               double *out) {
    __shared__ double2 s_exptab[128];
    d_load_exptab(s_exptab, threadIdx.x, blockDim.x);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double a;
        d_point_r_unknown(x[i], l[i], theta, P.s_dis, P.pmf_s, P.inv_s, P.S, P.mu_f, P.sigma_f, &a, s_exptab);
        out[i] = a;
    }
}

// ------------------------------------------------------------------------------------------
// Phase A: AT[t][n] = log p(bin n | theta_t) and V[t][n] = p(bin n | theta_t) (linear domain,
// r-unknown bins only).  grid = (ceil(Np/256), T_max, n_utr); n is the lane axis.
// ------------------------------------------------------------------------------------------
#define PA_TT 8      // theta grid points per workgroup: the bin's data, the UTR descriptor and the exp table are fetched once for all of them
__global__ __launch_bounds__(256) void k_phase_a(const UtrDesc *__restrict__ descs, DevParams P,
                                                 const double *__restrict__ x,
                                                 const double *__restrict__ l,
                                                 const double *__restrict__ r,
                                                 const double *__restrict__ pa,
                                                 const double *__restrict__ theta,
                                                 double *__restrict__ AT, double *__restrict__ V) {
    __shared__ double2 s_exptab[128];
    d_load_exptab(s_exptab, threadIdx.x, blockDim.x);
    __syncthreads();
    const UtrDesc d = descs[blockIdx.z];
    const int t0 = blockIdx.y * PA_TT, t1 = min(d.T, t0 + PA_TT);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 >= d.T || n >= d.Np) return;
    if (n >= d.N) {
        for (int t = t0; t < t1; ++t) {
            const size_t o = (size_t)d.at_off + (size_t)t * d.Np + n;
            AT[o] = 0.0;
            V[o] = 0.0;
        }
        return;
    }
    const double xn = x[d.bin_off + n], ln = l[d.bin_off + n];
    const double rn = r[d.bin_off + n], pan = pa[d.bin_off + n];
    for (int t = t0; t < t1; ++t) {
        const size_t o = (size_t)d.at_off + (size_t)t * d.Np + n;
        const double th = theta[d.theta_off + t];
        double a, v = 0.0;
        if (!isnan(pan)) {
            a = d_point_pa(xn, ln, pan, th, P.sigma_f);
        } else if (!isnan(rn)) {
            a = d_point_r_known(xn, ln, rn, th, P.s_dis, P.pmf_s, P.S, P.mu_f, P.sigma_f);
        } else {
            v = d_point_r_unknown(xn, ln, th, P.s_dis, P.pmf_s, P.inv_s, P.S, P.mu_f, P.sigma_f, &a, s_exptab);
        }
        AT[o] = a;
        V[o] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Phase B: M[i][j][n] = logsumexp_w( A[n,w] + logN(theta_w; theta_i, beta_j) - G_ij ) over the
// +-3 beta window (taichi_core.py:218-234).  One workgroup per (UTR, i); the window weights of
// all betas are built once in LDS.  r-unknown bins take the linear-domain form
// log( sum_w V[w][n] * exp(g_w - G) ) - mathematically the same sum, no exp per term; bins
// whose A was built in the log domain (pA-site / r-known) take the reference's two-pass form.
// LDS layout (f64): g[B][Wmax] | p[B][Wmax] | G[B] | lo[B] hi[B] (int32 pairs)
// ------------------------------------------------------------------------------------------
template <int BMAX>
__global__ __launch_bounds__(256, 5) void k_phase_b(const UtrDesc *__restrict__ descs, DevParams P,
                                                 const double *__restrict__ r,
                                                 const double *__restrict__ pa,
                                                 const double *__restrict__ theta,
                                                 const int32_t *__restrict__ loglist,
                                                 const double *__restrict__ AT,
                                                 const double *__restrict__ V,
                                                 double *__restrict__ M, int Wmax, int all_log,
                                                 int *__restrict__ err_flag, int n_utr, int T_max,
                                                 int32_t *__restrict__ tile_nend, int probe) {
    extern __shared__ double sm[];
    __shared__ double2 s_exptab[128];
    d_load_exptab(s_exptab, threadIdx.x, 256);
    // blocks b and b+8 share an XCD: all grid points of a UTR go to one XCD so its V rows (each is
    // read by ~43 neighbouring alphas) are fetched into one L2 only.  Placement only affects speed.
    const int id = blockIdx.x, slot = id >> 3;
    const int uu = (slot / T_max) * 8 + (id & 7);
    if (uu >= n_utr) return;
    const UtrDesc d = descs[uu];
    const int i = slot % T_max;
    if (i >= d.T) return;
    const int B = P.B;
    const int tid = threadIdx.x;
    double *g = sm;                       // [B][Wmax]  logN(theta_w; theta_i, beta_j)
    double *p = sm + (size_t)B * Wmax;    // [B][Wmax]  exp(g - G), 0 outside the window
    double *G = p + (size_t)B * Wmax;     // [B]
    int *lo = (int *)(G + B);             // [B]
    int *hi = lo + B;                     // [B]
    const double *th = theta + d.theta_off;
    const double ti = th[i];

    if (tid < B) {
        const double beta = P.betas[tid];
        int a = i, b = i;
        const double lob = ti - 3 * beta, hib = ti + 3 * beta;
        // the walks start from where a uniform grid would put the window's ends and correct in either direction (the
        // grid is ascending, so the boundary is unique): ~6 dependent loads instead of ~40 on the usual grid
        if (d.T > 1) {
            const double step = (i + 1 < d.T) ? th[i + 1] - ti : ti - th[i - 1];
            if (step > 0.0) {
                const double reach = 3 * beta / step;
                const int jump = (reach < 1e6) ? (int)reach : 0;
                a = max(0, i - jump);
                b = min(d.T - 1, i + jump);
            }
        }
        while (a < i && th[a] < lob) ++a;
        while (a > 0 && th[a - 1] >= lob) --a;          // searchsorted(left)
        while (b > i && th[b] > hib) --b;
        while (b + 1 < d.T && th[b + 1] <= hib) ++b;    // searchsorted(right) - 1
        if (b - a + 1 > Wmax) {
            atomicExch(err_flag, 1);
            b = a + Wmax - 1;
        }
        lo[tid] = a;
        hi[tid] = b;
    }
    __syncthreads();
    for (int e = tid; e < B * Wmax; e += blockDim.x) {
        const int j = e / Wmax, w = e - j * Wmax;
        const int tw = lo[j] + w;
        g[e] = (tw <= hi[j]) ? d_logpdf_normal(th[tw], ti, P.betas[j]) : 0.0;
    }
    __syncthreads();
    if (tid < B) {  // call_logp_theta_sum_kernel (taichi_core.py:160-169), serial order
        double psum = 0.0;
        const int W = hi[tid] - lo[tid] + 1;
        if (probe & 4) psum = 1.0;
        else
        for (int w = 0; w < W; ++w) psum += exp(g[(size_t)tid * Wmax + w]);
        G[tid] = log(psum);
    }
    __syncthreads();
    int lo_all = lo[0], hi_all = hi[0];
    for (int j = 1; j < B; ++j) {
        lo_all = min(lo_all, lo[j]);
        hi_all = max(hi_all, hi[j]);
    }
    const int Wall = hi_all - lo_all + 1;  // <= Wmax when windows nest (they do: symmetric in value)
    // p re-indexed relative to lo_all so the linear path needs no per-beta window test
    for (int e = tid; e < B * Wmax; e += blockDim.x) {
        const int j = e / Wmax, w = e - j * Wmax;
        const int tw = lo_all + w;
        double val = 0.0;
        if (w < Wall && tw >= lo[j] && tw <= hi[j]) val = exp(g[(size_t)j * Wmax + (tw - lo[j])] - G[j]);
        p[e] = val;
    }
    __syncthreads();
    if (Wall > Wmax) {
        if (tid == 0) atomicExch(err_flag, 2);
        return;
    }

    double *Mi = M + (size_t)d.m_off + (size_t)i * B * d.Np;
    // Tile extents (used by the M-step, em_lockstep.inc): bins are ordered by read start, so each tensor row is finite
    // on a prefix of the bins and holds the sentinel (read impossible) on the rest - 35-50 % of all entries.  For
    // every TILE_ROWS-row tile the table gets one past the last bin that is finite in ANY of its rows (rounded up to
    // 16): beyond it every row of the tile is the sentinel, and the M-step replaces that part of the dot product by
    // SENT * sum(v) instead of streaming it.  The rows of this workgroup lie in at most two tiles.
    const int tile0 = (i * B) / TILE_ROWS, j_split = (tile0 + 1) * TILE_ROWS - i * B;   // rows j >= j_split: tile0 + 1
    int fin0 = 0, fin1 = 0;
    auto note = [&](int j, int n, double val) {
        if (val != SENT) {
            if (j < j_split) fin0 = max(fin0, n + 1);
            else fin1 = max(fin1, n + 1);
        }
    };
    // ---- linear-domain path -----------------------------------------------------------------
    // out[j][n] = sum_w p[j][w] * V[w][n] is a [16 x W] x [W x N] product: on the f64 matrix cores one
    // v_mfma_f64_16x16x4_f64 covers 16 betas (13 used) x 16 bins x 4 taps.  A = weights from LDS (lane l:
    // beta l&15, tap l>>4), B = V straight from global/L2 (lane l: tap l>>4, bin l&15), D lane l reg q:
    // beta (l>>4)+4q, bin l&15.
    if (probe & 2) {
    } else if (!all_log && BMAX == 16) {
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int WP = ((Wmax + 3) & ~3) + 1;                     // odd pitch: conflict-light fragment reads
        double *pm = reinterpret_cast<double *>(hi + B + (B & 1)); // [16][WP], zero outside window / beyond B
        for (int e = tid; e < 16 * WP; e += blockDim.x) {
            const int j = e / WP, w = e - j * WP;
            pm[e] = (j < B && w < Wmax) ? p[(size_t)j * Wmax + w] : 0.0;
        }
        __syncthreads();
        const double *Vu = V + (size_t)d.at_off;
        const int lane = tid & 63, wave = tid >> 6, kq = lane >> 4, rc = lane & 15;
        const int nblocks = d.Np >> 4;
        // Tap steps whose four rows all exist (lo_all + w0 + 3 <= T - 1) run without a row clamp: the loop body is one
        // pointer add, one load, one LDS read and the MFMA per block of bins; the (at most one) step that reaches past
        // the last grid point clamps its row - its weight is 0 there.  Two blocks of 16 bins share every weight fragment.
        const int steps = (Wall + 3) >> 2;
        const int full = max(0, min(steps, (d.T - lo_all) >> 2));
        const size_t rstep = (size_t)4 * d.Np;
        const double *pw = pm + rc * WP + kq;
        const double *vbase = Vu + (size_t)(lo_all + kq) * d.Np + rc;
        // The 16 x 16 result block: row kq + 4q of lane (kq, rc) is beta kq + 4q of bin n.  Written without divergent
        // branches (the logs are taken on every lane, the stores and the extent bookkeeping are predicated): the
        // per-value `if`s of a straightforward version compiled to ~40 exec-mask sequences per block.
        auto finish = [&](int n, const v4d &acc) {
            const bool pad = n >= d.N;
            const int nn = pad ? d.N - 1 : n;
            const bool lin = !pad && isnan(pa[d.bin_off + nn]) && isnan(r[d.bin_off + nn]);   // r-unknown bin: this path's
            const bool wr = pad || lin;
            bool f0 = false, f1 = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = kq + 4 * q;
                const double a = acc[q];
                const bool pos = a > 0.0;
                const double lg = d_log_pos(pos ? a : 1.0);
                const double val = pad ? 0.0 : (pos ? lg : SENT);
                if (wr && j < B) Mi[(size_t)j * d.Np + n] = val;
                const bool fin = lin && pos && j < B;
                f0 |= fin && j < j_split;
                f1 |= fin && j >= j_split;
            }
            fin0 = max(fin0, f0 ? n + 1 : 0);
            fin1 = max(fin1, f1 ? n + 1 : 0);
        };
        int nb = wave;
        for (; nb + 4 < nblocks; nb += 8) {
            const double *v0 = vbase + nb * 16, *v1 = v0 + 64;
            v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            int st = 0;
            // Four tap steps per trip: their eight V fragments are requested together, then consumed - one wait on the
            // L2 per trip instead of one per step.  (The loop is bound by that latency, not by issue: 8 or 31
            // instructions per MFMA made no difference, batching the loads took 1.3 ms off; the 1-3 steps left over
            // are batched the same way: another 0.5 ms.  Measured and not kept: predicated batches of 4 / 6 / 12 steps
            // (the compiler serialises them), the next pair's first batch requested before this pair's logs and
            // stores (3 wavefronts per SIMD instead of 4: +1.5 ms; held to 4: -0.15 ms), batched loads in the
            // log-domain path and the grid points of the window searches in LDS (registers: 3 wavefronts, +2 ms).)
            auto batch = [&](auto nbc) {
                constexpr int NB = decltype(nbc)::value;
                double b0[NB], b1[NB], av[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    b0[k] = v0[k * rstep];
                    b1[k] = v1[k * rstep];
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) av[k] = pw[4 * (st + k)];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], b0[k], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], b1[k], acc1, 0, 0, 0);
                }
                v0 += NB * rstep;
                v1 += NB * rstep;
                st += NB;
            };
            while (st + 4 <= full) batch(std::integral_constant<int, 4>());
            switch (full - st) {
                case 3: batch(std::integral_constant<int, 3>()); break;
                case 2: batch(std::integral_constant<int, 2>()); break;
                case 1: batch(std::integral_constant<int, 1>()); break;
                default: break;
            }
            for (; st < steps; ++st) {
                const int wr = min(lo_all + 4 * st + kq, d.T - 1);
                const double av = pw[4 * st];
                const double *vr = Vu + (size_t)wr * d.Np + nb * 16 + rc;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, vr[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, vr[64], acc1, 0, 0, 0);
            }
            finish(nb * 16 + rc, acc0);
            finish(nb * 16 + 64 + rc, acc1);
        }
        for (; nb < nblocks; nb += 4) {
            const double *v0 = vbase + nb * 16;
            v4d acc0 = {0.0, 0.0, 0.0, 0.0};
            int st = 0;
#pragma unroll 4
            for (; st < full; ++st) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pw[4 * st], *v0, acc0, 0, 0, 0);
                v0 += rstep;
            }
            for (; st < steps; ++st) {
                const int wr = min(lo_all + 4 * st + kq, d.T - 1);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pw[4 * st], Vu[(size_t)wr * d.Np + nb * 16 + rc], acc0, 0, 0, 0);
            }
            finish(nb * 16 + rc, acc0);
        }
    } else if (!all_log) {
        const double *Vu = V + (size_t)d.at_off;
        for (int n = tid; n < d.Np; n += blockDim.x) {
            if (n >= d.N) {
                for (int j = 0; j < B; ++j) Mi[(size_t)j * d.Np + n] = 0.0;
                continue;
            }
            if (!isnan(pa[d.bin_off + n]) || !isnan(r[d.bin_off + n])) continue;
            if (B <= BMAX) {
                double acc[BMAX];
#pragma unroll
                for (int j = 0; j < BMAX; ++j) acc[j] = 0.0;
                for (int w = 0; w < Wall; ++w) {
                    const double v = Vu[(size_t)(lo_all + w) * d.Np + n];
#pragma unroll
                    for (int j = 0; j < BMAX; ++j)
                        if (j < B) acc[j] += v * p[(size_t)j * Wmax + w];
                }
#pragma unroll
                for (int j = 0; j < BMAX; ++j)
                    if (j < B) {
                        const double val = (acc[j] > 0.0) ? d_log_pos(acc[j]) : SENT;
                        Mi[(size_t)j * d.Np + n] = val;
                        note(j, n, val);
                    }
            } else {
                for (int j = 0; j < B; ++j) {
                    double acc = 0.0;
                    for (int w = 0; w < Wall; ++w)
                        acc += Vu[(size_t)(lo_all + w) * d.Np + n] * p[(size_t)j * Wmax + w];
                    const double val = (acc > 0.0) ? d_log_pos(acc) : SENT;
                    Mi[(size_t)j * d.Np + n] = val;
                    note(j, n, val);
                }
            }
        }
    } else if (all_log) {
        for (int n = d.N + tid; n < d.Np; n += blockDim.x)
            for (int j = 0; j < B; ++j) Mi[(size_t)j * d.Np + n] = 0.0;
    }
    // ---- log-domain path (cal_res_kernel, taichi_core.py:172-179) ------------------------------
    const int n_log = (probe & 1) ? 0 : (all_log ? d.N : d.n_log);
    const double *Au = AT + (size_t)d.at_off;
    for (int item = tid; item < n_log * B; item += blockDim.x) {
        const int j = item / n_log, q = item - j * n_log;
        const int n = all_log ? q : loglist[d.log_off + q];
        const int a = lo[j], W = hi[j] - lo[j] + 1;
        const double Gj = G[j];
        const double *gj = g + (size_t)j * Wmax;
        double mx = Au[(size_t)a * d.Np + n] + gj[0] - Gj;
        for (int w = 1; w < W; ++w) {
            const double t = Au[(size_t)(a + w) * d.Np + n] + gj[w] - Gj;
            if (t > mx) mx = t;
        }
        double sum = 0.0;
        // every argument is <= 0 (mx is the maximum): the table exp of the E-step, 18 instructions instead of ~40
        for (int w = 0; w < W; ++w) sum += d_exp_nonpos(Au[(size_t)(a + w) * d.Np + n] + gj[w] - Gj - mx, s_exptab);
        const double val = d_log_pos(sum) + mx;
        Mi[(size_t)j * d.Np + n] = val;
        note(j, n, val);
    }
    if (tile_nend) {
        __shared__ int s_fin[2];
        if (tid < 2) s_fin[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            fin0 = max(fin0, __shfl_xor(fin0, off, 64));
            fin1 = max(fin1, __shfl_xor(fin1, off, 64));
        }
        if ((tid & 63) == 0) {
            atomicMax(&s_fin[0], fin0);
            atomicMax(&s_fin[1], fin1);
        }
        __syncthreads();
        if (tid < 2 && s_fin[tid] > 0)
            atomicMax(&tile_nend[d.tile_off + tile0 + tid], min(d.Np, (s_fin[tid] + 15) & ~15));
    }
}

#ifdef SCAPE_HIP_TOOLS   // the round-1 job-at-a-time EM kernel: A/B builds only (-DSCAPE_HIP_TOOLS), not in the product library
// ------------------------------------------------------------------------------------------
// EM: one workgroup per job (UTR, K, restart).  apa_core.py:714-779.
// Column c of log_zmat is never stored: it is snap_lw[c] + M[snap_ia[c]][snap_ib[c]][:] where
// the snapshot is what cal_z_k used the last time column c was refreshed (stale-column E-step,
// apa_core.py:731).  Per round: one pass over the bins (responsibilities, weight sums, ELBO
// terms, the Z[:,k]*cnt vector into LDS), then the grid arg-max over the (alpha, beta) window.
// ------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(EM_THREADS) void k_em(
    const UtrDesc *__restrict__ descs, DevParams P, const double *__restrict__ cnt,
    const double *__restrict__ M, int kmax, const int32_t *__restrict__ job_utr,
    const int32_t *__restrict__ job_K, const int32_t *__restrict__ job_fixed,
    const int32_t *__restrict__ a_in, const int32_t *__restrict__ b_in,
    const double *__restrict__ ws_in, const int8_t *__restrict__ k_arr,
    int32_t *__restrict__ a_out, int32_t *__restrict__ b_out, double *__restrict__ ws_out,
    double *__restrict__ bic_out, int32_t *__restrict__ nlb_out, double *__restrict__ lb_out,
    unsigned long long *__restrict__ counters) {
    extern __shared__ double vk[];  // [Np] Z[:,k] * cnt of the component being updated
    __shared__ double s_red[EM_WAVES][CMAX + 3];
    __shared__ double s_tot[CMAX + 3];
    __shared__ double s_ws[CMAX], s_snaplw[CMAX];
    __shared__ int s_ia[CMAX], s_ib[CMAX], s_sia[CMAX], s_sib[CMAX];
    __shared__ double s_best[EM_WAVES];
    __shared__ int s_bestr[EM_WAVES];
    __shared__ int s_flag;

    const int job = blockIdx.x;
    const UtrDesc d = descs[job_utr[job]];
    const int K = job_K[job], C = K + 1;
    const bool fixed = job_fixed[job] != 0;
    const int B = P.B, nround = P.nround;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = d.N, Np = d.Np;
    const double *Mu = M + (size_t)d.m_off;
    const double *cu = cnt + d.bin_off;
    const int8_t *ka = k_arr + (size_t)job * nround;
    double *lb_arr = lb_out + (size_t)job * nround;

    if (tid < C) {
        s_ws[tid] = ws_in[(size_t)job * (kmax + 1) + tid];
        s_snaplw[tid] = d_logw(s_ws[tid]);                                   // :722-724
        if (tid < K) {
            s_ia[tid] = s_sia[tid] = a_in[(size_t)job * kmax + tid];
            s_ib[tid] = s_sib[tid] = b_in[(size_t)job * kmax + tid];
        }
    }
    __syncthreads();

    double lb = SENT, ell_last = 0.0;
    int n_lb = 0;
    unsigned long long slab_rows = 0;

    for (int it = 0; it < nround; ++it) {
        const int k = ka[it];
        if (tid == 0) {  // cal_z_k(para, k_arr[i]) (:731): refresh the snapshot of column k
            s_snaplw[k] = d_logw(s_ws[k]);
            if (k < K) {
                s_sia[k] = s_ia[k];
                s_sib[k] = s_ib[k];
            }
        }
        __syncthreads();

        size_t roff[CMAX];
        double slw[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            roff[c] = (c < K) ? ((size_t)s_sia[c] * B + s_sib[c]) * Np : 0;
            slw[c] = (c < C) ? s_snaplw[c] : 0.0;
        }

        // ---- E pass (norm_z :490-495, mstep head :525-529, maximize_ws numerators :499,
        //      exp_log_lik :570-573, entropy :560) ; second pass only if Z[:,k] += 1e-8 fires
        double tot[CMAX + 3];
        bool mod = false;
        for (int pass = 0; pass < 2; ++pass) {
            double wsum[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) wsum[c] = 0.0;
            double sumk = 0.0, ell = 0.0, ent = 0.0;
            for (int n = tid; n < Np; n += EM_THREADS) {
                double vkn = 0.0;
                if (n < N) {
                    const double cn = cu[n];
                    double lz[CMAX], z[CMAX];
                    double mx = 0.0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if (c < C) {
                            lz[c] = (c < K) ? slw[c] + Mu[roff[c] + n] : slw[c] + d.unif_ll;
                            mx = (c == 0 || lz[c] > mx) ? lz[c] : mx;
                        } else {
                            lz[c] = 0.0;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) z[c] = (c < C) ? exp((lz[c] - mx) * cn) : 0.0;
                    const double s = d_np_sum<CMAX>(z, C);
                    double zk = 0.0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        z[c] = z[c] / s;
                        if (c == k) {
                            sumk += z[c];
                            if (mod) z[c] += 1e-8;
                            zk = z[c];
                        }
                    }
                    const double s2 = d_np_sum<CMAX>(z, C);
                    double e[CMAX];
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if (c < C) {
                            wsum[c] += cn * z[c];
                            if (z[c] != 0.0) ell += (z[c] * cn) * lz[c];
                            const double pk = z[c] / s2;
                            e[c] = (pk > 0.0) ? -pk * log(pk) : 0.0;
                        } else {
                            e[c] = 0.0;
                        }
                    }
                    ent += cn * d_np_sum<CMAX>(e, C);
                    vkn = zk * cn;
                }
                vk[n] = vkn;
            }
            // block reduction (fixed order: lane tree, then waves 0..3)
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const double v = d_wave_sum(wsum[c]);
                if (lane == 0) s_red[wave][c] = v;
            }
            {
                double v = d_wave_sum(sumk);
                if (lane == 0) s_red[wave][CMAX] = v;
                v = d_wave_sum(ell);
                if (lane == 0) s_red[wave][CMAX + 1] = v;
                v = d_wave_sum(ent);
                if (lane == 0) s_red[wave][CMAX + 2] = v;
            }
            __syncthreads();
            if (tid < CMAX + 3) {
                double v = 0.0;
                for (int w = 0; w < EM_WAVES; ++w) v += s_red[w][tid];
                s_tot[tid] = v;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CMAX + 3; ++c) tot[c] = s_tot[c];
            if (pass == 0 && tot[CMAX] < 1e-8) {
                mod = true;  // redo the pass with Z[:,k] += 1e-8 (apa_core.py:528-529)
                __syncthreads();
                continue;
            }
            break;
        }

        // ---- maximize_ws (:498-505), all threads redundantly (uniform) ----------------------
        double wn[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) wn[c] = (c < C) ? tot[c] : 0.0;
        {
            const double s = d_np_sum<CMAX>(wn, C);
#pragma unroll
            for (int c = 0; c < CMAX; ++c) wn[c] = wn[c] / s;
            double wK = 0.0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c == K) wK = wn[c];
            if (wK > P.max_unif_ws) {
                const double sk = d_np_sum<CMAX>(wn, K);
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    if (c < K) wn[c] = (1 - P.max_unif_ws) * wn[c] / sk;
                    if (c == K) wn[c] = P.max_unif_ws;
                }
            }
        }
        double wk_new = 0.0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c == k) wk_new = wn[c];
        if (tid < CMAX) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c == tid && c < C) s_ws[c] = wn[c];
        }

        // ---- max_alpha_beta (:507-523): first arg-max over the inclusive theta window --------
        if (!fixed && k < K) {
            const int lo = (k == 0) ? 0 : s_ia[k - 1];
            const int hi = (k == K - 1) ? d.T - 1 : s_ia[k + 1];
            const double lw = d_logw(wk_new);
            const int r0 = lo * B, r1 = (hi + 1) * B;
            double best = -INFINITY;
            int best_r = r0;
            const int nq = Np >> 1;
            const double2 *vk2 = reinterpret_cast<const double2 *>(vk);
            for (int rr = r0 + wave; rr < r1; rr += EM_WAVES) {
                const double2 *row = reinterpret_cast<const double2 *>(Mu + (size_t)rr * Np);
                double acc0 = 0.0, acc1 = 0.0;
                for (int q = lane; q < nq; q += 64) {
                    const double2 m = row[q];
                    const double2 v = vk2[q];
                    acc0 += (lw + m.x) * v.x;
                    acc1 += (lw + m.y) * v.y;
                }
                const double sc = d_wave_sum(acc0 + acc1);
                if (lane == 0 && sc > best) {
                    best = sc;
                    best_r = rr;
                }
            }
            if (lane == 0) {
                s_best[wave] = best;
                s_bestr[wave] = best_r;
            }
            slab_rows += (unsigned long long)(r1 - r0);
            __syncthreads();
            if (tid == 0) {
                double bb = s_best[0];
                int br = s_bestr[0];
                for (int w = 1; w < EM_WAVES; ++w)
                    if (s_best[w] > bb || (s_best[w] == bb && s_bestr[w] < br)) {
                        bb = s_best[w];
                        br = s_bestr[w];
                    }
                if (lo <= hi) {
                    s_ia[k] = br / B;
                    s_ib[k] = br - (br / B) * B;
                }
            }
        }

        // ---- elbo (:559-561) and the stopping rule (:743-746), uniform across threads --------
        const double lb_new = tot[CMAX + 1] + tot[CMAX + 2];
        ell_last = tot[CMAX + 1];
        if (tid == 0) lb_arr[n_lb] = lb_new;
        ++n_lb;
        const bool stop = fabs(lb_new - lb) < fabs(1e-6 * lb);
        lb = lb_new;
        __syncthreads();
        if (stop) break;
    }

    // ---- cal_bic (:702-706), sort by alpha (:768-772), outputs -------------------------------
    if (tid == 0) {
        bic_out[job] = -2 * ell_last + (3 * K + 1) * log((double)N);
        nlb_out[job] = n_lb;
        int idx[CMAX];
        for (int i = 0; i < K; ++i) idx[i] = i;
        for (int i = 1; i < K; ++i) {  // stable insertion sort == numpy argsort for K <= 16
            const int v = idx[i];
            int j = i - 1;
            while (j >= 0 && s_ia[idx[j]] > s_ia[v]) {
                idx[j + 1] = idx[j];
                --j;
            }
            idx[j + 1] = v;
        }
        for (int i = 0; i < K; ++i) {
            a_out[(size_t)job * kmax + i] = s_ia[idx[i]];
            b_out[(size_t)job * kmax + i] = s_ib[idx[i]];
            ws_out[(size_t)job * (kmax + 1) + i] = s_ws[idx[i]];
        }
        ws_out[(size_t)job * (kmax + 1) + K] = s_ws[K];
        for (int i = K; i < kmax; ++i) {
            a_out[(size_t)job * kmax + i] = -1;
            b_out[(size_t)job * kmax + i] = -1;
            ws_out[(size_t)job * (kmax + 1) + i + 1] = 0.0;
        }
        atomicAdd(&counters[0], (unsigned long long)n_lb);
        atomicAdd(&counters[1], slab_rows * (unsigned long long)N);
        atomicAdd(&counters[2], (unsigned long long)n_lb * (unsigned long long)N * (unsigned long long)C);
    }
}

#endif

#include "phase_b_split.inc"
#include "em_lockstep.inc"
#include "mstep_ring.inc"
#include "mstep_multi.inc"

// ------------------------------------------------------------------------------------------
// labels: get_label (apa_core.py:873-881), one workgroup per selected model
// ------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void k_labels(const UtrDesc *__restrict__ descs, DevParams P,
                                                const double *__restrict__ cnt,
                                                const double *__restrict__ M, int kmax,
                                                const int32_t *__restrict__ sel_utr,
                                                const int32_t *__restrict__ sel_K,
                                                const int32_t *__restrict__ a_in,
                                                const int32_t *__restrict__ b_in,
                                                const double *__restrict__ ws_in,
                                                int32_t *__restrict__ labels) {
    __shared__ double2 s_exptab[128];
    d_load_exptab(s_exptab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int sel = blockIdx.x;
    const UtrDesc d = descs[sel_utr[sel]];
    const int K = sel_K[sel], C = K + 1, B = P.B;
    const double *Mu = M + (size_t)d.m_off;
    size_t roff[CMAX];
    double slw[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        roff[c] = (c < K) ? ((size_t)a_in[(size_t)sel * kmax + c] * B + b_in[(size_t)sel * kmax + c]) * d.Np : 0;
        slw[c] = (c < C) ? d_logw(ws_in[(size_t)sel * (kmax + 1) + c]) : 0.0;
    }
    for (int n = threadIdx.x; n < d.N; n += blockDim.x) {
        const double cn = cnt[d.bin_off + n];
        double lz[CMAX], z[CMAX], mx = 0.0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                lz[c] = (c < K) ? slw[c] + Mu[roff[c] + n] : slw[c] + d.unif_ll;
                mx = (c == 0 || lz[c] > mx) ? lz[c] : mx;
            } else {
                lz[c] = 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) z[c] = (c < C) ? d_exp_nonpos((lz[c] - mx) * cn, s_exptab) : 0.0;
        const double s = d_np_sum<CMAX>(z, C);
        int best = 0;
        double bz = 0.0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const double zc = z[c] / s;
            if (c < C && (c == 0 || zc > bz)) {
                bz = zc;
                best = c;
            }
        }
        labels[d.bin_off + n] = best;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap && p) return 0;
        // a buffer that has to grow gets 1/8 headroom: successive batches of a stream differ by a few per cent in
        // size, and re-allocating tens of GB for each of them costs more than the batch itself
        // (large buffers get it from the start: device memory costs ~25 ms per GB to allocate, and the second batch of a
        // stream is a few per cent larger than the first as often as not)
        // - not the very large ones: engines that share a device size their batches to a fraction of what is free, and
        //   12 % on top of each pushed two worker processes on one GPU over the edge (one of them ran 3x slower)
        const bool regrow = p != nullptr || (bytes > ((size_t)1 << 30) && bytes <= ((size_t)48 << 30));
        if (p) {
            (void)hipFree(p);
            p = nullptr;
            cap = 0;
        }
        if (bytes == 0) bytes = 16;
        size_t want = regrow ? bytes + bytes / 8 : bytes;
        if (hipMalloc(&p, want) != hipSuccess) {     // no room for the headroom: exact size
            (void)hipGetLastError();
            p = nullptr;
            want = bytes;
            HIPCHK(hipMalloc(&p, want));
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

struct EventPair {
    hipEvent_t a, b;
};

#define N_COUNTERS (4 + 5 * 64 + 16 + 48)   // (+16: ESTEP_STAMPS, +48: M4_STAMPS tools builds) rounds, slab elements (v1), z elements, unused, then five 64-way sharded counters (em_lockstep.inc)
struct scape_hip_ctx {
    int device = 0;
    std::atomic<bool> busy{false};   // a handle serves one host thread at a time (scape_hip.h)
    hipStream_t stream = nullptr, stream2 = nullptr;
    char name[256] = {0};
    DevParams prm;
    bool loaded = false, built = false;
    bool build_unchecked = false;     // Phase A/B are queued, their device-side consistency flag has not been read yet
    int n_utr = 0, T_max = 0, Np_max = 0, W_max = 1;
    int64_t n_bins = 0;
    std::vector<UtrDesc> h_desc;
    size_t at_total = 0, m_total = 0;
    DevBuf d_x, d_l, d_r, d_pa, d_cnt, d_theta, d_desc, d_loglist, d_AT, d_V, d_M, d_err, d_counters, d_tile_nend;
    DevBuf d_tbi, d_tbG, d_tbpm, d_mlog, d_logbin;   // Phase B: window tables, compact log-domain values, log bin -> UTR
    size_t n_theta_total = 0, mlog_total = 0, n_logbins = 0;
    int pb_form = 0;                  // Phase B form of the last batch_build (scape_hip_batch_phase_b_form)
    bool pb_slide = false;            // every UTR of the batch has a uniform theta grid with interior points: k_pb_logcol / k_pb_slide
    int tiles_max_all = 1;
    size_t tiles_total = 0;
    DevBuf j_utr, j_K, j_fixed, j_a, j_b, j_ws, j_karr, j_ao, j_bo, j_wso, j_bic, j_nlb, j_lb, j_sel, j_lbsel;
    int last_em_jobs = 0;     // jobs of the last completed scape_hip_batch_em call (scape_hip_batch_em_fetch_lb)
    DevBuf l_utr, l_K, l_a, l_b, l_ws, l_labels;
    // lock-step EM state (em_lockstep.inc)
    DevBuf e_ia, e_ib, e_sia, e_sib, e_ws, e_slw, e_lb, e_ell, e_nlb, e_status, e_rdk, e_rdlo, e_rdhi, e_rdm,
        e_rdlw, e_rdsv, e_rdn0, e_rdn1, e_V, e_Vsuf, e_voff, e_ptscore, e_ptrow, e_ptoff, e_ujoff, e_ujlist, e_active, e_elist;
    std::vector<EventPair> ev[6];
    double ms_acc[6] = {0, 0, 0, 0, 0, 0};
    int n_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long h_counters[3] = {0, 0, 0};
    unsigned long long h_traffic[4] = {0, 0, 0, 0};   // last EM call: M-step tensor bytes, v bytes requested / unique, launches
};

static int ev_begin(scape_hip_ctx *c, int which) {
    EventPair e;
    HIPCHK(hipEventCreate(&e.a));
    HIPCHK(hipEventCreate(&e.b));
    HIPCHK(hipEventRecord(e.a, c->stream));
    c->ev[which].push_back(e);
    return 0;
}
static int ev_end(scape_hip_ctx *c, int which) {
    HIPCHK(hipEventRecord(c->ev[which].back().b, c->stream));
    return 0;
}
static int ev_collect(scape_hip_ctx *c) {
    HIPCHK(hipStreamSynchronize(c->stream));
    const bool trace = getenv("SCAPE_HIP_ROUND_TRACE") != nullptr;   // per-launch durations on stderr
    for (int w = 0; w < 6; ++w) {
        int i = 0;
        for (auto &e : c->ev[w]) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            if (trace && w >= 4) fprintf(stderr, "[trace] kind %d launch %d %.4f ms\n", w, i, ms);
            ++i;
            c->ms_acc[w] += ms;
            c->n_acc[w] += 1;
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        c->ev[w].clear();
    }
    return 0;
}

static int round_up(int v, int g) { return (v + g - 1) / g * g; }

static int set_device(scape_hip_ctx *c) {
    HIPCHK(hipSetDevice(c->device));
    return 0;
}

// Calls on one handle are not re-entrant: a second host thread entering while a call is in flight gets an
// error instead of racing on the handle's device buffers.
struct BusyGuard {
    scape_hip_ctx *c;
    bool ok;
    explicit BusyGuard(scape_hip_ctx *ctx) : c(ctx), ok(ctx && !ctx->busy.exchange(true)) {}
    ~BusyGuard() { if (ok) c->busy.store(false); }
};
#define CTX_GUARD(c)                                                                                      \
    if (!(c)) return fail("ctx is NULL");                                                                 \
    BusyGuard busy_guard_(c);                                                                             \
    if (!busy_guard_.ok) return fail("handle is in use by another host thread (calls on a handle are not re-entrant)")
#define CTX_ENTER(c)                                                                                      \
    CTX_GUARD(c);                                                                                         \
    if (set_device(c)) return 1

static void fill_params(DevParams &d, double mu_f, double sigma_f, double max_unif_ws, int B,
                        const double *betas, int S, const double *s, const double *pmf, int nround) {
    memset(&d, 0, sizeof(d));
    d.mu_f = mu_f;
    d.sigma_f = sigma_f;
    d.max_unif_ws = max_unif_ws;
    d.B = B;
    d.S = S;
    d.nround = nround;
    for (int i = 0; i < B; ++i) d.betas[i] = betas[i];
    for (int i = 0; i < S; ++i) {
        d.s_dis[i] = s[i];
        d.pmf_s[i] = pmf[i];
        d.inv_s[i] = 1 / s[i];
    }
}

// widest +-3*beta window (in grid points) over all grid points of one theta grid
static int max_window(const double *th, int T, double beta) {
    int best = 1, a = 0, b = 0;
    for (int i = 0; i < T; ++i) {
        const double lo = th[i] - 3 * beta, hi = th[i] + 3 * beta;
        while (a < i && th[a] < lo) ++a;
        if (b < i) b = i;
        while (b + 1 < T && th[b + 1] <= hi) ++b;
        best = std::max(best, b - a + 1);
    }
    return best;
}

static int launch_phase_b(scape_hip_ctx *c, const DevParams &prm, int n_utr, int T_max, int Wmax,
                          const UtrDesc *d_desc, const double *d_r, const double *d_pa,
                          const double *d_theta, const int32_t *d_loglist, const double *d_AT,
                          const double *d_V, double *d_M, int all_log, int32_t *d_tile_nend) {
    if (c->d_err.ensure(sizeof(int))) return 1;
    HIPCHK(hipMemsetAsync(c->d_err.p, 0, sizeof(int), c->stream));
    const size_t lds = ((size_t)2 * prm.B * Wmax + prm.B) * sizeof(double) + (size_t)2 * (prm.B + 1) * sizeof(int) +
                       (size_t)16 * (((Wmax + 3) & ~3) + 1) * sizeof(double);
    if (lds > 150 * 1024) return fail("Phase B: window table does not fit LDS (n_beta x window too large)");
    dim3 grid((unsigned)(((n_utr + 7) / 8) * 8 * T_max));
#ifdef SCAPE_HIP_TOOLS   // tools-only builds (-DSCAPE_HIP_TOOLS): timing probe that switches parts of Phase B off - results are wrong with it
    const char *pe = getenv("SCAPE_HIP_PB_PROBE");   // 1 no log path, 2 no linear path, 4 no G exps
    const int probe = pe ? atoi(pe) : 0;
#else
    const int probe = 0;
#endif
    if (prm.B <= 16)
        hipLaunchKernelGGL(k_phase_b<16>, grid, dim3(256), lds, c->stream, d_desc, prm, d_r, d_pa, d_theta,
                           d_loglist, d_AT, d_V, d_M, Wmax, all_log, c->d_err.as<int>(), n_utr, T_max, d_tile_nend, probe);
    else
        hipLaunchKernelGGL(k_phase_b<1>, grid, dim3(256), lds, c->stream, d_desc, prm, d_r, d_pa, d_theta,
                           d_loglist, d_AT, d_V, d_M, Wmax, all_log, c->d_err.as<int>(), n_utr, T_max, d_tile_nend, probe);
    HIPCHK(hipGetLastError());
    return 0;
}

static int check_err_flag(scape_hip_ctx *c, const char *what);
// reads the consistency flag of a queued batch_build (synchronises the stream)
static int finish_build(scape_hip_ctx *c) {
    if (!c->build_unchecked) return 0;
    c->build_unchecked = false;
    if (check_err_flag(c, "batch_build")) {
        c->built = false;
        return 1;
    }
    return 0;
}

static int check_err_flag(scape_hip_ctx *c, const char *what) {
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, c->d_err.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (flag) return fail(std::string(what) + ": device-side consistency check failed (flag " + std::to_string(flag) + ")");
    return 0;
}

// host driver of the lock-step EM (kernels in em_lockstep.inc); job tables are already on the device
static int em_lockstep(scape_hip_ctx *c, int n_jobs, int kmax, const int32_t *job_utr, const int32_t *job_fixed) {
    const size_t nj = n_jobs;
    const int nround = c->prm.nround, B = c->prm.B;
    // host-side index tables: jobs grouped by UTR, ragged offsets of V and of the per-tile partials
    std::vector<int64_t> ujoff(c->n_utr + 1, 0), voff(nj), ptoff(nj);
    std::vector<int32_t> ujlist(nj);
    for (size_t j = 0; j < nj; ++j) ujoff[job_utr[j] + 1]++;
    for (int u = 0; u < c->n_utr; ++u) ujoff[u + 1] += ujoff[u];
    {
        std::vector<int64_t> fill(ujoff.begin(), ujoff.end() - 1);
        for (size_t j = 0; j < nj; ++j) ujlist[fill[job_utr[j]]++] = (int32_t)j;
    }
    // UTRs that have at least one job in this call, in index order: the M-step grid runs over this list, so a call
    // that touches a few UTRs of a large resident batch (reference-stream mode) still uses every XCD
    std::vector<int32_t> active;
    for (int u = 0; u < c->n_utr; ++u)
        if (ujoff[u + 1] > ujoff[u]) active.push_back(u);
    // Blocks are dealt round-robin over the 8 XCDs and the M-step gives XCD x the UTRs at positions x, x + 8, ... of this
    // list: sorted by tensor size, the 8 UTRs that are in flight side by side are of one size class (their tiles take
    // alike) and every XCD gets the same mix - in index order an XCD that drew the large UTRs finishes last while
    // the others idle.  Placement only: a UTR's results do not depend on where or next to what it runs.
    const bool size_order = !getenv("SCAPE_HIP_NO_SIZE_ORDER") && !getenv("SCAPE_HIP_TWO_STREAMS");
    if (size_order)
        std::stable_sort(active.begin(), active.end(), [&](int32_t a, int32_t b) {
            const UtrDesc &da = c->h_desc[a], &db = c->h_desc[b];
            return (int64_t)da.T * da.Np > (int64_t)db.T * db.Np;
        });
    const int n_active = (int)active.size();
    int max_jobs_utr = 1;
    for (int u : active) max_jobs_utr = std::max<int>(max_jobs_utr, (int)(ujoff[u + 1] - ujoff[u]));
    int tiles_max = 1;
    for (int u : active) tiles_max = std::max(tiles_max, (c->h_desc[u].T * B + MT_ROWS - 1) / MT_ROWS);
    // Small calls (the ~100 jobs of one UTR, or of a few streams' current UTRs) leave most of the GPU idle and an
    // M-step round takes as long as ONE workgroup needs for its pass over a tile: the jobs of a tile are then cut
    // into passes of 16 that separate workgroups take (k2_mstep).  Every score is the same MFMA chain either way,
    // so a UTR's result does not depend on the size of the call it is part of.
    const char *env_r = getenv("SCAPE_HIP_SPLIT_MAXTILES"), *env_w = getenv("SCAPE_HIP_WIDE_MAXJOBS");
    const int split_maxtiles = env_r ? atoi(env_r) : 1024, wide_maxjobs = env_w ? atoi(env_w) : 1024;
    const bool wide = n_jobs <= wide_maxjobs && kmax + 1 <= 16;   // E-step: 4 wavefronts per job (k2_estep_cs)
    const bool job_split = (long long)n_active * tiles_max <= split_maxtiles;
    // M-step kernel: k3_mstep (LDS-DMA rings, mstep_ring.inc) unless SCAPE_HIP_MSTEP=v2 asks for the register-staged
    // k2_mstep (A/B runs; identical bits).  k3's DMA source offsets are 32-bit byte offsets from the start of the job vectors.
    const bool debug_env = getenv("SCAPE_HIP_DEBUG") != nullptr;     // the tile / job histograms are k2 / k3 only
    const char *env_m = getenv("SCAPE_HIP_MSTEP");
    bool ring_mstep = !(env_m && strcmp(env_m, "v2") == 0);
    constexpr int pt_rows = MT_ROWS;
    size_t vtot = 0, pttot = 0;
    for (size_t j = 0; j < nj; ++j) {
        const UtrDesc &d = c->h_desc[job_utr[j]];
        const int nt = (d.T * B + pt_rows - 1) / pt_rows;
        voff[j] = (int64_t)vtot;
        ptoff[j] = (int64_t)pttot;
        vtot += (size_t)d.Np;
        pttot += (size_t)nt;
    }
    if (c->e_ia.ensure(nj * kmax * 4) || c->e_ib.ensure(nj * kmax * 4) || c->e_sia.ensure(nj * kmax * 4) ||
        c->e_sib.ensure(nj * kmax * 4) || c->e_ws.ensure(nj * (kmax + 1) * 8) || c->e_slw.ensure(nj * (kmax + 1) * 8) ||
        c->e_lb.ensure(nj * 8) || c->e_ell.ensure(nj * 8) || c->e_nlb.ensure(nj * 4) || c->e_status.ensure(nj * 4) ||
        c->e_rdk.ensure(nj * 4) || c->e_rdlo.ensure(nj * 4) || c->e_rdhi.ensure(nj * 4) || c->e_rdm.ensure(nj * 4) ||
        c->e_rdlw.ensure(nj * 8) || c->e_rdsv.ensure(nj * 8) || c->e_rdn0.ensure(nj * 4) || c->e_rdn1.ensure(nj * 4) || c->e_V.ensure(vtot * 8) || c->e_Vsuf.ensure((vtot / 16 + nj + 1) * 8) || c->e_voff.ensure(nj * 8) ||
        c->e_ptscore.ensure(pttot * 8) || c->e_ptrow.ensure(pttot * 4) || c->e_ptoff.ensure(nj * 8) ||
        c->e_ujoff.ensure((c->n_utr + 1) * 8) || c->e_ujlist.ensure(nj * 4) || c->e_active.ensure((size_t)c->n_utr * 4))
        return 1;
    if (vtot * 8 >= ((size_t)1 << 31)) ring_mstep = false;
    // k4_mstep (mstep_multi.inc): several tiles per workgroup, set-up and epilogue off the critical path - the kernel of
    // wave-sized calls; calls with few live tiles keep k3_mstep with a tile's jobs cut into passes for several workgroups
    const bool multi_mstep = ring_mstep && !job_split && !debug_env && pttot < ((size_t)1 << 31) && !(env_m && strcmp(env_m, "v3") == 0);
    if (multi_mstep && M4_LDS_BYTES > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k4_mstep), hipFuncAttributeMaxDynamicSharedMemorySize, M4_LDS_BYTES));
    if (ring_mstep && M3_LDS_BYTES > 64 * 1024)   // per device; a few microseconds
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k3_mstep), hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS_BYTES));
    // The E-step gives XCD x the x-th eighth of its job list: with the jobs of UTR active[x], active[x + 8], ... there,
    // every XCD gets the same mix of sizes, and the v vectors a UTR's jobs write are read by the M-step tiles of the same
    // UTR on the same XCD (L2; speed only).
    std::vector<int32_t> elist;
    if (size_order && n_active >= 8) {
        elist.reserve(nj);
        for (int x = 0; x < 8; ++x)
            for (int k = x; k < n_active; k += 8)
                for (int64_t q = ujoff[active[k]]; q < ujoff[active[k] + 1]; ++q) elist.push_back(ujlist[q]);
        if (c->e_elist.ensure(nj * 4)) return 1;
        HIPCHK(hipMemcpyAsync(c->e_elist.p, elist.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (n_active) HIPCHK(hipMemcpyAsync(c->e_active.p, active.data(), (size_t)n_active * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->e_voff.p, voff.data(), nj * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->e_ptoff.p, ptoff.data(), nj * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->e_ujoff.p, ujoff.data(), (c->n_utr + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->e_ujlist.p, ujlist.data(), nj * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // the host vectors go out of scope
    EmState S;
    S.ia = c->e_ia.as<int32_t>();
    S.ib = c->e_ib.as<int32_t>();
    S.sia = c->e_sia.as<int32_t>();
    S.sib = c->e_sib.as<int32_t>();
    S.ws = c->e_ws.as<double>();
    S.slw = c->e_slw.as<double>();
    S.lb = c->e_lb.as<double>();
    S.ell = c->e_ell.as<double>();
    S.nlb = c->e_nlb.as<int32_t>();
    S.status = c->e_status.as<int32_t>();
    S.rd_k = c->e_rdk.as<int32_t>();
    S.rd_lo = c->e_rdlo.as<int32_t>();
    S.rd_hi = c->e_rdhi.as<int32_t>();
    S.rd_m = c->e_rdm.as<int32_t>();
    S.rd_lw = c->e_rdlw.as<double>();
    S.rd_sv = c->e_rdsv.as<double>();
    S.rd_n0 = c->e_rdn0.as<int32_t>();
    S.rd_n1 = c->e_rdn1.as<int32_t>();
    S.V = c->e_V.as<double>();
    S.Vsuf = c->e_Vsuf.as<double>();
    S.voff = c->e_voff.as<int64_t>();
    S.pt_score = c->e_ptscore.as<double>();
    S.pt_row = c->e_ptrow.as<int32_t>();
    S.ptoff = c->e_ptoff.as<int64_t>();
    unsigned long long *dbg = nullptr;
    const bool debug = getenv("SCAPE_HIP_DEBUG") != nullptr;
    if (debug) {
        HIPCHK(hipMalloc((void **)&dbg, 24 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(dbg, 0, 24 * sizeof(unsigned long long), c->stream));
    }
    const bool fine = getenv("SCAPE_HIP_ROUND_TIMING") != nullptr;
    bool any_m = false;   // fixed-inference jobs (mstep_fixed) have no grid arg-max
    for (size_t j = 0; j < nj && !any_m; ++j) any_m = job_fixed[j] == 0;
    // Two halves of the UTRs advance on two streams: the E-step is f64-VALU work, the M-step an HBM stream,
    // so one half's E-step overlaps the other half's M-step.  (Per-kernel event timing and the debug
    // counters use a single stream.)
    // Measured (512 UTRs x 2k reads): no gain - the M-step needs 256 VGPRs x 2 waves/SIMD to stream at full
    // rate, which leaves no room for E-step waves on the same SIMD - so it is off unless requested.
    const bool split = any_m && !fine && !debug && c->n_utr >= 16 && n_jobs >= 2048 && getenv("SCAPE_HIP_TWO_STREAMS");
    const int n_grp = split ? 2 : 1;
    // groups are halves of the ACTIVE list; g_u0 holds the UTR index bounds of each half for the job ranges
    const int a_half = split ? n_active / 2 : n_active;
    int g_a0[3] = {0, a_half, n_active};
    int g_u0[3] = {0, (split && a_half < n_active) ? active[a_half] : c->n_utr, c->n_utr};
    hipStream_t g_stream[2] = {c->stream, c->stream2};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (split) {
        HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_fork, c->stream));
        HIPCHK(hipStreamWaitEvent(c->stream2, ev_fork, 0));
    }
    if (!any_m) {
        // only fixed-inference jobs (the ws-only re-fits of rm_component): all rounds in one launch
        const int ngj = (int)nj;
        const int32_t *jl = c->e_ujlist.as<int32_t>();
        if (fine && ev_begin(c, 4)) return 1;
#define LAUNCH_ALL(CM)                                                                                         \
    hipLaunchKernelGGL(k2_estep_all_rounds<CM>, dim3((unsigned)(((ngj + 7) / 8) * 8)), dim3(64), 0, c->stream, \
                       c->d_desc.as<UtrDesc>(), c->prm, c->d_cnt.as<double>(), c->d_M.as<double>(), kmax,      \
                       c->j_utr.as<int32_t>(), c->j_K.as<int32_t>(), c->j_fixed.as<int32_t>(),                 \
                       c->j_a.as<int32_t>(), c->j_b.as<int32_t>(), c->j_ws.as<double>(), c->j_karr.as<int8_t>(), \
                       S, c->j_ao.as<int32_t>(), c->j_bo.as<int32_t>(), c->j_wso.as<double>(),                 \
                       c->j_bic.as<double>(), c->j_nlb.as<int32_t>(), c->j_lb.as<double>(),                    \
                       c->d_counters.as<unsigned long long>(), jl, ngj)
        if (kmax + 1 <= 4) LAUNCH_ALL(4);
        else if (kmax + 1 <= 8) LAUNCH_ALL(8);
        else if (kmax + 1 <= 12) LAUNCH_ALL(12);
        else if (kmax + 1 <= 16) LAUNCH_ALL(16);
        else if (kmax + 1 <= 24) LAUNCH_ALL(24);
        else if (kmax + 1 <= 32) LAUNCH_ALL(32);
        else LAUNCH_ALL(64);
#undef LAUNCH_ALL
        HIPCHK(hipGetLastError());
        if (fine && ev_end(c, 4)) return 1;
        if (dbg) (void)hipFree(dbg);
        return 0;
    }
    unsigned long long executed_prev = 0;
    const int probe_mask = (n_jobs >= 8192) ? 7 : 3;
    int rc = 0;
    for (int r = 0; r <= nround && !rc; ++r) {
        for (int g = 0; g < n_grp && !rc; ++g) {
            const int u0 = g_u0[g], u1 = g_u0[g + 1];
            const int64_t j0 = ujoff[u0], j1 = ujoff[u1];
            const int ngj = (int)(j1 - j0);
            if (ngj == 0) continue;
            hipStream_t st = g_stream[g];
            const int32_t *jl = elist.empty() ? c->e_ujlist.as<int32_t>() + j0 : c->e_elist.as<int32_t>();
            if (fine && ev_begin(c, 4)) return 1;
#define LAUNCH_E(KERNEL, CM, THREADS)                                                                          \
    hipLaunchKernelGGL(KERNEL<CM>, dim3((unsigned)(((ngj + 7) / 8) * 8)), dim3(THREADS), 0, st,                \
                       c->d_desc.as<UtrDesc>(), c->prm, c->d_cnt.as<double>(), c->d_M.as<double>(), kmax,      \
                       c->j_utr.as<int32_t>(), c->j_K.as<int32_t>(), c->j_fixed.as<int32_t>(),                 \
                       c->j_a.as<int32_t>(), c->j_b.as<int32_t>(), c->j_ws.as<double>(), c->j_karr.as<int8_t>(), \
                       S, c->j_ao.as<int32_t>(), c->j_bo.as<int32_t>(), c->j_wso.as<double>(),                 \
                       c->j_bic.as<double>(), c->j_nlb.as<int32_t>(), c->j_lb.as<double>(),                    \
                       c->d_counters.as<unsigned long long>(), r, jl, ngj)
            if (wide) {
                if (kmax + 1 <= 4) LAUNCH_E(k2_estep_cs, 4, 256);
                else if (kmax + 1 <= 8) LAUNCH_E(k2_estep_cs, 8, 256);
                else if (kmax + 1 <= 12) LAUNCH_E(k2_estep_cs, 12, 256);
                else LAUNCH_E(k2_estep_cs, 16, 256);
            } else if (kmax + 1 <= 4) LAUNCH_E(k2_estep, 4, 64);
            else if (kmax + 1 <= 8) LAUNCH_E(k2_estep, 8, 64);
            else if (kmax + 1 <= 12) LAUNCH_E(k2_estep, 12, 64);
            else if (kmax + 1 <= 16) LAUNCH_E(k2_estep, 16, 64);
            else if (kmax + 1 <= 24) LAUNCH_E(k2_estep, 24, 64);
            else if (kmax + 1 <= 32) LAUNCH_E(k2_estep, 32, 64);
            else LAUNCH_E(k2_estep, 64, 64);      // K up to 63: the column arrays live in scratch there - slow, and rare (a re-run loop that keeps growing, apa_core.py:1023-1030)
#undef LAUNCH_E
            HIPCHK(hipGetLastError());
            if (fine && ev_end(c, 4)) return 1;
            if (r < nround && any_m) {
                if (fine && ev_begin(c, 5)) return 1;
                const int nu = g_a0[g + 1] - g_a0[g];
                // few live tiles: passes of 16 jobs, one workgroup each (gridDim.y covers the longest job list of a UTR)
                const int jobs_per_pass = job_split ? 16 : MT_MAXJ;
                const unsigned psplit = job_split ? (unsigned)std::min(64, (max_jobs_utr + 15) / 16) : 1u;
#define MSTEP_ARGS c->d_desc.as<UtrDesc>(), c->prm, c->d_M.as<double>(), c->e_active.as<int32_t>() + g_a0[g], nu, tiles_max,           \
                   c->e_ujoff.as<int64_t>(), c->e_ujlist.as<int32_t>(), S.V, S.Vsuf, S.voff, S.rd_m, S.rd_lo, S.rd_hi, S.rd_lw,  \
                   S.rd_sv, S.rd_n0, S.rd_n1, S.ptoff, S.pt_score, S.pt_row, c->d_tile_nend.as<int32_t>(),                    \
                   c->d_counters.as<unsigned long long>(), jobs_per_pass, dbg
                if (multi_mstep) {
                    const int groups_max = (tiles_max + M4_TPW - 1) / M4_TPW;
                    hipLaunchKernelGGL(k4_mstep, dim3((unsigned)(((nu + 7) / 8) * 8 * groups_max)), dim3(64 * M4_NW), M4_LDS_BYTES, st,
                                       c->d_desc.as<UtrDesc>(), c->prm, c->d_M.as<double>(), c->e_active.as<int32_t>() + g_a0[g], nu, groups_max,
                                       c->e_ujoff.as<int64_t>(), c->e_ujlist.as<int32_t>(), S.V, S.Vsuf, S.voff, S.rd_m, S.rd_lo, S.rd_hi, S.rd_lw,
                                       S.rd_sv, S.rd_n0, S.rd_n1, S.ptoff, S.pt_score, S.pt_row, c->d_tile_nend.as<int32_t>(),
                                       c->d_counters.as<unsigned long long>());
                } else if (ring_mstep)
                    hipLaunchKernelGGL(k3_mstep, dim3((unsigned)(((nu + 7) / 8) * 8 * tiles_max), psplit), dim3(256), M3_LDS_BYTES, st, MSTEP_ARGS);
                else
                    hipLaunchKernelGGL(k2_mstep, dim3((unsigned)(((nu + 7) / 8) * 8 * tiles_max), psplit), dim3(256), 0, st, MSTEP_ARGS);
#undef MSTEP_ARGS
                c->h_traffic[3] += 1;
                HIPCHK(hipGetLastError());
                if (fine && ev_end(c, 5)) return 1;
            }
        }
        // early exit: no job executed a round since the last probe -> every job has been finalised.  The probe waits for
        // the stream (a bubble of a few tens of microseconds): every 4 rounds for small calls, whose jobs all finish
        // early and whose rounds are short, every 8 for wave-sized ones, which run to the last round anyway
        if (r < nround && (r & probe_mask) == probe_mask) {
            unsigned long long executed = 0, shard[64];
            if (split) HIPCHK(hipStreamSynchronize(c->stream2));
            HIPCHK(hipMemcpyAsync(shard, c->d_counters.as<unsigned long long>() + CNT_EXEC, sizeof(shard),
                                  hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int i = 0; i < 64; ++i) executed += shard[i];
            if (executed == executed_prev) break;
            executed_prev = executed;
        }
        if (debug && r < nround) {      // cumulative tensor bytes the M-step has asked for (its own tally), per round
            unsigned long long shard[64], tb = 0;
            HIPCHK(hipMemcpy(shard, c->d_counters.as<unsigned long long>() + CNT_MSTEP_TENSOR, sizeof(shard), hipMemcpyDeviceToHost));
            for (int i = 0; i < 64; ++i) tb += shard[i];
            fprintf(stderr, "[em bytes %d] %llu\n", r, tb);
        }
        if (debug && (r == 0 || r == 10 || r == 25 || r == nround - 1)) {
            unsigned long long h[24];
            HIPCHK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            // k3_mstep only: k-steps x jobs if every job stopped at its own support / as run (16-job groups over the union)
            fprintf(stderr, "[em round %d] job-halfchunks own-support %llu, union %llu, padded to groups %llu; sorted-by-n1 groups %llu\n", r, h[16], h[17], h[18], h[19]);
            fprintf(stderr, "[em round %d] tiles %llu active %llu sum_cnt %llu passes %llu | G-hist:", r, h[0], h[1], h[2], h[3]);
            for (int i = 4; i < 16; ++i) fprintf(stderr, " %llu", h[i]);
            fprintf(stderr, "\n");
            HIPCHK(hipMemsetAsync(dbg, 0, 24 * sizeof(unsigned long long), c->stream));
        }
    }
    if (split) {
        HIPCHK(hipEventRecord(ev_join, c->stream2));
        HIPCHK(hipStreamWaitEvent(c->stream, ev_join, 0));
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipEventDestroy(ev_fork);
        (void)hipEventDestroy(ev_join);
    }
    if (dbg) (void)hipFree(dbg);
    return 0;
}

extern "C" {

int scape_hip_abi_version(void) { return SCAPE_HIP_ABI_VERSION; }
const char *scape_hip_last_error(void) { return g_err.c_str(); }

int scape_hip_device_count(int *count) {
    if (!count) return fail("count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return 0;
}

int scape_hip_create(int device, scape_hip_ctx **out) {
    if (!out) return fail("out is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("no such HIP device: " + std::to_string(device));
    HIPCHK(hipSetDevice(device));
    scape_hip_ctx *c = new scape_hip_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    snprintf(c->name, sizeof(c->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipStreamCreate(&c->stream2));
    if (c->d_counters.ensure(N_COUNTERS * sizeof(unsigned long long))) return 1;
    *out = c;
    return 0;
}

int scape_hip_device_name(scape_hip_ctx *c, char *buf, int buflen) {
    if (!c || !buf || buflen <= 0) return fail("bad argument");
    snprintf(buf, buflen, "%s", c->name);
    return 0;
}

int scape_hip_batch_free(scape_hip_ctx *c) {
    if (!c) return fail("ctx is NULL");
    CTX_ENTER(c);
    DevBuf *all[] = {&c->d_tbi, &c->d_tbG, &c->d_tbpm, &c->d_mlog, &c->d_logbin, &c->d_tile_nend, &c->d_x, &c->d_l, &c->d_r, &c->d_pa, &c->d_cnt, &c->d_theta, &c->d_desc, &c->d_loglist,
                     &c->d_AT, &c->d_V, &c->d_M, &c->j_utr, &c->j_K, &c->j_fixed, &c->j_a, &c->j_b, &c->j_ws,
                     &c->j_karr, &c->j_ao, &c->j_bo, &c->j_wso, &c->j_bic, &c->j_nlb, &c->j_lb, &c->l_utr,
                     &c->l_K, &c->l_a, &c->l_b, &c->l_ws, &c->l_labels, &c->e_ia, &c->e_ib, &c->e_sia, &c->e_sib,
                     &c->e_ws, &c->e_slw, &c->e_lb, &c->e_ell, &c->e_nlb, &c->e_status, &c->e_rdk, &c->e_rdlo,
                     &c->e_rdhi, &c->e_rdm, &c->e_rdlw, &c->e_rdsv, &c->e_rdn0, &c->e_rdn1, &c->e_V, &c->e_Vsuf, &c->e_voff, &c->e_ptscore,
                     &c->e_ptrow, &c->e_ptoff, &c->e_ujoff, &c->e_ujlist, &c->e_active, &c->e_elist, &c->j_sel, &c->j_lbsel};
    for (DevBuf *b : all) b->release();
    c->last_em_jobs = 0;
    c->loaded = c->built = c->build_unchecked = false;
    c->n_utr = 0;
    return 0;
}

int scape_hip_destroy(scape_hip_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    scape_hip_batch_free(c);
    c->d_err.release();
    c->d_counters.release();
    for (int w = 0; w < 6; ++w)
        for (auto &e : c->ev[w]) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->stream2);
    delete c;
    return 0;
}

// ---- operator level ------------------------------------------------------------------------
static int op_common(scape_hip_ctx *c, int n, const double *const *in, int n_in, double **dev, double **dout) {
    if (!c) return fail("ctx is NULL");
    if (n < 0) return fail("negative length");
    if (set_device(c)) return 1;
    for (int i = 0; i < n_in; ++i) {
        if (!in[i] && n > 0) return fail("NULL input array");
        HIPCHK(hipMalloc((void **)&dev[i], std::max<size_t>(16, (size_t)n * sizeof(double))));
        if (n) HIPCHK(hipMemcpyAsync(dev[i], in[i], (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipMalloc((void **)dout, std::max<size_t>(16, (size_t)n * sizeof(double))));
    return 0;
}
static int op_finish(scape_hip_ctx *c, int n, double **dev, int n_in, double *dout, double *out) {
    int rc = 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = fail(std::string("kernel launch: ") + hipGetErrorString(e));
    if (!rc && n) {
        e = hipMemcpyAsync(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(std::string("copy back: ") + hipGetErrorString(e));
    }
    e = hipStreamSynchronize(c->stream);
    if (!rc && e != hipSuccess) rc = fail(std::string("sync: ") + hipGetErrorString(e));
    for (int i = 0; i < n_in; ++i) (void)hipFree(dev[i]);
    (void)hipFree(dout);
    return rc;
}

int scape_hip_loglik_xlr_t_pa(scape_hip_ctx *c, const double *x, const double *l, const double *pa,
                              int32_t n, double theta, double sigma_f, double *out) {
    CTX_GUARD(c);
    const double *in[3] = {x, l, pa};
    double *dev[3] = {nullptr, nullptr, nullptr}, *dout = nullptr;
    if (op_common(c, n, in, 3, dev, &dout)) return 1;
    if (n) hipLaunchKernelGGL(k_op_pa, dim3((n + 255) / 256), dim3(256), 0, c->stream, dev[0], dev[1], dev[2], n, theta, sigma_f, dout);
    return op_finish(c, n, dev, 3, dout, out);
}

int scape_hip_loglik_xlr_t_r_known(scape_hip_ctx *c, const double *x, const double *l, const double *r,
                                   int32_t n, const double *s_dis, const double *pmf_s, int32_t n_s,
                                   double theta, double mu_f, double sigma_f, double *out) {
    CTX_GUARD(c);
    if (n_s < 1 || n_s > SCAPE_MAX_S) return fail("n_s out of range");
    DevParams P;
    fill_params(P, mu_f, sigma_f, 0, 0, nullptr, n_s, s_dis, pmf_s, 0);
    const double *in[3] = {x, l, r};
    double *dev[3] = {nullptr, nullptr, nullptr}, *dout = nullptr;
    if (op_common(c, n, in, 3, dev, &dout)) return 1;
    if (n) hipLaunchKernelGGL(k_op_r_known, dim3((n + 255) / 256), dim3(256), 0, c->stream, dev[0], dev[1], dev[2], n, P, theta, dout);
    return op_finish(c, n, dev, 3, dout, out);
}

int scape_hip_loglik_xlr_t_r_unknown(scape_hip_ctx *c, const double *x, const double *l, const double *r,
                                     int32_t n, const double *s_dis, const double *pmf_s, int32_t n_s,
                                     double theta, double mu_f, double sigma_f, double *out) {
    CTX_GUARD(c);
    (void)r;
    if (n_s < 1 || n_s > SCAPE_MAX_S) return fail("n_s out of range");
    DevParams P;
    fill_params(P, mu_f, sigma_f, 0, 0, nullptr, n_s, s_dis, pmf_s, 0);
    const double *in[2] = {x, l};
    double *dev[2] = {nullptr, nullptr}, *dout = nullptr;
    if (op_common(c, n, in, 2, dev, &dout)) return 1;
    if (n) hipLaunchKernelGGL(k_op_r_unknown, dim3((n + 255) / 256), dim3(256), 0, c->stream, dev[0], dev[1], n, P, theta, dout);
    return op_finish(c, n, dev, 2, dout, out);
}

int scape_hip_get_loglik_marginal_tensor(scape_hip_ctx *c, const double *all_theta, int32_t T,
                                         const double *betas, int32_t B, const double *A, int32_t N,
                                         double *out) {
    if (!c) return fail("ctx is NULL");
    if (T < 1 || N < 1) return fail("empty theta grid or no fragments");
    if (B < 1 || B > SCAPE_MAX_BETA) return fail("n_beta out of range");
    if (!all_theta || !betas || !A || !out) return fail("NULL argument");
    CTX_ENTER(c);
    if (finish_build(c)) return 1;   // this call reuses the handle's error word: read a queued build's flag first
    for (int i = 1; i < T; ++i)
        if (!(all_theta[i] >= all_theta[i - 1])) return fail("all_theta must be ascending");
    DevParams P;
    fill_params(P, 0, 1, 0, B, betas, 0, nullptr, nullptr, 0);
    const int Np = round_up(N, PITCH);
    double bmax = betas[0];
    for (int j = 1; j < B; ++j) bmax = std::max(bmax, betas[j]);
    const int Wmax = max_window(all_theta, T, bmax);
    std::vector<double> at((size_t)T * Np, 0.0);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t) at[(size_t)t * Np + n] = A[(size_t)n * T + t];
    UtrDesc d;
    memset(&d, 0, sizeof(d));
    d.N = N;
    d.Np = Np;
    d.T = T;
    DevBuf dAT, dM, dth, ddesc;
    int rc = 0;
    auto cleanup = [&]() {
        dAT.release();
        dM.release();
        dth.release();
        ddesc.release();
    };
    if (dAT.ensure(at.size() * sizeof(double)) || dM.ensure((size_t)T * B * Np * sizeof(double)) ||
        dth.ensure((size_t)T * sizeof(double)) || ddesc.ensure(sizeof(UtrDesc))) {
        cleanup();
        return 1;
    }
    hipError_t e;
    e = hipMemcpyAsync(dAT.p, at.data(), at.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dth.p, all_theta, (size_t)T * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ddesc.p, &d, sizeof(d), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) {
        cleanup();
        return fail(std::string("upload: ") + hipGetErrorString(e));
    }
    rc = launch_phase_b(c, P, 1, T, Wmax, ddesc.as<UtrDesc>(), nullptr, nullptr, dth.as<double>(), nullptr,
                        dAT.as<double>(), nullptr, dM.as<double>(), 1, nullptr);
    if (!rc) rc = check_err_flag(c, "get_loglik_marginal_tensor");
    if (!rc) {
        e = hipMemcpy2DAsync(out, (size_t)N * sizeof(double), dM.p, (size_t)Np * sizeof(double),
                             (size_t)N * sizeof(double), (size_t)T * B, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(std::string("copy back: ") + hipGetErrorString(e));
    }
    cleanup();
    return rc;
}

// ---- batched level ---------------------------------------------------------------------------
int scape_hip_batch_load(scape_hip_ctx *c, const scape_hip_params *p, int32_t n_utr, const int64_t *bin_off,
                         const double *x, const double *l, const double *r, const double *pa,
                         const double *cnt, const int64_t *theta_off, const double *all_theta,
                         const double *utr_L, const double *min_theta, const double *unif_ll) {
    if (!c || !p) return fail("NULL ctx/params");
    if (n_utr < 1 || n_utr > 65535) return fail("n_utr must be in [1, 65535] per batch");
    if (p->n_beta < 1 || p->n_beta > SCAPE_MAX_BETA) return fail("n_beta out of range");
    if (p->n_s < 1 || p->n_s > SCAPE_MAX_S) return fail("n_s out of range");
    if (p->nround < 1 || p->nround > 127) return fail("nround out of range");
    if (!bin_off || !x || !l || !r || !pa || !cnt || !theta_off || !all_theta || !utr_L || !min_theta || !unif_ll)
        return fail("NULL array argument");
    CTX_ENTER(c);
    HIPCHK(hipStreamSynchronize(c->stream));   // a queued build of the previous batch still reads its buffers
    c->loaded = c->built = c->build_unchecked = false;
    fill_params(c->prm, p->mu_f, p->sigma_f, p->max_unif_ws, p->n_beta, p->betas, p->n_s, p->s_dis, p->pmf_s, p->nround);
    double bmax = p->betas[0];
    for (int j = 1; j < p->n_beta; ++j) bmax = std::max(bmax, p->betas[j]);

    c->h_desc.assign(n_utr, UtrDesc());
    std::vector<int32_t> loglist;
    size_t at_total = 0, m_total = 0, tiles_total = 0, mlog_total = 0;
    std::vector<int32_t> logbin_utr;
    bool pb_slide = p->n_beta <= 16;
    int T_max = 0, Np_max = 0, W_max = 1, tiles_max_all = 1;
    for (int u = 0; u < n_utr; ++u) {
        UtrDesc &d = c->h_desc[u];
        const int64_t N = bin_off[u + 1] - bin_off[u], T = theta_off[u + 1] - theta_off[u];
        if (N < 1 || T < 1 || N > (1 << 26) || T > 65535) return fail("UTR " + std::to_string(u) + ": bad n_frag / n_theta");
        const double *th = all_theta + theta_off[u];
        for (int64_t i = 1; i < T; ++i)
            if (!(th[i] >= th[i - 1])) return fail("UTR " + std::to_string(u) + ": all_theta must be ascending");
        d.bin_off = bin_off[u];
        d.theta_off = theta_off[u];
        d.N = (int)N;
        d.Np = round_up((int)N, PITCH);
        d.T = (int)T;
        d.at_off = (int64_t)at_total;
        d.m_off = (int64_t)m_total;
        d.log_off = (int64_t)loglist.size();
        d.tile_off = (int64_t)tiles_total;
        {
            const int nt = (int)((T * p->n_beta + TILE_ROWS - 1) / TILE_ROWS);
            tiles_total += (size_t)nt;
            tiles_max_all = std::max(tiles_max_all, nt);
        }
        for (int64_t n = 0; n < N; ++n)
            if (!std::isnan(pa[bin_off[u] + n]) || !std::isnan(r[bin_off[u] + n])) loglist.push_back((int32_t)n);
        d.n_log = (int)(loglist.size() - (size_t)d.log_off);
        d.mlog_off = (int64_t)mlog_total;
        if (d.n_log > 0 && d.n_log <= PB_LOGCAP) mlog_total += (size_t)T * p->n_beta * d.n_log;
        logbin_utr.insert(logbin_utr.end(), (size_t)d.n_log, (int32_t)u);
        {   // uniform grid of integer values: theta_w - theta_i is exact and depends on w - i only
            d.c_lo = 1;
            d.c_hi = 0;
            const double step0 = (T > 1) ? th[1] - th[0] : 0.0;
            bool uni = step0 > 0.0;
            for (int64_t i = 0; i < T && uni; ++i) uni = std::floor(th[i]) == th[i] && (i == 0 || th[i] - th[i - 1] == step0);
            if (uni) {
                const int rmax = (int)std::floor(3 * bmax / step0) + 1;     // reach of the widest window in grid points, + 1 to spare
                d.c_lo = rmax + 1;
                d.c_hi = (int)T - 2 - rmax;
            }
            if (d.c_lo > d.c_hi) pb_slide = false;
        }
        d.unif_ll = unif_ll[u];
        d.L = utr_L[u];
        d.min_theta = min_theta[u];
        at_total += (size_t)T * d.Np;
        m_total += (size_t)T * p->n_beta * d.Np;
        T_max = std::max(T_max, d.T);
        Np_max = std::max(Np_max, d.Np);
        W_max = std::max(W_max, max_window(th, d.T, bmax));
    }
    const int64_t nb = bin_off[n_utr], nt = theta_off[n_utr];
    c->n_utr = n_utr;
    c->n_bins = nb;
    c->T_max = T_max;
    c->Np_max = Np_max;
    c->W_max = W_max;
    c->at_total = at_total;
    c->m_total = m_total;
    c->tiles_total = tiles_total;
    c->tiles_max_all = tiles_max_all;
    c->n_theta_total = (size_t)nt;
    c->mlog_total = mlog_total;
    c->n_logbins = logbin_utr.size();
    c->pb_slide = pb_slide && W_max <= 4 * PB_STEPS && T_max <= 4096;
    const bool trace_load = getenv("SCAPE_HIP_DEBUG") != nullptr;
    const auto t_alloc0 = std::chrono::steady_clock::now();
    if (c->d_x.ensure(nb * 8) || c->d_l.ensure(nb * 8) || c->d_r.ensure(nb * 8) || c->d_pa.ensure(nb * 8) ||
        c->d_cnt.ensure(nb * 8) || c->d_theta.ensure(nt * 8) || c->d_desc.ensure(n_utr * sizeof(UtrDesc)) ||
        c->d_loglist.ensure(loglist.size() * 4) || c->d_AT.ensure(at_total * 8) || c->d_V.ensure(at_total * 8) ||
        c->d_M.ensure(m_total * 8) || c->d_tile_nend.ensure(tiles_total * 4))
        return 1;
    if (p->n_beta <= 16) {   // the three-kernel Phase B (phase_b_split.inc)
        const size_t WP = (size_t)(((W_max + 3) & ~3) + 1);
        if (c->d_tbi.ensure((size_t)nt * 40 * 4) || c->d_tbG.ensure((size_t)nt * 16 * 8) || c->d_tbpm.ensure((size_t)nt * 16 * WP * 8) ||
            c->d_mlog.ensure(mlog_total * 8) || c->d_logbin.ensure(logbin_utr.size() * 4))
            return 1;
        if (!logbin_utr.empty())
            HIPCHK(hipMemcpyAsync(c->d_logbin.p, logbin_utr.data(), logbin_utr.size() * 4, hipMemcpyHostToDevice, c->stream));
    }
    const auto t_up0 = std::chrono::steady_clock::now();
    HIPCHK(hipMemcpyAsync(c->d_x.p, x, nb * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_l.p, l, nb * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_r.p, r, nb * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_pa.p, pa, nb * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_cnt.p, cnt, nb * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_theta.p, all_theta, nt * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_desc.p, c->h_desc.data(), n_utr * sizeof(UtrDesc), hipMemcpyHostToDevice, c->stream));
    if (!loglist.empty())
        HIPCHK(hipMemcpyAsync(c->d_loglist.p, loglist.data(), loglist.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (trace_load) {
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "scape_hip_batch_load: %d UTRs, tensors %.2f GB, device buffers (re)allocated in %.1f ms, uploads %.1f ms\n", n_utr,
                (double)(m_total + 2 * at_total) * 8e-9, std::chrono::duration<double, std::milli>(t_up0 - t_alloc0).count(),
                std::chrono::duration<double, std::milli>(t1 - t_up0).count());
    }
    c->loaded = true;
    return 0;
}

int scape_hip_batch_bytes(scape_hip_ctx *c, int64_t *bytes_batch, int64_t *bytes_free, int64_t *bytes_total) {
    if (!c) return fail("ctx is NULL");
    CTX_ENTER(c);
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (bytes_batch) *bytes_batch = c->loaded ? (int64_t)((2 * c->at_total + c->m_total) * 8 + c->n_bins * 40) : 0;
    if (bytes_free) *bytes_free = (int64_t)f;
    if (bytes_total) *bytes_total = (int64_t)t;
    return 0;
}

int scape_hip_batch_phase_b_form(scape_hip_ctx *c, int32_t *form) {
    if (!c || !form) return fail("ctx / form is NULL");
    *form = c->pb_form;
    return 0;
}

int scape_hip_batch_build(scape_hip_ctx *c) {
    if (!c) return fail("ctx is NULL");
    if (!c->loaded) return fail("no batch loaded");
    CTX_ENTER(c);
    if (finish_build(c)) return 1;   // a second build on the handle must not wipe the first one's pending flag
    dim3 grid((c->Np_max + 255) / 256, (c->T_max + PA_TT - 1) / PA_TT, c->n_utr);
    if (ev_begin(c, 0)) return 1;
    hipLaunchKernelGGL(k_phase_a, grid, dim3(256), 0, c->stream, c->d_desc.as<UtrDesc>(), c->prm, c->d_x.as<double>(),
                       c->d_l.as<double>(), c->d_r.as<double>(), c->d_pa.as<double>(), c->d_theta.as<double>(),
                       c->d_AT.as<double>(), c->d_V.as<double>());
    HIPCHK(hipGetLastError());
    if (ev_end(c, 0)) return 1;
    if (ev_begin(c, 1)) return 1;
    HIPCHK(hipMemsetAsync(c->d_tile_nend.p, 0, c->tiles_total * 4, c->stream));   // Phase B raises the extents by atomicMax
    // SCAPE_HIP_PHASE_B = v2: k_phase_b (one kernel); split: tables + log bins, then the matrix path per alpha (measured,
    // round 3: 10.2 + 10.9 ms against 15.0 for k_phase_b - the log path's gather of A[w][n] at a row pitch of 8.5 KB keeps
    // the texture addresser busy); default where every theta grid is uniform: tables, log-bin columns, sliding windows
    const char *pbm = getenv("SCAPE_HIP_PHASE_B");
    const bool pb_v2 = pbm && strcmp(pbm, "v2") == 0, pb_split = pbm && strcmp(pbm, "split") == 0;
    if (c->prm.B <= 16 && !pb_v2 && (pb_split || c->pb_slide)) {
        const bool slide = c->pb_slide && !pb_split;
        c->pb_form = slide ? 2 : 1;
        if (c->d_err.ensure(sizeof(int))) return 1;
        HIPCHK(hipMemsetAsync(c->d_err.p, 0, sizeof(int), c->stream));
        const int Wmax = c->W_max, WP = ((Wmax + 3) & ~3) + 1, B = c->prm.B;
        const size_t lds_t = ((size_t)B * Wmax + 16) * sizeof(double) + 32 * sizeof(int);
        const size_t lds_l = ((size_t)16 * WP + 16 * PB_LOGCAP) * sizeof(double) + PB_LOGCAP * sizeof(int);
        if (lds_t > 60 * 1024 || lds_l > 60 * 1024) return fail("Phase B: window table does not fit LDS (n_beta x window too large)");
        hipLaunchKernelGGL(k_pb_tables<16>, dim3((unsigned)c->T_max, (unsigned)c->n_utr), dim3(PB_TAB_THREADS), lds_t, c->stream,
                           c->d_desc.as<UtrDesc>(), c->prm, c->d_theta.as<double>(), Wmax, WP, c->d_tbi.as<int32_t>(), c->d_tbG.as<double>(),
                           c->d_tbpm.as<double>(), c->d_err.as<int>(), c->T_max, slide ? nullptr : c->d_loglist.as<int32_t>(), c->d_AT.as<double>(),
                           c->d_M.as<double>(), c->d_mlog.as<double>(), c->d_tile_nend.as<int32_t>());
        HIPCHK(hipGetLastError());
        if (slide) {
            if (c->n_logbins) {
                const size_t lds_c = ((size_t)((c->T_max + 1) & ~1) + (size_t)B * Wmax) * sizeof(double) + 16 * sizeof(int);
                hipLaunchKernelGGL(k_pb_logcol, dim3((unsigned)c->n_logbins), dim3(256), lds_c, c->stream, c->d_desc.as<UtrDesc>(), c->prm,
                                   c->d_theta.as<double>(), c->d_loglist.as<int32_t>(), c->d_logbin.as<int32_t>(), c->d_AT.as<double>(),
                                   c->d_tbi.as<int32_t>(), c->d_tbG.as<double>(), c->d_M.as<double>(), c->d_mlog.as<double>(),
                                   c->d_tile_nend.as<int32_t>(), Wmax);
                HIPCHK(hipGetLastError());
            }
            const int blocks_max = (c->Np_max + 63) / 64;
            const size_t lds_s = ((size_t)16 * WP + 4 * PB_RING * 16) * sizeof(double);
            hipLaunchKernelGGL(k_pb_slide, dim3((unsigned)(((c->n_utr + 7) / 8) * 8 * blocks_max)), dim3(256), lds_s, c->stream,
                               c->d_desc.as<UtrDesc>(), c->prm, c->d_r.as<double>(), c->d_pa.as<double>(), c->d_loglist.as<int32_t>(),
                               c->d_V.as<double>(), c->d_M.as<double>(), c->d_mlog.as<double>(), WP, c->d_tbi.as<int32_t>(),
                               c->d_tbpm.as<double>(), c->n_utr, blocks_max, c->d_tile_nend.as<int32_t>());
        } else {
            hipLaunchKernelGGL(k_phase_b_lin, dim3((unsigned)(((c->n_utr + 7) / 8) * 8 * c->T_max)), dim3(256), lds_l, c->stream,
                               c->d_desc.as<UtrDesc>(), c->prm, c->d_r.as<double>(), c->d_pa.as<double>(), c->d_loglist.as<int32_t>(),
                               c->d_V.as<double>(), c->d_M.as<double>(), c->d_mlog.as<double>(), WP, c->d_tbi.as<int32_t>(),
                               c->d_tbpm.as<double>(), c->n_utr, c->T_max, c->d_tile_nend.as<int32_t>());
        }
        HIPCHK(hipGetLastError());
    } else if ((c->pb_form = 0), launch_phase_b(c, c->prm, c->n_utr, c->T_max, c->W_max, c->d_desc.as<UtrDesc>(), c->d_r.as<double>(),
                       c->d_pa.as<double>(), c->d_theta.as<double>(), c->d_loglist.as<int32_t>(),
                       c->d_AT.as<double>(), c->d_V.as<double>(), c->d_M.as<double>(), 0, c->d_tile_nend.as<int32_t>()))
        return 1;
    if (ev_end(c, 1)) return 1;
    // The kernels are only queued here: the caller's next call (normally scape_hip_batch_em, whose host-side table
    // checks then run while the GPU builds the tensors) launches behind them on the same stream, and reads the
    // consistency flag at its own synchronisation point.  SCAPE_HIP_SYNC_BUILD restores the synchronous behaviour.
    c->build_unchecked = true;
    c->built = true;
    if (getenv("SCAPE_HIP_SYNC_BUILD") && finish_build(c)) return 1;
    return 0;
}

int scape_hip_batch_em(scape_hip_ctx *c, int32_t n_jobs, int32_t kmax, const int32_t *job_utr,
                       const int32_t *job_K, const int32_t *job_fixed, const int32_t *alpha_idx,
                       const int32_t *beta_idx, const double *ws, const int8_t *k_arr,
                       int32_t *alpha_idx_out, int32_t *beta_idx_out, double *ws_out, double *bic_out,
                       int32_t *n_lb_out, double *lb_out) {
    if (!c) return fail("ctx is NULL");
    if (!c->built) return fail("batch_build has not run");
    if (n_jobs < 1) return fail("n_jobs < 1");
    if (kmax < 1 || kmax > SCAPE_MAX_K) return fail("kmax out of range");
    if (!job_utr || !job_K || !job_fixed || !alpha_idx || !beta_idx || !ws || !k_arr || !alpha_idx_out ||
        !beta_idx_out || !ws_out || !bic_out || !n_lb_out)
        return fail("NULL array argument");          // lb_out may be NULL: scape_hip_batch_em_fetch_lb
    CTX_ENTER(c);
    c->last_em_jobs = 0;
    const int nround = c->prm.nround, B = c->prm.B;
    // host-side validation: every index a kernel dereferences is checked here
    for (int j = 0; j < n_jobs; ++j) {
        const int u = job_utr[j], K = job_K[j];
        if (u < 0 || u >= c->n_utr) return fail("job " + std::to_string(j) + ": bad UTR index");
        if (K < 0 || K > kmax) return fail("job " + std::to_string(j) + ": K out of range");
        const int T = c->h_desc[u].T;
        for (int k = 0; k < K; ++k) {
            const int a = alpha_idx[(size_t)j * kmax + k], b = beta_idx[(size_t)j * kmax + k];
            if (a < 0 || a >= T || b < 0 || b >= B) return fail("job " + std::to_string(j) + ": alpha/beta index out of range");
            if (k > 0 && a < alpha_idx[(size_t)j * kmax + k - 1]) return fail("job " + std::to_string(j) + ": alpha indices must be non-decreasing");
        }
        for (int t = 0; t < nround; ++t) {
            const int k = k_arr[(size_t)j * nround + t];
            if (k < 0 || k > K || (K > 0 && k >= K)) return fail("job " + std::to_string(j) + ": k_arr entry out of range");
        }
    }
    {
        // k2_mstep keeps the per-tile list of a UTR's jobs in a fixed LDS table (em_lockstep.inc, s_all)
        std::vector<int32_t> per_utr(c->n_utr, 0);
        for (int j = 0; j < n_jobs; ++j)
            if (!job_fixed[j] && ++per_utr[job_utr[j]] > SCAPE_MAX_JOBS_PER_UTR)
                return fail("UTR " + std::to_string(job_utr[j]) + ": more than " + std::to_string(SCAPE_MAX_JOBS_PER_UTR) +
                            " non-fixed jobs in one call (split the call)");
    }
    const size_t nj = n_jobs;
    if (c->j_utr.ensure(nj * 4) || c->j_K.ensure(nj * 4) || c->j_fixed.ensure(nj * 4) || c->j_a.ensure(nj * kmax * 4) ||
        c->j_b.ensure(nj * kmax * 4) || c->j_ws.ensure(nj * (kmax + 1) * 8) || c->j_karr.ensure(nj * nround) ||
        c->j_ao.ensure(nj * kmax * 4) || c->j_bo.ensure(nj * kmax * 4) || c->j_wso.ensure(nj * (kmax + 1) * 8) ||
        c->j_bic.ensure(nj * 8) || c->j_nlb.ensure(nj * 4) || c->j_lb.ensure(nj * nround * 8))
        return 1;
    HIPCHK(hipMemcpyAsync(c->j_utr.p, job_utr, nj * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->j_K.p, job_K, nj * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->j_fixed.p, job_fixed, nj * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->j_a.p, alpha_idx, nj * kmax * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->j_b.p, beta_idx, nj * kmax * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->j_ws.p, ws, nj * (kmax + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->j_karr.p, k_arr, nj * nround, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_counters.p, 0, N_COUNTERS * sizeof(unsigned long long), c->stream));
    HIPCHK(hipMemsetAsync(c->j_lb.p, 0, nj * nround * 8, c->stream));
    c->h_traffic[3] = 0;
    if (ev_begin(c, 2)) return 1;
#ifdef SCAPE_HIP_TOOLS
    const char *mode = getenv("SCAPE_HIP_EM");
    if (mode && strcmp(mode, "v1") == 0) {
        const size_t lds = (size_t)c->Np_max * sizeof(double);
        if (lds > 150 * 1024) return fail("n_frag too large for the LDS-resident v1 EM kernel (use the default lock-step EM)");
        if (kmax + 1 > 32) return fail("v1 EM kernel: K <= 31");
#define LAUNCH_EM(CM)                                                                                         \
    hipLaunchKernelGGL(k_em<CM>, dim3(n_jobs), dim3(EM_THREADS), lds, c->stream, c->d_desc.as<UtrDesc>(),     \
                       c->prm, c->d_cnt.as<double>(), c->d_M.as<double>(), kmax, c->j_utr.as<int32_t>(),      \
                       c->j_K.as<int32_t>(), c->j_fixed.as<int32_t>(), c->j_a.as<int32_t>(),                  \
                       c->j_b.as<int32_t>(), c->j_ws.as<double>(), c->j_karr.as<int8_t>(),                    \
                       c->j_ao.as<int32_t>(), c->j_bo.as<int32_t>(), c->j_wso.as<double>(),                   \
                       c->j_bic.as<double>(), c->j_nlb.as<int32_t>(), c->j_lb.as<double>(),                   \
                       c->d_counters.as<unsigned long long>())
        if (kmax + 1 <= 8) LAUNCH_EM(8);
        else if (kmax + 1 <= 16) LAUNCH_EM(16);
        else LAUNCH_EM(32);
#undef LAUNCH_EM
        HIPCHK(hipGetLastError());
    } else
#endif
    if (em_lockstep(c, n_jobs, kmax, job_utr, job_fixed)) return 1;
    if (ev_end(c, 2)) return 1;
    HIPCHK(hipMemcpyAsync(alpha_idx_out, c->j_ao.p, nj * kmax * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(beta_idx_out, c->j_bo.p, nj * kmax * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(ws_out, c->j_wso.p, nj * (kmax + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(bic_out, c->j_bic.p, nj * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(n_lb_out, c->j_nlb.p, nj * 4, hipMemcpyDeviceToHost, c->stream));
    if (lb_out) HIPCHK(hipMemcpyAsync(lb_out, c->j_lb.p, nj * nround * 8, hipMemcpyDeviceToHost, c->stream));
    unsigned long long hc[N_COUNTERS];
    HIPCHK(hipMemcpyAsync(hc, c->d_counters.p, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (finish_build(c)) return 1;        // a batch_build queued before this call: its flag is final now
    c->last_em_jobs = n_jobs;
#ifdef M4_STAMPS   // tools-only: k4_mstep's half-chunk iterations by running column groups and wavefront role (whole call)
    {
        const unsigned long long *st = hc + 4 + 5 * 64 + 16;
        for (int ga = 0; ga < 4; ++ga)
            for (int role = 0; role < 2; ++role) {
                const double n = (double)st[32 + ga * 2 + role];
                const unsigned long long *v = st + 4 * (ga * 2 + role);
                if (n > 0)
                    fprintf(stderr, "[mstep stamps] %d running groups, %s wavefronts: %.3g iterations; mean cycles: dma wait %.0f, barrier %.0f, issue %.0f, reads+mfma %.0f (total %.0f)\n",
                            ga + 1, role ? "vector" : "tile", n, v[0] / n, v[1] / n, v[2] / n, v[3] / n, (v[0] + v[1] + v[2] + v[3]) / n);
            }
    }
#endif
#ifdef ESTEP_STAMPS   // tools-only: where a k2_estep wavefront's lifetime goes (cycle sums over all job-rounds of the call)
    {
        const unsigned long long *st = hc + 4 + 5 * 64;
        const double n = (double)st[7];
        if (n > 0)
            fprintf(stderr, "[estep stamps] job-rounds %.0f; mean cycles: set-up %.0f, bin loop %.0f, reductions+weights %.0f, suffix sums %.0f, tail %.0f (total %.0f)\n",
                    n, st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, (st[0] + st[1] + st[2] + st[3] + st[4]) / n);
    }
#endif
    for (int i = 0; i < 3; ++i) c->h_counters[i] = hc[i];
    c->h_traffic[0] = c->h_traffic[1] = c->h_traffic[2] = 0;
    for (int i = 0; i < 64; ++i) {
        c->h_counters[1] += hc[CNT_SLAB + i];   // sharded slab-element counter of the lock-step EM
        c->h_traffic[0] += hc[CNT_MSTEP_TENSOR + i];
        c->h_traffic[1] += hc[CNT_MSTEP_VREQ + i];
        c->h_traffic[2] += hc[CNT_MSTEP_VUNI + i];
    }
    return 0;
}

__global__ void k_gather_rows(const double *__restrict__ src, const int32_t *__restrict__ idx, double *__restrict__ dst,
                              int rowlen, int n_sel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sel * rowlen) dst[i] = src[(size_t)idx[i / rowlen] * rowlen + i % rowlen];
}

int scape_hip_batch_em_fetch_lb(scape_hip_ctx *c, int32_t n_sel, const int32_t *job_idx, double *lb_out) {
    if (!c) return fail("ctx is NULL");
    if (n_sel < 1 || !job_idx || !lb_out) return fail("fetch_lb: bad arguments");
    CTX_ENTER(c);
    if (finish_build(c)) return 1;
    if (c->last_em_jobs < 1) return fail("fetch_lb: no completed scape_hip_batch_em call on this handle");
    for (int i = 0; i < n_sel; ++i)
        if (job_idx[i] < 0 || job_idx[i] >= c->last_em_jobs) return fail("fetch_lb: job index out of range");
    const int nround = c->prm.nround;
    if (c->j_sel.ensure((size_t)n_sel * 4) || c->j_lbsel.ensure((size_t)n_sel * nround * 8)) return 1;
    HIPCHK(hipMemcpyAsync(c->j_sel.p, job_idx, (size_t)n_sel * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n_sel * nround + 255) / 256)), dim3(256), 0, c->stream,
                       c->j_lb.as<double>(), c->j_sel.as<int32_t>(), c->j_lbsel.as<double>(), nround, n_sel);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(lb_out, c->j_lbsel.p, (size_t)n_sel * nround * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int scape_hip_batch_labels(scape_hip_ctx *c, int32_t n_sel, int32_t kmax, const int32_t *sel_utr,
                           const int32_t *sel_K, const int32_t *alpha_idx, const int32_t *beta_idx,
                           const double *ws, int32_t *labels_out) {
    if (!c) return fail("ctx is NULL");
    if (!c->built) return fail("batch_build has not run");
    if (n_sel < 1) return fail("n_sel < 1");
    if (kmax < 1 || kmax > SCAPE_MAX_K) return fail("kmax out of range");
    if (!sel_utr || !sel_K || !alpha_idx || !beta_idx || !ws || !labels_out) return fail("NULL array argument");
    CTX_ENTER(c);
    const int B = c->prm.B;
    for (int j = 0; j < n_sel; ++j) {
        const int u = sel_utr[j], K = sel_K[j];
        if (u < 0 || u >= c->n_utr) return fail("labels: bad UTR index");
        if (K < 0 || K > kmax) return fail("labels: K out of range");
        for (int k = 0; k < K; ++k) {
            const int a = alpha_idx[(size_t)j * kmax + k], b = beta_idx[(size_t)j * kmax + k];
            if (a < 0 || a >= c->h_desc[u].T || b < 0 || b >= B) return fail("labels: alpha/beta index out of range");
        }
    }
    const size_t ns = n_sel;
    if (c->l_utr.ensure(ns * 4) || c->l_K.ensure(ns * 4) || c->l_a.ensure(ns * kmax * 4) || c->l_b.ensure(ns * kmax * 4) ||
        c->l_ws.ensure(ns * (kmax + 1) * 8) || c->l_labels.ensure((size_t)c->n_bins * 4))
        return 1;
    HIPCHK(hipMemcpyAsync(c->l_utr.p, sel_utr, ns * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_K.p, sel_K, ns * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_a.p, alpha_idx, ns * kmax * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_b.p, beta_idx, ns * kmax * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_ws.p, ws, ns * (kmax + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (ev_begin(c, 3)) return 1;
#define LAUNCH_LAB(CM)                                                                                         \
    hipLaunchKernelGGL(k_labels<CM>, dim3(n_sel), dim3(256), 0, c->stream, c->d_desc.as<UtrDesc>(), c->prm,    \
                       c->d_cnt.as<double>(), c->d_M.as<double>(), kmax, c->l_utr.as<int32_t>(),               \
                       c->l_K.as<int32_t>(), c->l_a.as<int32_t>(), c->l_b.as<int32_t>(), c->l_ws.as<double>(), \
                       c->l_labels.as<int32_t>())
    if (kmax + 1 <= 8) LAUNCH_LAB(8);
    else if (kmax + 1 <= 16) LAUNCH_LAB(16);
    else if (kmax + 1 <= 32) LAUNCH_LAB(32);
    else LAUNCH_LAB(64);
#undef LAUNCH_LAB
    HIPCHK(hipGetLastError());
    if (ev_end(c, 3)) return 1;
    // copy back only the selected UTRs' bins; contiguous runs of selected UTRs go in one transfer
    {
        std::vector<char> sel(c->n_utr, 0);
        for (int j = 0; j < n_sel; ++j) sel[sel_utr[j]] = 1;
        int u = 0;
        while (u < c->n_utr) {
            if (!sel[u]) {
                ++u;
                continue;
            }
            int v = u;
            while (v + 1 < c->n_utr && sel[v + 1]) ++v;
            const int64_t b0 = c->h_desc[u].bin_off, b1 = c->h_desc[v].bin_off + c->h_desc[v].N;
            HIPCHK(hipMemcpyAsync(labels_out + b0, c->l_labels.as<int32_t>() + b0, (size_t)(b1 - b0) * 4,
                                  hipMemcpyDeviceToHost, c->stream));
            u = v + 1;
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (finish_build(c)) return 1;
    return 0;
}

int scape_hip_batch_fetch_loglik(scape_hip_ctx *c, int32_t utr, double *A_out) {
    if (!c || !A_out) return fail("NULL argument");
    if (!c->built) return fail("batch_build has not run");
    if (utr < 0 || utr >= c->n_utr) return fail("bad UTR index");
    CTX_ENTER(c);
    if (finish_build(c)) return 1;
    const UtrDesc &d = c->h_desc[utr];
    std::vector<double> at((size_t)d.T * d.Np);
    HIPCHK(hipMemcpy(at.data(), c->d_AT.as<double>() + d.at_off, at.size() * 8, hipMemcpyDeviceToHost));
    for (int n = 0; n < d.N; ++n)
        for (int t = 0; t < d.T; ++t) A_out[(size_t)n * d.T + t] = at[(size_t)t * d.Np + n];
    return 0;
}

int scape_hip_batch_fetch_tensor(scape_hip_ctx *c, int32_t utr, double *M_out) {
    if (!c || !M_out) return fail("NULL argument");
    if (!c->built) return fail("batch_build has not run");
    if (utr < 0 || utr >= c->n_utr) return fail("bad UTR index");
    CTX_ENTER(c);
    if (finish_build(c)) return 1;
    const UtrDesc &d = c->h_desc[utr];
    HIPCHK(hipMemcpy2D(M_out, (size_t)d.N * 8, c->d_M.as<double>() + d.m_off, (size_t)d.Np * 8, (size_t)d.N * 8,
                       (size_t)d.T * c->prm.B, hipMemcpyDeviceToHost));
    return 0;
}

int scape_hip_timing_reset(scape_hip_ctx *c) {
    if (!c) return fail("ctx is NULL");
    CTX_ENTER(c);
    if (ev_collect(c)) return 1;
    for (int w = 0; w < 6; ++w) {
        c->ms_acc[w] = 0;
        c->n_acc[w] = 0;
    }
    return 0;
}

int scape_hip_timing_get(scape_hip_ctx *c, int32_t which, double *ms_total, int32_t *n_launches) {
    if (!c) return fail("ctx is NULL");
    if (which < 0 || which > 5) return fail("which out of range");
    CTX_ENTER(c);
    if (ev_collect(c)) return 1;
    if (ms_total) *ms_total = c->ms_acc[which];
    if (n_launches) *n_launches = c->n_acc[which];
    return 0;
}

int scape_hip_em_counters(scape_hip_ctx *c, int64_t *rounds, int64_t *slab_elems, int64_t *z_elems) {
    if (!c) return fail("ctx is NULL");
    if (rounds) *rounds = (int64_t)c->h_counters[0];
    if (slab_elems) *slab_elems = (int64_t)c->h_counters[1];
    if (z_elems) *z_elems = (int64_t)c->h_counters[2];
    return 0;
}

int scape_hip_em_traffic(scape_hip_ctx *c, int64_t *mstep_tensor_bytes, int64_t *mstep_v_bytes_requested,
                         int64_t *mstep_v_bytes_unique, int64_t *mstep_launches) {
    if (!c) return fail("ctx is NULL");
    if (mstep_tensor_bytes) *mstep_tensor_bytes = (int64_t)c->h_traffic[0];
    if (mstep_v_bytes_requested) *mstep_v_bytes_requested = (int64_t)c->h_traffic[1];
    if (mstep_v_bytes_unique) *mstep_v_bytes_unique = (int64_t)c->h_traffic[2];
    if (mstep_launches) *mstep_launches = (int64_t)c->h_traffic[3];
    return 0;
}

}  // extern "C"
