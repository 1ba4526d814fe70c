/*
 * scape_hip.h - C ABI of libscape_hip.so, the MI355X (gfx950) implementation of
 * SCAPE's `infer_pa` hot path.
 *
 * Conventions
 *   - plain pointers and sizes only; the CALLER owns every host buffer, the library
 *     owns device memory behind an opaque context (one context per GPU);
 *   - every entry point returns 0 on success, non-zero on error;
 *     scape_hip_last_error() returns the message of the calling thread's last error;
 *   - calls on one context are not re-entrant (one host thread per context): a call that arrives while
 *     another is in flight on the same context fails with an error; use one context per thread;
 *   - all floating point is IEEE f64; SCAPE_SENT is the reference's finite "-inf"
 *     (np.finfo('f').min, reference src/scape/taichi_core.py:8, apa_core.py:428).
 *
 * Reference interface replaced (paths relative to the reference checkout):
 *   operator level : src/scape/apa_core.py:23 imports four host functions from
 *                    src/scape/taichi_core.py (:183, :200, :210, :237); the four
 *                    scape_hip_loglik_* / scape_hip_get_loglik_marginal_tensor entry
 *                    points below take the same arrays and return the same arrays.
 *   batched level  : the per-UTR body of ApaModel.run (apa_core.py:930-981) -
 *                    Phase A (:954-957), Phase B (:959), em_algo (:714-779) and
 *                    get_label (:873-881) - for many UTRs at once.
 */
#ifndef SCAPE_HIP_H
#define SCAPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCAPE_HIP_ABI_VERSION 4
#define SCAPE_SENT (-3.4028234663852886e38)
#define SCAPE_MAX_BETA 160   /* max len(predef_beta_arr): beta_step >= 0.44 at the default max_beta = 70 (ABI 3: 64) */
#define SCAPE_MAX_S 64       /* max len(s_dis_arr)       */
#define SCAPE_MAX_K 63       /* max number of pA components per model (K+1 <= 64; ABI 3: 31) */
#define SCAPE_MAX_JOBS_PER_UTR 1024  /* max non-fixed jobs on one UTR in one scape_hip_batch_em call */

typedef struct scape_hip_ctx scape_hip_ctx;

/* model constants shared by every UTR of a batch (ApaModel.__init__, apa_core.py:333-437) */
typedef struct scape_hip_params {
    double mu_f;            /* apa_core.py:399 */
    double sigma_f;         /* apa_core.py:400 */
    double max_unif_ws;     /* apa_core.py:414 */
    int32_t n_beta;         /* len(predef_beta_arr), apa_core.py:942 */
    int32_t n_s;            /* len(s_dis_arr), apa_core.py:394 */
    int32_t nround;         /* ApaModel.nround = 50, apa_core.py:422 */
    int32_t reserved;
    double betas[SCAPE_MAX_BETA];
    double s_dis[SCAPE_MAX_S];
    double pmf_s[SCAPE_MAX_S];
} scape_hip_params;

int scape_hip_abi_version(void);
const char *scape_hip_last_error(void);
int scape_hip_device_count(int *count);
int scape_hip_create(int device, scape_hip_ctx **out);
int scape_hip_destroy(scape_hip_ctx *ctx);
int scape_hip_device_name(scape_hip_ctx *ctx, char *buf, int buflen);

/* ---- operator level: the taichi_core seam ------------------------------------------- */
/* taichi_core.py:183-197 loglik_xlr_t_pa(x_arr, l_arr, pa_arr, theta, sigma_f) -> out[n] */
int scape_hip_loglik_xlr_t_pa(scape_hip_ctx *ctx, const double *x, const double *l,
                              const double *pa, int32_t n, double theta, double sigma_f,
                              double *out);
/* taichi_core.py:200-207 loglik_xlr_t_r_known(x, l, r, s_dis, pmf_s, theta, mu_f, sigma_f) */
int scape_hip_loglik_xlr_t_r_known(scape_hip_ctx *ctx, const double *x, const double *l,
                                   const double *r, int32_t n, const double *s_dis,
                                   const double *pmf_s, int32_t n_s, double theta, double mu_f,
                                   double sigma_f, double *out);
/* taichi_core.py:210-215 loglik_xlr_t_r_unknown(x, l, r, s_dis, pmf_s, theta, mu_f, sigma_f);
   r is ignored by the reference kernel too and may be NULL */
int scape_hip_loglik_xlr_t_r_unknown(scape_hip_ctx *ctx, const double *x, const double *l,
                                     const double *r, int32_t n, const double *s_dis,
                                     const double *pmf_s, int32_t n_s, double theta, double mu_f,
                                     double sigma_f, double *out);
/* taichi_core.py:237-246 get_loglik_marginal_tensor(all_theta, predef_beta_arr, loglik_xlr_t_arr)
   loglik_xlr_t_arr is [n_frag, n_theta] row-major, out is [n_theta, n_beta, n_frag] */
int scape_hip_get_loglik_marginal_tensor(scape_hip_ctx *ctx, const double *all_theta,
                                         int32_t n_theta, const double *betas, int32_t n_beta,
                                         const double *loglik_xlr_t_arr, int32_t n_frag,
                                         double *out);

/* ---- batched level -------------------------------------------------------------------- */
/*
 * Upload a batch of binned UTRs (output of bin_data, apa_core.py:285-327).  Ragged arrays:
 * UTR u owns bins [bin_off[u], bin_off[u+1]) of x/l/r/pa/cnt and grid points
 * [theta_off[u], theta_off[u+1]) of all_theta (ascending; apa_core.py:940).  r / pa are NaN
 * where unknown (apa_core.py:439-452).  utr_L = ApaModel.L (apa_core.py:387), min_theta
 * (:407), unif_ll = lik_f0(log=True) (:576-584).  Allocates the Phase-A matrix and the
 * marginal tensor on the device.  Replaces any previously loaded batch.
 */
int scape_hip_batch_load(scape_hip_ctx *ctx, const scape_hip_params *params, int32_t n_utr,
                         const int64_t *bin_off, const double *x, const double *l,
                         const double *r, const double *pa, const double *cnt,
                         const int64_t *theta_off, const double *all_theta,
                         const double *utr_L, const double *min_theta, const double *unif_ll);
/* device bytes the loaded batch needs / the device offers (for sizing waves on the host) */
int scape_hip_batch_bytes(scape_hip_ctx *ctx, int64_t *bytes_batch, int64_t *bytes_free,
                          int64_t *bytes_total);
/* Phase A (apa_core.py:954-957) then Phase B (apa_core.py:959) for every UTR of the batch.  The kernels are queued,
   not awaited: the next call on the context runs behind them, and a device-side consistency failure of the build is
   reported by that call (batch_em / batch_labels / batch_fetch_*). */
int scape_hip_batch_build(scape_hip_ctx *ctx);
/* which form of Phase B the last batch_build queued (diagnostics / tests; the results do not depend on it):
   0 = one kernel (the form of get_loglik_marginal_tensor; non-uniform theta grids, > 16 beta values),
   1 = window tables + per-alpha matrix path, 2 = window tables + log-bin columns + sliding windows (uniform theta grids) */
int scape_hip_batch_phase_b_form(scape_hip_ctx *ctx, int32_t *form);
/*
 * n_jobs em_algo calls (apa_core.py:714-779).  Job j works on UTR job_utr[j] with job_K[j]
 * components from the init (alpha_idx, beta_idx = indices into that UTR's all_theta / betas,
 * ws) and the update order k_arr (gen_k_arr, apa_core.py:653-677); job_fixed[j] != 0 selects
 * mstep_fixed (apa_core.py:552-557).  Tables are padded to kmax / kmax+1 / nround per job.
 * Outputs (same padding): sorted alpha_idx / beta_idx / ws, bic, lb_arr and its length.
 * Limits (checked, an error is returned): kmax <= SCAPE_MAX_K; at most SCAPE_MAX_JOBS_PER_UTR jobs with
 * job_fixed == 0 on any one UTR per call (the reference's own sweep is (n_max_apa - n_min_apa + 1) * 10
 * jobs per UTR, apa_core.py:846-871, :965).
 */
int scape_hip_batch_em(scape_hip_ctx *ctx, int32_t n_jobs, int32_t kmax, const int32_t *job_utr,
                       const int32_t *job_K, const int32_t *job_fixed, const int32_t *alpha_idx,
                       const int32_t *beta_idx, const double *ws, const int8_t *k_arr,
                       int32_t *alpha_idx_out, int32_t *beta_idx_out, double *ws_out,
                       double *bic_out, int32_t *n_lb_out, double *lb_out);
/*
 * lb_out of scape_hip_batch_em may be NULL (ABI version 3): the lb_arr rows (nround doubles per job, 20 MB for the
 * 51,200 jobs of a 512-UTR sweep) then stay on the device, and the caller fetches the rows of the few jobs it keeps -
 * the BIC winners that become Parameters.lb_arr (apa_core.py:760-766, :972) - with this call.  job_idx are indices
 * into the job tables of the LAST scape_hip_batch_em call on this handle; lb_out is [n_sel][nround], rows valid up
 * to that job's n_lb.
 */
int scape_hip_batch_em_fetch_lb(scape_hip_ctx *ctx, int32_t n_sel, const int32_t *job_idx, double *lb_out);
/*
 * get_label (apa_core.py:873-881) for n_sel models: per-bin arg-max of the responsibilities.
 * labels_out is indexed like the batch's bins (bin_off of sel_utr[i]); only the bins of the
 * selected UTRs are written.
 */
int scape_hip_batch_labels(scape_hip_ctx *ctx, int32_t n_sel, int32_t kmax, const int32_t *sel_utr,
                           const int32_t *sel_K, const int32_t *alpha_idx, const int32_t *beta_idx,
                           const double *ws, int32_t *labels_out);
/* parity-test access to intermediates: A[n_frag, n_theta] and M[n_theta, n_beta, n_frag] of one UTR */
int scape_hip_batch_fetch_loglik(scape_hip_ctx *ctx, int32_t utr, double *A_out);
int scape_hip_batch_fetch_tensor(scape_hip_ctx *ctx, int32_t utr, double *M_out);
int scape_hip_batch_free(scape_hip_ctx *ctx);

/*
 * Kernel timing measured with HIP events on the stream the kernels are launched on.
 * which: 0 = Phase A, 1 = Phase B, 2 = EM (one whole scape_hip_batch_em call), 3 = labels,
 * 4 = the per-round E-step kernel, 5 = the per-round M-step kernel (4 and 5 are recorded only when the
 * environment variable SCAPE_HIP_ROUND_TIMING is set).  Returns the sum of the durations and the number
 * of launches since the last scape_hip_timing_reset().
 */
int scape_hip_timing_reset(scape_hip_ctx *ctx);
int scape_hip_timing_get(scape_hip_ctx *ctx, int32_t which, double *ms_total, int32_t *n_launches);
/* EM work counters of the last scape_hip_batch_em call (for the algorithmic-bytes formula):
   sum over jobs of rounds, and of rounds x window rows x n_beta x n_frag (tensor elements read
   by the grid arg-max M-step, apa_core.py:507-523), and of rounds x n_frag x (K+1) */
int scape_hip_em_counters(scape_hip_ctx *ctx, int64_t *rounds, int64_t *slab_elems,
                          int64_t *z_elems);

/* HBM-side byte tally of the M-step kernel over the last scape_hip_batch_em call, counted by the kernel
   itself: tensor-tile bytes it streamed (each live 64-row tile x the bins it needs, once per round), the
   v-vector bytes as requested by every workgroup (mostly L2 hits) and counted once per (job, round), and
   the number of M-step launches.  Replaces nothing in the reference (measurement only). */
int scape_hip_em_traffic(scape_hip_ctx *ctx, int64_t *mstep_tensor_bytes, int64_t *mstep_v_bytes_requested,
                         int64_t *mstep_v_bytes_unique, int64_t *mstep_launches);

#ifdef __cplusplus
}
#endif
#endif
