/* scape_host.h - C ABI of libscape_host.so: host-side native helpers of the infer_pa path.
 *
 * No GPU code and no likelihood / EM arithmetic lives here (that is libscape_hip.so, scape_hip.h).
 * This library draws the EM restart tables the reference draws from numpy's legacy RandomState:
 *   sample_alpha / init_para   /root/reference/src/scape/apa_core.py:781-829
 *   init_ws                    apa_core.py:809-815
 *   gen_k_arr                  apa_core.py:653-677
 *   rm_component's draws       apa_core.py:843 -> :709 -> :720
 * bit for bit (MT19937, legacy seeding, random_sample, masked-rejection bounded integers,
 * choice(replace=False, p=...)), so that a 512-UTR batch is planned in milliseconds instead of
 * seconds.  Generator state = uint32[625]: the 624 key words of numpy's
 * RandomState.get_state()[1] followed by get_state()[2] (pos).
 *
 * Return codes: 0 ok; 1 = input not covered by the fast path (the caller uses the Python sampler,
 * scape_amd/host.py::Sampler, which draws the same numbers or raises what numpy raises);
 * 2 = out of memory.
 */
#ifndef SCAPE_HOST_H
#define SCAPE_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCAPE_HOST_ABI_VERSION 2
#define SCAPE_HOST_MAX_K 64

int scape_host_abi_version(void);

/* numpy.random.RandomState(seed) for an integer seed < 2**32 */
void scape_host_mt_seed(uint32_t seed, uint32_t *state625);
/* RandomState.random_sample() */
double scape_host_mt_double(uint32_t *state625);

/* init_ws(K): w[K+1] */
int scape_host_init_ws(uint32_t *state625, int K, double max_unif_ws, double *w);
/* gen_k_arr(K, n_round): ka[n_round] */
int scape_host_k_arr(uint32_t *state625, int K, int n_round, int8_t *ka);
/* init_para(K) then gen_k_arr(K): a[K] theta index, b[K] beta index, w[K+1], ka[n_round].
 * peaks / peak_w: the coverage peaks and their normalised weights (apa_core.py:781-790);
 * theta[T]: the UTR's theta grid; L: UTR length; n_beta: len(predef_beta_arr);
 * shift_scale = 5 * beta_step (apa_core.py:826). */
int scape_host_init_job(uint32_t *state625, const double *peaks, const double *peak_w, int n_peak,
                        const double *theta, int T, int L, int n_beta, double shift_scale, double max_unif_ws,
                        int K, int n_round, int32_t *a, int32_t *b, double *w, int8_t *ka);

/* One K sweep of one UTR (run() apa_core.py:965 -> em_optim0 :846-871) continuing the caller's stream:
 * K = n_max .. n_min (descending), n_trial restarts each, written as rows of the padded job tables
 * (jk[row] = K; a, b pitch kmax; w pitch kmax+1; ka pitch n_round; rows beyond K are left untouched -
 * zero them first).  Returns 1 without drawing anything if an input is not covered. */
int scape_host_sweep(uint32_t *state625, const double *peaks, const double *peak_w, int n_peak,
                     const double *theta, int T, int L, int n_beta, double shift_scale, double max_unif_ws,
                     int n_max, int n_min, int n_trial, int n_round, int kmax,
                     int32_t *jk, int32_t *a, int32_t *b, double *w, int8_t *ka);

/* scape_host_sweep for many independent streams at once (the current UTRs of several chunk files,
 * scape_amd/engine.py::Engine.run_streams), spread over n_threads host threads.  Every item needs its own
 * state625; status is set per item (0 ok / 1 declined, nothing drawn). */
struct scape_host_sweep_item {
    uint32_t *state625;
    const double *peaks, *peak_w, *theta;
    int32_t n_peak, T, L, n_beta;
    double shift_scale, max_unif_ws;
    int32_t n_max, n_min, kmax, status;
    int32_t *jk, *a, *b;
    double *w;
    int8_t *ka;
};
int scape_host_sweep_batch(struct scape_host_sweep_item *items, int n, int n_trial, int n_round, int n_threads);

/* Whole-batch plan (scape_amd/engine.py::Engine.plan): for every UTR u, seeded RandomState(seeds[u]),
 * the n_trial restarts of every K = n_max[u] .. n_min[u] (descending, apa_core.py:846-871) into the
 * padded job tables of scape_hip_batch_em, the generator state left behind, and the prune tables
 * for every K' < n_max[u].  status[u] = 0 ok / 1 fall back to Python for this UTR (its table rows are
 * then undefined).  UTRs are spread over n_threads host threads. */
struct scape_host_plan_args {
    int32_t n_utr, n_threads, n_trial, n_round;
    int32_t kmax;                 /* row pitch of a / b (w: kmax+1) */
    int32_t kcap;                 /* row pitch of the prune tables */
    const uint32_t *seeds;        /* [n_utr] */
    const int64_t *peak_off;      /* [n_utr+1] into peaks / peak_w */
    const double *peaks, *peak_w;
    const int64_t *theta_off;     /* [n_utr+1] into theta */
    const double *theta;
    const int32_t *L, *n_max, *n_min, *n_beta;   /* [n_utr] each */
    const double *shift_scale, *max_unif_ws;     /* [n_utr] each */
    const int64_t *spans;         /* [n_utr+1] first job row of each UTR */
    int32_t *ju, *jk;             /* [n_job] */
    int32_t *a, *b;               /* [n_job][kmax] */
    double *w;                    /* [n_job][kmax+1] */
    int8_t *ka;                   /* [n_job][n_round] */
    uint32_t *states;             /* [n_utr][625] */
    double *prune_w;              /* [n_utr][kcap][kcap+1] */
    int8_t *prune_ka;             /* [n_utr][kcap][n_round] */
    uint32_t *prune_states;       /* [n_utr][kcap][625] */
    int32_t *status;              /* [n_utr] */
};
int scape_host_plan(const struct scape_host_plan_args *args);

#ifdef __cplusplus
}
#endif
#endif
