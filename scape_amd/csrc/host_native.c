/* libscape_host.so - host-side native helpers of the infer_pa path (plain C, no GPU code).
 *
 * Restart sampling: the reference draws every EM restart from numpy's legacy RandomState
 * (apa_core.py:781-829 sample_alpha / init_para / init_ws, :653-677 gen_k_arr).  At MI355X rates the
 * Python sampler (scape_amd/host.py::Sampler) is ~60x slower than the GPU side it feeds, so the same
 * draws are produced here, bit for bit: MT19937 in numpy's legacy seeding, random_sample doubles,
 * the legacy masked-rejection bounded integers (shuffle / permutation / randint) and the
 * choice(replace=False, p=...) loop.  tests/test_host.py checks every table against the Python
 * sampler.  Any input the fast path does not cover returns status 1 and the caller falls back to the
 * Python sampler for that UTR (which then raises whatever numpy raises).
 *
 * Read binning (apa_core.py:285-327) and the coverage profile (:454-462, :680-700) stay in Python.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "scape_host.h"

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];
    int32_t pos;
} mt_t; /* == uint32[625], the layout the Python side hands in and out */

static void mt_seed(mt_t *s, uint32_t seed)
{
    for (int i = 0; i < MT_N; i++) {
        s->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    s->pos = MT_N;
}

static void mt_refill(mt_t *s)
{
    uint32_t *k = s->key, y;
    int i;
    for (i = 0; i < MT_N - MT_M; i++) {
        y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    for (; i < MT_N - 1; i++) {
        y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    y = (k[MT_N - 1] & 0x80000000u) | (k[0] & 0x7fffffffu);
    k[MT_N - 1] = k[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    s->pos = 0;
}

static inline uint32_t mt_u32(mt_t *s)
{
    if (s->pos == MT_N) mt_refill(s);
    uint32_t y = s->key[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static inline double mt_double(mt_t *s)
{
    int32_t a = (int32_t)(mt_u32(s) >> 5), b = (int32_t)(mt_u32(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* numpy legacy random_interval: uniform on [0, max], smallest-mask rejection on 32-bit draws */
static inline uint32_t mt_interval(mt_t *s, uint32_t max)
{
    if (max == 0) return 0;
    uint32_t mask = max, v;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    while ((v = (mt_u32(s) & mask)) > max) {
    }
    return v;
}

static void shuffle_i32(mt_t *s, int32_t *a, int n)
{
    for (int i = n - 1; i >= 1; i--) {
        uint32_t j = mt_interval(s, (uint32_t)i);
        int32_t t = a[i];
        a[i] = a[j];
        a[j] = t;
    }
}

/* init_ws (apa_core.py:809-815) */
static void draw_init_ws(mt_t *s, int K, double max_unif_ws, double *w)
{
    double tot = 0.0;
    for (int i = 0; i <= K; i++) {
        w[i] = mt_double(s);
        tot += w[i];
    }
    for (int i = 0; i <= K; i++) w[i] /= tot;
    if (w[K] > max_unif_ws) {
        double f = 1.0 - max_unif_ws;
        for (int i = 0; i < K; i++) w[i] *= f;
        w[K] = max_unif_ws;
    }
}

/* gen_k_arr (apa_core.py:653-677): no draw at all for K <= 1 */
static void draw_k_arr(mt_t *s, int K, int n, int8_t *out)
{
    if (K <= 1) {
        memset(out, 0, (size_t)n);
        return;
    }
    int32_t arr[SCAPE_HOST_MAX_K];
    for (int i = 0; i < K; i++) arr[i] = i;
    shuffle_i32(s, arr, K);                        /* permutation(K) */
    for (int t0 = 0; t0 < n; t0 += K) {
        shuffle_i32(s, arr, K);
        int m = K < n - t0 ? K : n - t0;
        for (int i = 0; i < m; i++) out[t0 + i] = (int8_t)arr[i];
    }
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* find_nearest (apa_core.py:535-549): nearest grid index, ties to the upper point */
static int32_t snap(const double *g, int T, double v)
{
    int lo = 0, hi = T;                            /* lower bound */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (g[mid] < v) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return 0;
    if (lo == T) return T - 1;
    return (v - g[lo - 1]) >= (g[lo] - v) ? lo : lo - 1;
}

typedef struct {
    const double *peaks, *peak_w, *theta;
    int n_peak, T, L, n_beta;
    double shift_scale, max_unif_ws;                /* shift_scale = 5 * beta_step (:826) */
} utr_in;

/* scratch sized by the caller: perm[L], and n_peak doubles x2 */
typedef struct {
    int32_t *perm;
    int perm_cap;
    double *p, *cdf;
    int32_t *found;
    int pk_cap;
} scratch;

static int scratch_fit(scratch *sc, int L, int n_peak)
{
    if (L > sc->perm_cap) {
        free(sc->perm);
        sc->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
        sc->perm_cap = sc->perm ? L : 0;
        if (!sc->perm) return 1;
    }
    if (n_peak > sc->pk_cap) {
        free(sc->p); free(sc->cdf); free(sc->found);
        sc->p = (double *)malloc(sizeof(double) * (size_t)n_peak);
        sc->cdf = (double *)malloc(sizeof(double) * (size_t)n_peak);
        sc->found = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_peak);
        sc->pk_cap = (sc->p && sc->cdf && sc->found) ? n_peak : 0;
        if (!sc->pk_cap) return 1;
    }
    return 0;
}

static void scratch_free(scratch *sc)
{
    free(sc->perm); free(sc->p); free(sc->cdf); free(sc->found);
    memset(sc, 0, sizeof(*sc));
}

/* Can the fast path draw for this UTR / K at all?  (numpy would raise, or the tables would not fit) */
static int utr_supported(const utr_in *q, int K)
{
    if (K < 1 || K > SCAPE_HOST_MAX_K || q->T < 1 || q->n_beta < 1 || q->L < 1) return 0;
    if (K > q->n_peak && K - q->n_peak > q->L) return 0;     /* choice(): larger sample than population */
    if (K <= q->n_peak) {
        double tot = 0.0;
        int nz = 0;
        for (int i = 0; i < q->n_peak; i++) {
            double v = q->peak_w[i];
            if (!(v >= 0.0) || !isfinite(v)) return 0;
            tot += v;
            nz += v > 0.0;
        }
        if (fabs(tot - 1.0) > 1e-9 || nz < K) return 0;
    }
    return 1;
}

/* init_para (apa_core.py:817-829) followed by the k_arr em_algo draws first (:720) */
static int draw_init_job(mt_t *s, const utr_in *q, int K, int n_round, scratch *sc,
                         int32_t *a, int32_t *b, double *w, int8_t *ka)
{
    double res[SCAPE_HOST_MAX_K];
    if (scratch_fit(sc, K > q->n_peak ? q->L : 0, q->n_peak)) return 2;
    if (K <= q->n_peak) {
        /* RandomState.choice(peaks, size=K, replace=False, p=peak_w) */
        int n_uniq = 0, P = q->n_peak;
        memcpy(sc->p, q->peak_w, sizeof(double) * (size_t)P);
        while (n_uniq < K) {
            int m = K - n_uniq;
            double x[SCAPE_HOST_MAX_K];
            for (int i = 0; i < m; i++) x[i] = mt_double(s);
            for (int i = 0; i < n_uniq; i++) sc->p[sc->found[i]] = 0.0;
            double acc = 0.0;
            for (int i = 0; i < P; i++) {
                acc += sc->p[i];
                sc->cdf[i] = acc;
            }
            double last = sc->cdf[P - 1];
            for (int i = 0; i < P; i++) sc->cdf[i] /= last;
            int base = n_uniq;
            for (int i = 0; i < m; i++) {
                int lo = 0, hi = P;                 /* searchsorted(side='right') */
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (sc->cdf[mid] <= x[i]) lo = mid + 1; else hi = mid;
                }
                if (lo >= P) lo = P - 1;            /* cannot happen for x < 1; keeps indices valid */
                int dup = 0;
                for (int j = base; j < n_uniq; j++) dup |= sc->found[j] == lo;
                if (!dup) sc->found[n_uniq++] = lo; /* first occurrences, in draw order */
            }
        }
        for (int i = 0; i < K; i++) res[i] = q->peaks[sc->found[i]];
    } else {
        /* peaks + RandomState.choice(L, size=K-n_peak, replace=False) = permutation(L)[:size] */
        int m = K - q->n_peak;
        for (int i = 0; i < q->L; i++) sc->perm[i] = i;
        shuffle_i32(s, sc->perm, q->L);
        for (int i = 0; i < q->n_peak; i++) res[i] = q->peaks[i];
        for (int i = 0; i < m; i++) res[q->n_peak + i] = (double)sc->perm[i];
    }
    double scale = q->shift_scale;
    for (int i = 0; i < K; i++) {
        double u = 0.0 + (1.0 - 0.0) * mt_double(s);
        res[i] = res[i] + rint(scale * (2.0 * u - 1.0));
    }
    qsort(res, (size_t)K, sizeof(double), cmp_double);
    for (int i = 0; i < K; i++) a[i] = snap(q->theta, q->T, res[i]);
    /* RandomState.choice(betas, size=K, replace=True) = randint(0, n_beta): masked rejection */
    for (int i = 0; i < K; i++) b[i] = (int32_t)mt_interval(s, (uint32_t)(q->n_beta - 1));
    draw_init_ws(s, K, q->max_unif_ws, w);
    draw_k_arr(s, K, n_round, ka);
    return 0;
}

/* ------------------------------------------------------------------ single-stream entry points */
void scape_host_mt_seed(uint32_t seed, uint32_t *state625) { mt_seed((mt_t *)state625, seed); }

double scape_host_mt_double(uint32_t *state625) { return mt_double((mt_t *)state625); }

int scape_host_init_ws(uint32_t *state625, int K, double max_unif_ws, double *w)
{
    if (K < 0 || K > SCAPE_HOST_MAX_K) return 1;
    draw_init_ws((mt_t *)state625, K, max_unif_ws, w);
    return 0;
}

int scape_host_k_arr(uint32_t *state625, int K, int n_round, int8_t *ka)
{
    if (K < 0 || K > SCAPE_HOST_MAX_K || n_round < 0) return 1;
    draw_k_arr((mt_t *)state625, K, n_round, ka);
    return 0;
}

int scape_host_init_job(uint32_t *state625, const double *peaks, const double *peak_w, int n_peak,
                        const double *theta, int T, int L, int n_beta, double shift_scale, double max_unif_ws,
                        int K, int n_round, int32_t *a, int32_t *b, double *w, int8_t *ka)
{
    utr_in q = {peaks, peak_w, theta, n_peak, T, L, n_beta, shift_scale, max_unif_ws};
    if (!utr_supported(&q, K) || n_round < 0) return 1;
    scratch sc;
    memset(&sc, 0, sizeof(sc));
    int rc = draw_init_job((mt_t *)state625, &q, K, n_round, &sc, a, b, w, ka);
    scratch_free(&sc);
    return rc;
}

/* One K sweep (run() :965 -> em_optim0 :846-871) continuing the caller's stream: K = n_max .. n_min, n_trial
 * restarts each, rows in that order, pitch kmax (w: kmax+1).  Nothing is drawn when it returns 1. */
int scape_host_sweep(uint32_t *state625, const double *peaks, const double *peak_w, int n_peak,
                     const double *theta, int T, int L, int n_beta, double shift_scale, double max_unif_ws,
                     int n_max, int n_min, int n_trial, int n_round, int kmax,
                     int32_t *jk, int32_t *a, int32_t *b, double *w, int8_t *ka)
{
    utr_in q = {peaks, peak_w, theta, n_peak, T, L, n_beta, shift_scale, max_unif_ws};
    if (n_min < 1 || n_max < n_min || n_max > kmax || kmax > SCAPE_HOST_MAX_K || n_trial < 1 || n_round < 0) return 1;
    for (int K = n_max; K >= n_min; K--)
        if (!utr_supported(&q, K)) return 1;
    scratch sc;
    memset(&sc, 0, sizeof(sc));
    mt_t s;
    memcpy(&s, state625, sizeof(s));
    int rc = 0;
    size_t j = 0;
    for (int K = n_max; K >= n_min && !rc; K--)
        for (int t = 0; t < n_trial && !rc; t++, j++) {
            jk[j] = K;
            rc = draw_init_job(&s, &q, K, n_round, &sc, a + j * kmax, b + j * kmax, w + j * (kmax + 1), ka + j * n_round);
        }
    scratch_free(&sc);
    if (!rc) memcpy(state625, &s, sizeof(s));
    return rc;
}

/* Many sweeps at once, one per item, every item with its OWN generator state (independent streams: the current
 * UTRs of different chunk files); items are spread over host threads. */
typedef struct {
    struct scape_host_sweep_item *items;
    int n, n_trial, n_round, next;
    pthread_mutex_t mu;
} sweep_shared;

static void *sweep_worker(void *arg)
{
    sweep_shared *S = (sweep_shared *)arg;
    for (;;) {
        pthread_mutex_lock(&S->mu);
        int i = S->next++;
        pthread_mutex_unlock(&S->mu);
        if (i >= S->n) break;
        struct scape_host_sweep_item *it = &S->items[i];
        it->status = scape_host_sweep(it->state625, it->peaks, it->peak_w, it->n_peak, it->theta, it->T, it->L,
                                      it->n_beta, it->shift_scale, it->max_unif_ws, it->n_max, it->n_min, S->n_trial,
                                      S->n_round, it->kmax, it->jk, it->a, it->b, it->w, it->ka);
    }
    return NULL;
}

int scape_host_sweep_batch(struct scape_host_sweep_item *items, int n, int n_trial, int n_round, int n_threads)
{
    if (!items || n < 0) return 1;
    sweep_shared S;
    S.items = items;
    S.n = n;
    S.n_trial = n_trial;
    S.n_round = n_round;
    S.next = 0;
    pthread_mutex_init(&S.mu, NULL);
    int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if (nt > n) nt = n > 0 ? n : 1;
    pthread_t th[64];
    int started = 0;
    for (int i = 1; i < nt; i++)
        if (pthread_create(&th[started], NULL, sweep_worker, &S) == 0) started++;
    sweep_worker(&S);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&S.mu);
    return 0;
}

/* ------------------------------------------------------------------ whole-batch plan, threaded over UTRs */
typedef struct {
    const struct scape_host_plan_args *A;
    int next;                                       /* work counter (guarded by mu) */
    pthread_mutex_t mu;
} plan_shared;

static void plan_one(const struct scape_host_plan_args *A, int u, scratch *sc)
{
    utr_in q = {A->peaks + A->peak_off[u], A->peak_w + A->peak_off[u], A->theta + A->theta_off[u],
                (int)(A->peak_off[u + 1] - A->peak_off[u]), (int)(A->theta_off[u + 1] - A->theta_off[u]),
                A->L[u], A->n_beta[u], A->shift_scale[u], A->max_unif_ws[u]};
    int kmax = A->kmax, kcap = A->kcap, nr = A->n_round;
    int nmax = A->n_max[u], nmin = A->n_min[u];
    A->status[u] = 1;
    if (nmin < 1 || nmax < nmin || nmax > kmax || nmax > kcap) return;
    if ((int64_t)(nmax - nmin + 1) * A->n_trial != A->spans[u + 1] - A->spans[u]) return;
    for (int K = nmax; K >= nmin; K--)
        if (!utr_supported(&q, K)) return;
    mt_t s;
    mt_seed(&s, A->seeds[u]);
    int64_t j = A->spans[u];
    for (int K = nmax; K >= nmin; K--)
        for (int t = 0; t < A->n_trial; t++, j++) {
            A->ju[j] = u;
            A->jk[j] = K;
            if (draw_init_job(&s, &q, K, nr, sc, A->a + j * kmax, A->b + j * kmax, A->w + j * (kmax + 1),
                              A->ka + j * nr))
                return;
        }
    memcpy(A->states + (size_t)u * 625, &s, sizeof(s));
    /* rm_component's draws (init_ws + gen_k_arr, :843 -> :709 -> :720) for every possible K' < n_max,
       each continuing from the state the sweep left behind */
    for (int Kp = 0; Kp < nmax; Kp++) {
        mt_t c = s;
        size_t o = (size_t)u * kcap + Kp;
        draw_init_ws(&c, Kp, q.max_unif_ws, A->prune_w + o * (kcap + 1));
        draw_k_arr(&c, Kp, nr, A->prune_ka + o * nr);
        memcpy(A->prune_states + o * 625, &c, sizeof(c));
    }
    A->status[u] = 0;
}

static void *plan_worker(void *arg)
{
    plan_shared *S = (plan_shared *)arg;
    scratch sc;
    memset(&sc, 0, sizeof(sc));
    for (;;) {
        pthread_mutex_lock(&S->mu);
        int lo = S->next;
        S->next += 8;
        pthread_mutex_unlock(&S->mu);
        if (lo >= S->A->n_utr) break;
        int hi = lo + 8 < S->A->n_utr ? lo + 8 : S->A->n_utr;
        for (int u = lo; u < hi; u++) plan_one(S->A, u, &sc);
    }
    scratch_free(&sc);
    return NULL;
}

int scape_host_plan(const struct scape_host_plan_args *A)
{
    if (!A || A->n_utr < 0 || A->kmax < 1 || A->kmax > SCAPE_HOST_MAX_K || A->kcap < 1 || A->n_trial < 1
        || A->n_round < 0)
        return 1;
    plan_shared S;
    S.A = A;
    S.next = 0;
    pthread_mutex_init(&S.mu, NULL);
    int nt = A->n_threads < 1 ? 1 : (A->n_threads > 64 ? 64 : A->n_threads);
    if (nt > (A->n_utr + 7) / 8) nt = (A->n_utr + 7) / 8;
    pthread_t th[64];
    int started = 0;
    for (int i = 1; i < nt; i++)
        if (pthread_create(&th[started], NULL, plan_worker, &S) == 0) started++;
    plan_worker(&S);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&S.mu);
    return 0;
}

int scape_host_abi_version(void) { return SCAPE_HOST_ABI_VERSION; }
