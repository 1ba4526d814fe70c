/*
 * scape_oracle.c - CPU restatement of the reference's infer_pa arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under scape_amd/ (the product) may import,
 * link or call this file; it is used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / timed CPU baseline.
 *
 * Every function restates one reference function in scalar f64 C, in the
 * reference's own evaluation order (citations are to /root/reference/src/scape/).
 * Pinned by tests/test_oracle_golden.py against (1) the scalar known answers of
 * taichi_code_test.py:514-593, (2) the reference's example fixtures
 * (examples/<...>/pkl_output/<...>.res.pkl: K, alpha, beta, labels) and (3) traces of the
 * reference's own Python run in the build container (tests/golden/trace_*.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SENT (-3.4028234663852886e38) /* float(np.finfo('f').min), taichi_core.py:8, apa_core.py:428 */
#define PI_REF 3.141592653589793      /* taichi_core.py:9 */

/* ---- base functions (taichi_core.py:24-97) -------------------------------- */
double so_my_log(double x) { return (x <= 0.0) ? SENT : log(x); }                     /* :24-29 */
double so_logpdf_normal(double x, double mu, double sigma) {                         /* :31-33 */
    double z = (x - mu) / sigma;
    return -0.5 * (z * z) - log(sigma) - 0.5 * log(2 * PI_REF);
}
double so_pdf_normal(double x, double mu, double sigma) {                            /* :35-37 */
    double z = (x - mu) / sigma;
    return exp(-0.5 * (z * z)) / sqrt(2 * PI_REF) / sigma;
}
double so_logsumexp(const double *v, int n) {                                        /* :40-54 */
    double mx = v[0], sum = 0.0;
    for (int i = 0; i < n; i++) if (v[i] > mx) mx = v[i];
    for (int i = 0; i < n; i++) sum += exp(v[i] - mx);
    return log(sum) + mx;
}
double so_loglik_l_xt(double x, double l, double theta) {                            /* :56-62 */
    double u = theta - x;
    return (l <= u) ? -log(u) : SENT;
}
double so_lik_l_xt(double x, double l, double theta) {                               /* :64-70 */
    double u = theta - x;
    return (l <= u) ? 1 / u : 0.0;
}
double so_loglik_x_st_pa(double pa, double theta, double sigma_f) {                  /* :72-74 */
    return so_logpdf_normal(pa - theta, 0, sigma_f);
}
double so_loglik_x_st(double x, double s, double theta, double mu_f, double sigma_f) { /* :77-79 */
    return so_logpdf_normal(x, theta + s - mu_f, sigma_f);
}
double so_lik_x_st(double x, double s, double theta, double mu_f, double sigma_f) {  /* :81-83 */
    return so_pdf_normal(x, theta + s - mu_f, sigma_f);
}
double so_loglik_r_s(double r, double s) { return (r <= s) ? -log(s) : SENT; }       /* :85-90 */
double so_lik_r_s(double r, double s) { return (r <= s) ? 1 / s : 0.0; }             /* :92-97 */

/* ---- the three point-likelihood kernels (taichi_core.py:101-157) ----------- */
void so_loglik_xlr_t_pa(const double *x, const double *l, const double *pa, int n,
                        double theta, double sigma_f, double *out) {                 /* :101-107 */
    for (int i = 0; i < n; i++)
        out[i] = so_loglik_l_xt(x[i], l[i], theta) + so_loglik_x_st_pa(pa[i], theta, sigma_f);
}

void so_loglik_xlr_t_r_known(const double *x, const double *l, const double *r, int n,
                             const double *s, const double *pmf, int S,
                             double theta, double mu_f, double sigma_f, double *out) { /* :111-132 */
    double *tmp = (double *)malloc(sizeof(double) * (size_t)S);
    for (int i = 0; i < n; i++) {
        double tmpn = 0.0;
        for (int j = 0; j < S; j++) {
            if (s[j] < r[i]) { tmp[j] = SENT; continue; }
            tmpn += pmf[j];
            tmp[j] = so_loglik_r_s(r[i], s[j]) + so_loglik_x_st(x[i], s[j], theta, mu_f, sigma_f)
                     + so_loglik_l_xt(x[i], l[i], theta) + log(pmf[j]);
        }
        out[i] = so_logsumexp(tmp, S) - log(tmpn);
    }
    free(tmp);
}

void so_loglik_xlr_t_r_unknown(const double *x, const double *l, int n,
                               const double *s, const double *pmf, int S,
                               double theta, double mu_f, double sigma_f, double *out) { /* :141-157 */
    for (int i = 0; i < n; i++) {
        double v = 0.0;
        for (int j = 0; j < S; j++)
            v += 1 / s[j] * so_lik_x_st(x[i], s[j], theta, mu_f, sigma_f)
                 * so_lik_l_xt(x[i], l[i], theta) * pmf[j];
        if (v < 1e-300) v = 0.0;
        out[i] = so_my_log(v);
    }
}

/* ---- Phase A: A[n, t] over all theta (apa_core.py:954-957, :620-640, :439-452) */
void so_phase_a(const double *x, const double *l, const double *r, const double *pa, int N,
                const double *theta, int T, const double *s, const double *pmf, int S,
                double mu_f, double sigma_f, double *A /* [N*T] row-major */) {
    for (int n = 0; n < N; n++) {
        int has_pa = !isnan(pa[n]), has_r = !isnan(r[n]);
        for (int t = 0; t < T; t++) {
            double o;
            if (has_pa)      so_loglik_xlr_t_pa(x + n, l + n, pa + n, 1, theta[t], sigma_f, &o);
            else if (has_r)  so_loglik_xlr_t_r_known(x + n, l + n, r + n, 1, s, pmf, S, theta[t], mu_f, sigma_f, &o);
            else             so_loglik_xlr_t_r_unknown(x + n, l + n, 1, s, pmf, S, theta[t], mu_f, sigma_f, &o);
            A[(size_t)n * T + t] = o;
        }
    }
}

/* np.searchsorted on an ascending array */
static int ss_left(const double *a, int n, double v) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
static int ss_right(const double *a, int n, double v) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (v < a[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

/* ---- Phase B (taichi_core.py:160-246) --------------------------------------- */
void so_marginal_lxr(double alpha, double beta, const double *theta, int T,
                     const double *A, int N, double *out /* [N] */) {                /* :218-234 */
    int min_ind = ss_left(theta, T, alpha - 3 * beta);
    int max_ind = ss_right(theta, T, alpha + 3 * beta) - 1;
    int W = max_ind - min_ind + 1;
    double *logp = (double *)malloc(sizeof(double) * (size_t)(W > 0 ? W : 1));
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(W > 0 ? W : 1));
    double psum = 0.0;                                                               /* :160-169 */
    for (int i = 0; i < W; i++) {
        logp[i] = so_logpdf_normal(theta[i + min_ind], alpha, beta);
        psum += exp(logp[i]);
    }
    double lps = log(psum);
    for (int n = 0; n < N; n++) {                                                    /* :172-179 */
        for (int i = 0; i < W; i++) tmp[i] = A[(size_t)n * T + i + min_ind] + logp[i] - lps;
        out[n] = so_logsumexp(tmp, W);
    }
    free(logp); free(tmp);
}

void so_marginal_tensor(const double *theta, int T, const double *betas, int B,
                        const double *A, int N, double *M /* [T*B*N] */) {           /* :237-246 */
    for (int i = 0; i < T; i++)
        for (int j = 0; j < B; j++)
            so_marginal_lxr(theta[i], betas[j], theta, T, A, N, M + ((size_t)i * B + j) * N);
}

/* ---- numpy's pairwise summation (what np.sum does on a contiguous f64 run) -- */
static double np_pairwise_sum(const double *a, long n) {
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}
/* np.sum(a) for a 1-D contiguous array: numpy's reduction iterator hands the inner loop at most
   8192 elements at a time (its buffer size), each piece is summed pairwise and added on */
static double np_sum(const double *a, long n) {
    if (n <= 8192) return np_pairwise_sum(a, n);
    double acc = np_pairwise_sum(a, 8192);
    for (long o = 8192; o < n; o += 8192) acc += np_pairwise_sum(a + o, (n - o < 8192) ? n - o : 8192);
    return acc;
}
double so_np_sum(const double *a, long n) { return np_sum(a, n); }

/* ---- EM (apa_core.py:473-573, :702-779) ------------------------------------- */
typedef struct {
    const double *M; int T, B, N;
    const double *theta, *betas, *cnt;
    double unif_ll, L, min_theta, max_unif_ws;
} so_model;

static void cal_z_k(const so_model *m, int K, const double *alpha, const double *beta,
                    const double *ws, int k, double *logz /* [N,(K+1)] */) {         /* :473-488 */
    int C = K + 1;
    double lw = (ws[k] <= 0.0) ? SENT : log(ws[k]);
    if (k < K) {
        int ai = ss_left(m->theta, m->T, alpha[k]);
        int bi = ss_left(m->betas, m->B, beta[k]);
        const double *row = m->M + ((size_t)ai * m->B + bi) * m->N;
        for (int n = 0; n < m->N; n++) logz[(size_t)n * C + k] = lw + row[n];
    } else {
        for (int n = 0; n < m->N; n++) logz[(size_t)n * C + k] = lw + m->unif_ll;
    }
}

static void norm_z(const so_model *m, int C, const double *logz, double *Z) {        /* :490-495 */
    for (int n = 0; n < m->N; n++) {
        const double *lz = logz + (size_t)n * C;
        double *z = Z + (size_t)n * C;
        double mx = lz[0];
        for (int c = 1; c < C; c++) if (lz[c] > mx) mx = lz[c];
        for (int c = 0; c < C; c++) z[c] = exp((lz[c] - mx) * m->cnt[n]);
        double s = np_pairwise_sum(z, C);
        for (int c = 0; c < C; c++) z[c] = z[c] / s;
    }
}

static double exp_log_lik(const so_model *m, int C, const double *logz, const double *Z,
                          double *scratch) {                                         /* :570-573 */
    long cntnz = 0;
    for (int n = 0; n < m->N; n++)
        for (int c = 0; c < C; c++) {
            double z = Z[(size_t)n * C + c];
            if (z != 0) scratch[cntnz++] = (z * m->cnt[n]) * logz[(size_t)n * C + c];
        }
    return np_sum(scratch, cntnz);
}

static double elbo(const so_model *m, int C, const double *logz, const double *Z,
                   double *scratch, double *scratch2) {                              /* :559-561 */
    double ell = exp_log_lik(m, C, logz, Z, scratch);
    /* scipy.stats.entropy(Z, axis=1): pk = Z / sum(Z, axis=1); sum(entr(pk)) */
    for (int n = 0; n < m->N; n++) {
        const double *z = Z + (size_t)n * C;
        double s = np_pairwise_sum(z, C);
        double e[64];
        for (int c = 0; c < C; c++) {
            double pk = z[c] / s;
            e[c] = (pk > 0) ? -pk * log(pk) : ((pk == 0) ? 0.0 : -INFINITY);
        }
        scratch2[n] = m->cnt[n] * np_pairwise_sum(e, C);
    }
    return ell + np_sum(scratch2, m->N);
}

/*
 * One em_algo call (apa_core.py:714-779) from a given init and a given k_arr.
 * alpha/beta/ws are updated in place to the returned (sorted) parameters;
 * alpha is returned already rounded (np.rint) as f64.
 * Returns the number of rounds executed (= len(lb_arr)).
 */
int so_em_algo(const double *M, int T, int B, int N, const double *theta, const double *betas,
               const double *cnt, double unif_ll, double L, double min_theta, double max_unif_ws,
               int K, double *alpha, double *beta, double *ws, const int64_t *k_arr, int nround,
               int fixed, double *bic_out, double *lb_arr) {
    so_model m = {M, T, B, N, theta, betas, cnt, unif_ll, L, min_theta, max_unif_ws};
    int C = K + 1;
    double *logz = (double *)calloc((size_t)N * C, sizeof(double));
    double *Z = (double *)calloc((size_t)N * C, sizeof(double));
    double *scratch = (double *)malloc(sizeof(double) * (size_t)N * C);
    double *scratch2 = (double *)malloc(sizeof(double) * (size_t)N);
    double *col = (double *)malloc(sizeof(double) * (size_t)N);
    double lb = SENT;
    int n_lb = 0;

    for (int k = 0; k < C; k++) cal_z_k(&m, K, alpha, beta, ws, k, logz);           /* :722-724 */

    for (int it = 0; it < nround; it++) {
        int k = (int)k_arr[it];
        cal_z_k(&m, K, alpha, beta, ws, k, logz);                                    /* :731 */
        norm_z(&m, C, logz, Z);                                                      /* :733 */

        /* mstep / mstep_fixed (:525-533, :552-557) */
        for (int n = 0; n < N; n++) col[n] = Z[(size_t)n * C + k];
        if (np_sum(col, N) < 1e-8)
            for (int n = 0; n < N; n++) Z[(size_t)n * C + k] += 1e-8;
        /* maximize_ws (:498-505): ws = cnt @ Z ; ws /= sum */
        for (int c = 0; c < C; c++) {
            double acc = 0.0;
            for (int n = 0; n < N; n++) acc += cnt[n] * Z[(size_t)n * C + c];
            ws[c] = acc;
        }
        {
            double s = np_pairwise_sum(ws, C);
            for (int c = 0; c < C; c++) ws[c] = ws[c] / s;
            if (ws[K] > max_unif_ws) {
                double sk = np_pairwise_sum(ws, K);
                for (int c = 0; c < K; c++) ws[c] = (1 - max_unif_ws) * ws[c] / sk;
                ws[K] = max_unif_ws;
            }
        }
        if (!fixed) {                                                                /* :507-523 */
            double lo = (k == 0) ? min_theta : alpha[k - 1];
            double hi = (k == K - 1) ? L : alpha[k + 1];
            double lw = (ws[k] <= 0.0) ? SENT : log(ws[k]);
            double best = 0; int have = 0, best_a = 0, best_b = 0;
            for (int a = 0; a < T; a++) {
                if (!(theta[a] >= lo && theta[a] <= hi)) continue;
                for (int b = 0; b < B; b++) {
                    const double *row = M + ((size_t)a * B + b) * N;
                    for (int n = 0; n < N; n++)
                        scratch[n] = ((lw + row[n]) * Z[(size_t)n * C + k]) * cnt[n];
                    double sc = np_sum(scratch, N);
                    if (!have || sc > best) { best = sc; best_a = a; best_b = b; have = 1; }
                }
            }
            /* an empty window makes the reference raise (max() of an empty dict); cannot
               happen for alpha on the grid, keep parameters unchanged if it ever does */
            if (have) { alpha[k] = theta[best_a]; beta[k] = betas[best_b]; }
        }
        double lb_new = elbo(&m, C, logz, Z, scratch, scratch2);                     /* :740 */
        lb_arr[n_lb++] = lb_new;
        if (fabs(lb_new - lb) < fabs(1e-6 * lb)) break;                              /* :743 */
        lb = lb_new;
    }
    *bic_out = -2 * exp_log_lik(&m, C, logz, Z, scratch) + (3 * K + 1) * log((double)N); /* :702-706 */

    /* sort by alpha (stable: numpy's argsort uses insertion sort below 17 elements) (:768-772) */
    {
        int idx[64];
        double a2[64], b2[64], w2[64];
        for (int i = 0; i < K; i++) idx[i] = i;
        for (int i = 1; i < K; i++) {
            int v = idx[i], j = i - 1;
            while (j >= 0 && alpha[idx[j]] > alpha[v]) { idx[j + 1] = idx[j]; j--; }
            idx[j + 1] = v;
        }
        for (int i = 0; i < K; i++) { a2[i] = rint(alpha[idx[i]]); b2[i] = beta[idx[i]]; w2[i] = ws[idx[i]]; }
        for (int i = 0; i < K; i++) { alpha[i] = a2[i]; beta[i] = b2[i]; ws[i] = w2[i]; }
    }
    free(logz); free(Z); free(scratch); free(scratch2); free(col);
    return n_lb;
}

/* get_label (apa_core.py:873-881): per-bin arg-max of the responsibilities */
void so_get_label(const double *M, int T, int B, int N, const double *theta, const double *betas,
                  const double *cnt, double unif_ll, int K, const double *alpha, const double *beta,
                  const double *ws, int64_t *label) {
    so_model m = {M, T, B, N, theta, betas, cnt, unif_ll, 0, 0, 0};
    int C = K + 1;
    double *logz = (double *)calloc((size_t)N * C, sizeof(double));
    double *Z = (double *)calloc((size_t)N * C, sizeof(double));
    for (int k = 0; k < C; k++) cal_z_k(&m, K, alpha, beta, ws, k, logz);
    norm_z(&m, C, logz, Z);
    for (int n = 0; n < N; n++) {
        int best = 0;
        for (int c = 1; c < C; c++) if (Z[(size_t)n * C + c] > Z[(size_t)n * C + best]) best = c;
        label[n] = best;
    }
    free(logz); free(Z);
}

/*
 * Whole hot path for one UTR on the CPU (used as bench.py's cpu_baseline "port"):
 * Phase A, Phase B, then n_jobs em_algo calls from given init tables.
 * init tables are [n_jobs, KMAX] / [n_jobs, KMAX+1] / [n_jobs, nround], padded.
 */
int so_utr_jobs(const double *x, const double *l, const double *r, const double *pa,
                const double *cnt, int N, const double *theta, int T, const double *betas, int B,
                const double *s, const double *pmf, int S, double mu_f, double sigma_f,
                double unif_ll, double L, double min_theta, double max_unif_ws,
                int n_jobs, int KMAX, const int32_t *job_K, const int32_t *job_fixed,
                double *alpha, double *beta, double *ws, const int64_t *k_arr, int nround,
                double *bic, double *lb_arr, int32_t *n_lb) {
    double *A = (double *)malloc(sizeof(double) * (size_t)N * T);
    double *M = (double *)malloc(sizeof(double) * (size_t)T * B * N);
    if (!A || !M) { free(A); free(M); return -1; }
    so_phase_a(x, l, r, pa, N, theta, T, s, pmf, S, mu_f, sigma_f, A);
    so_marginal_tensor(theta, T, betas, B, A, N, M);
    for (int j = 0; j < n_jobs; j++)
        n_lb[j] = so_em_algo(M, T, B, N, theta, betas, cnt, unif_ll, L, min_theta, max_unif_ws,
                             job_K[j], alpha + (size_t)j * KMAX, beta + (size_t)j * KMAX,
                             ws + (size_t)j * (KMAX + 1), k_arr + (size_t)j * nround, nround,
                             job_fixed[j], bic + j, lb_arr + (size_t)j * nround);
    free(A); free(M);
    return 0;
}
