"""CPU restatement of the reference's ``infer_pa`` path (numpy + oracle/scape_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product (``scape_amd/``) never imports it.

Each function cites the reference lines it restates (paths relative to
``/root/reference/src/scape/``).  Parity status: pinned - see
``tests/test_oracle_golden.py`` (reference example fixtures + traces of the
reference's own Python code + scalar known answers of ``taichi_code_test.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np
from scipy.signal import find_peaks

SENT = float(np.finfo("f").min)          # apa_core.py:428, taichi_core.py:8
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_D = ctypes.POINTER(ctypes.c_double)
_I64 = ctypes.POINTER(ctypes.c_int64)
_I32 = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "libscape_oracle.so")
    src = os.path.join(_HERE, "scape_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libscape_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for name in ("so_my_log", "so_loglik_r_s", "so_lik_r_s"):
            getattr(_LIB, name).restype = ctypes.c_double
        for name, nargs in (("so_my_log", 1), ("so_logpdf_normal", 3), ("so_pdf_normal", 3),
                            ("so_loglik_l_xt", 3), ("so_lik_l_xt", 3), ("so_loglik_x_st_pa", 3),
                            ("so_loglik_x_st", 5), ("so_lik_x_st", 5), ("so_loglik_r_s", 2),
                            ("so_lik_r_s", 2)):
            f = getattr(_LIB, name)
            f.restype = ctypes.c_double
            f.argtypes = [ctypes.c_double] * nargs
        _LIB.so_logsumexp.restype = ctypes.c_double
        _LIB.so_logsumexp.argtypes = [_D, ctypes.c_int]
        _LIB.so_np_sum.restype = ctypes.c_double
        _LIB.so_np_sum.argtypes = [_D, ctypes.c_long]
        _LIB.so_em_algo.restype = ctypes.c_int
        _LIB.so_utr_jobs.restype = ctypes.c_int
    return _LIB


def _p(a, t=_D):
    return a.ctypes.data_as(t)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------------------
# L3 operators (taichi_core.py:183-246) - same signatures as the reference seam
# ---------------------------------------------------------------------------
def loglik_xlr_t_pa(x_arr, l_arr, pa_arr, theta, sigma_f):
    x, l, pa = _f64(x_arr), _f64(l_arr), _f64(pa_arr)
    out = np.zeros(len(x))
    lib().so_loglik_xlr_t_pa(_p(x), _p(l), _p(pa), ctypes.c_int(len(x)), ctypes.c_double(theta),
                             ctypes.c_double(sigma_f), _p(out))
    return out


def loglik_xlr_t_r_known(x_arr, l_arr, r_arr, s_dis_arr, pmf_s_arr, theta, mu_f, sigma_f):
    x, l, r, s, pmf = map(_f64, (x_arr, l_arr, r_arr, s_dis_arr, pmf_s_arr))
    out = np.zeros(len(x))
    lib().so_loglik_xlr_t_r_known(_p(x), _p(l), _p(r), ctypes.c_int(len(x)), _p(s), _p(pmf),
                                  ctypes.c_int(len(s)), ctypes.c_double(theta), ctypes.c_double(mu_f),
                                  ctypes.c_double(sigma_f), _p(out))
    return out


def loglik_xlr_t_r_unknown(x_arr, l_arr, r_arr, s_dis_arr, pmf_s_arr, theta, mu_f, sigma_f):
    x, l, s, pmf = map(_f64, (x_arr, l_arr, s_dis_arr, pmf_s_arr))
    out = np.zeros(len(x))
    lib().so_loglik_xlr_t_r_unknown(_p(x), _p(l), ctypes.c_int(len(x)), _p(s), _p(pmf),
                                    ctypes.c_int(len(s)), ctypes.c_double(theta), ctypes.c_double(mu_f),
                                    ctypes.c_double(sigma_f), _p(out))
    return out


def get_loglik_marginal_tensor(all_theta, predef_beta_arr, loglik_xlr_t_arr):
    th, be, A = _f64(all_theta), _f64(predef_beta_arr), _f64(loglik_xlr_t_arr)
    N, T = A.shape
    M = np.zeros((T, len(be), N))
    lib().so_marginal_tensor(_p(th), ctypes.c_int(T), _p(be), ctypes.c_int(len(be)), _p(A),
                             ctypes.c_int(N), _p(M))
    return M


def phase_a(x, l, r, pa, all_theta, s_dis, pmf, mu_f, sigma_f):
    x, l, r, pa, th, s, pmf = map(_f64, (x, l, r, pa, all_theta, s_dis, pmf))
    A = np.zeros((len(x), len(th)))
    lib().so_phase_a(_p(x), _p(l), _p(r), _p(pa), ctypes.c_int(len(x)), _p(th), ctypes.c_int(len(th)),
                     _p(s), _p(pmf), ctypes.c_int(len(s)), ctypes.c_double(mu_f),
                     ctypes.c_double(sigma_f), _p(A))
    return A


# numpy twins of Phase A / Phase B used to cross-check the C code
def phase_a_np(x, l, r, pa, all_theta, s_dis, pmf, mu_f, sigma_f):
    x, l, r, pa, th = map(_f64, (x, l, r, pa, all_theta))
    with np.errstate(all="ignore"):
        u = th[None, :] - x[:, None]
        ok = l[:, None] <= u
        ll_l = np.where(ok, -np.log(np.where(ok, u, 1.0)), SENT)
        lk_l = np.where(ok, 1.0 / np.where(ok, u, 1.0), 0.0)
        A = np.empty_like(u)
        is_pa = ~np.isnan(pa)
        is_rk = ~np.isnan(r) & ~is_pa
        is_ru = ~is_pa & ~is_rk
        z = (pa[:, None] - th[None, :]) / sigma_f
        A_pa = ll_l + (-0.5 * z ** 2 - np.log(sigma_f) - 0.5 * np.log(2 * np.pi))
        v = np.zeros_like(u)
        for sj, pj in zip(s_dis, pmf):
            zz = (x[:, None] - (th[None, :] + sj - mu_f)) / sigma_f
            v = v + 1 / sj * (np.exp(-0.5 * zz ** 2) / np.sqrt(2 * np.pi) / sigma_f) * lk_l * pj
        v = np.where(v < 1e-300, 0.0, v)
        A_ru = np.where(v <= 0, SENT, np.log(np.where(v <= 0, 1.0, v)))
        A[is_pa] = A_pa[is_pa]
        A[is_ru] = A_ru[is_ru]
        for n in np.where(is_rk)[0]:
            for t in range(len(th)):
                A[n, t] = loglik_xlr_t_r_known(x[n:n + 1], l[n:n + 1], r[n:n + 1], s_dis, pmf, th[t],
                                               mu_f, sigma_f)[0]
    return A


def phase_b_np(all_theta, betas, A, rows=None):
    th = _f64(all_theta)
    rows = range(len(th)) if rows is None else rows
    out = np.empty((len(rows), len(betas), A.shape[0]))
    with np.errstate(all="ignore"):
        for ii, i in enumerate(rows):
            for j, b in enumerate(betas):
                lo = np.searchsorted(th, th[i] - 3 * b, "left")
                hi = np.searchsorted(th, th[i] + 3 * b, "right") - 1
                g = -0.5 * ((th[lo:hi + 1] - th[i]) / b) ** 2 - np.log(b) - 0.5 * np.log(2 * np.pi)
                G = np.log(np.sum(np.exp(g)))
                t = A[:, lo:hi + 1] + g[None, :] - G
                mx = t.max(axis=1)
                out[ii, j] = np.log(np.exp(t - mx[:, None]).sum(axis=1)) + mx
    return out


# ---------------------------------------------------------------------------
# host-level restatement (apa_core.py)
# ---------------------------------------------------------------------------
def bin_reads(x, l, r, pa, steps=(5, 10, 10, 5)):
    """apa_core.py:285-327."""
    cols = [np.asarray(c, dtype=np.float64) for c in (x, l, r, pa)]
    labels = []
    for v, step in zip(cols, steps):
        with np.errstate(all="ignore"):
            top = np.nanmax(v) if not np.all(np.isnan(v)) else np.nan
        edges = np.array([0, step]) if np.isnan(top) else np.arange(0, step + top, step)
        labels.append(np.digitize(np.where(np.isnan(v), -1, v), edges, right=False))
    mat = np.column_stack(labels)
    _, idx, cnt = np.unique(mat, axis=0, return_inverse=True, return_counts=True)
    idx = idx.reshape(-1)
    binned = [np.bincount(idx, v) / cnt for v in cols]
    return binned[0], binned[1], binned[2], binned[3], cnt, idx


def smooth(y, bw):
    """ApaModel.ker_smooth, apa_core.py:680-700."""
    ny = len(y)
    w = np.exp(-np.arange(-3 * bw, 3 * bw + 1) ** 2 / (2 * bw * bw))
    half = int(3 * bw)
    out = np.zeros_like(y, dtype=np.float64)
    for i in range(ny):
        lo, hi = i - half, i + half
        a, b = max(lo, 0), min(hi, ny - 1)
        ww = w[a - lo:b - lo + 1]
        out[i] = np.sum(ww * y[a:b + 1]) / np.sum(ww) if (lo < 0 or hi >= ny) else np.sum(w * y[lo:hi + 1]) / np.sum(w)
    return out


def coverage_profile(bx, bl, cnt, L, beta_step):
    """ApaModel._get_coverage_profile, apa_core.py:454-462."""
    cov = np.zeros(int(L))
    for xi, li, ci in zip(bx, bl, cnt):
        cov[int(xi):int(xi) + int(li)] += ci
    xs = np.concatenate([np.arange(-100, 0), np.arange(int(L)), int(L) + np.arange(100)])
    ys = smooth(np.concatenate([np.zeros(100), cov, np.zeros(100)]), bw=beta_step * 3)
    return xs, ys


def nearest_on_grid(grid, vals):
    """ApaModel.find_nearest, apa_core.py:535-549 (ties go to the upper grid point)."""
    idx = np.searchsorted(grid, vals, side="left")
    out = idx.copy()
    for i, j in enumerate(idx):
        if j == 0:
            continue
        if j == len(grid):
            out[i] = len(grid) - 1
        elif vals[i] - grid[j - 1] >= grid[j] - vals[i]:
            out[i] = j
        else:
            out[i] = j - 1
    return out, grid[out]


def gen_k_arr(K, n):
    """ApaModel.gen_k_arr, apa_core.py:653-677 (the no-repeat swap can never fire)."""
    if K <= 1:
        return np.zeros(n, dtype="int")
    arr = np.random.permutation(K)
    out = []
    for t in range(n):
        if t % K == 0:
            np.random.shuffle(arr)
        out.append(arr[t % K])
    return np.array(out, dtype="int")


@dataclass
class Para:
    """Mirror of the fields em_algo touches on ``Parameters`` (apa_core.py:236-258)."""
    alpha_arr: np.ndarray
    beta_arr: np.ndarray
    ws: np.ndarray
    K: int
    bic: float = np.nan
    lb_arr: list = field(default_factory=list)
    title: str = ""


class Model:
    """Per-UTR state of ApaModel (apa_core.py:333-437) - grids, binned data, tensors."""

    def __init__(self, x, l, r, pa, n_max_apa=5, n_min_apa=1, utr_length=2000, min_LA=20, max_LA=150,
                 mu_f=300, sigma_f=50, min_pa_gap=100, max_beta=70, theta_step=9, beta_step=5,
                 min_ws=0.05, max_unif_ws=0.15, **_ignored):
        self.n_max_apa, self.n_min_apa = n_max_apa, n_min_apa
        self.x, self.l, self.r, self.pa, self.cnt, self.idx = bin_reads(x, l, r, pa)
        self.N = len(self.cnt)
        self.L = utr_length if utr_length > 2000 else 2000
        assert all(0 <= v < utr_length for v in self.x)
        self.s_dis = np.arange(min_LA, max_LA, 10)
        self.pmf = np.repeat(1 / len(self.s_dis), len(self.s_dis))
        self.pmf = self.pmf / sum(self.pmf)
        self.max_LA = max_LA
        self.mu_f, self.sigma_f = mu_f, sigma_f
        self.min_pa_gap, self.max_beta, self.theta_step, self.beta_step = min_pa_gap, max_beta, theta_step, beta_step
        self.min_theta = int(min(self.l)) + 0.0
        self.all_theta = np.arange(int(self.min_theta), int(self.L), int(theta_step)) + 0.0
        self.betas = np.arange(beta_step, max_beta, beta_step) + 0.0
        self.min_ws, self.max_unif_ws = min_ws, max_unif_ws
        self.nround = 50
        self.unif_ll = float(np.log((1 / self.L) * (1 / self.L) * (1 / max_LA)))   # lik_f0 :576-584
        self.A = self.M = self.cov = None
        self.calls = []                      # trace of em_algo calls (for tests)

    # -- phases A/B (apa_core.py:951-959) --
    def build(self):
        self.cov = coverage_profile(self.x, self.l, self.cnt, self.L, self.beta_step)
        self.A = phase_a(self.x, self.l, self.r, self.pa, self.all_theta, self.s_dis, self.pmf,
                         self.mu_f, self.sigma_f)
        self.M = get_loglik_marginal_tensor(self.all_theta, self.betas, self.A)

    # -- init sampling (apa_core.py:781-829) --
    def sample_alpha(self, K):
        xs, ys = self.cov
        pk, _ = find_peaks(ys, distance=self.min_pa_gap)
        peaks = xs[pk]
        bw = self.beta_step * 3
        pw = np.array([sum(ys[i - bw:i + bw + 1]) for i in pk], dtype=np.float64)
        pw = pw / sum(pw)
        if K <= len(pk):
            res = np.random.choice(peaks, size=K, replace=False, p=pw)
        else:
            res = np.random.choice(self.L, size=K - len(pk), replace=False)
            res = np.concatenate((peaks, res))
        shift = np.rint(5 * self.beta_step * (2 * np.random.uniform(low=0.0, high=1.0, size=K) - 1))
        res = np.sort(res + shift)
        return nearest_on_grid(self.all_theta, res)[1]

    def init_ws(self, K):
        ws = np.random.uniform(size=(K + 1))
        ws = ws / sum(ws)
        if ws[-1] > self.max_unif_ws:
            ws[:-1] = ws[:-1] * (1 - self.max_unif_ws)
            ws[-1] = self.max_unif_ws
        return ws

    def init_para(self, K):
        a = self.sample_alpha(K)
        b = np.random.choice(self.betas, size=K, replace=True)
        return Para(alpha_arr=a, beta_arr=b, ws=self.init_ws(K), K=K)

    # -- EM (apa_core.py:714-779) through the C restatement --
    def em_algo(self, para, fixed=False):
        K = para.K
        if K > 63:      # the C restatement keeps per-component scratch in 64-entry arrays (a NaN likelihood makes the re-run loop grow K without end)
            raise RuntimeError(f"oracle: K = {K} is beyond the restatement's 64 columns")
        k_arr = np.ascontiguousarray(gen_k_arr(K, self.nround), dtype=np.int64)
        a = _f64(para.alpha_arr).copy()
        b = _f64(para.beta_arr).copy()
        w = _f64(para.ws).copy()
        rec = dict(K=K, fixed=fixed, a0=a.copy(), b0=b.copy(), w0=w.copy(), k_arr=k_arr.copy())
        bic = ctypes.c_double(0)
        lb = np.zeros(self.nround)
        th, be, cnt = _f64(self.all_theta), _f64(self.betas), _f64(self.cnt)
        n_lb = lib().so_em_algo(_p(self.M), ctypes.c_int(len(th)), ctypes.c_int(len(be)),
                                ctypes.c_int(self.N), _p(th), _p(be), _p(cnt),
                                ctypes.c_double(self.unif_ll), ctypes.c_double(self.L),
                                ctypes.c_double(self.min_theta), ctypes.c_double(self.max_unif_ws),
                                ctypes.c_int(K), _p(a), _p(b), _p(w), _p(k_arr, _I64),
                                ctypes.c_int(self.nround), ctypes.c_int(int(fixed)),
                                ctypes.byref(bic), _p(lb))
        para.alpha_arr = a.astype("int")
        para.beta_arr, para.ws = b, w
        para.bic = np.float64(bic.value)
        para.lb_arr = [np.float64(v) for v in lb[:n_lb]]
        rec.update(a1=a.copy(), b1=b.copy(), w1=w.copy(), bic=bic.value, lb=lb[:n_lb].copy())
        self.calls.append(rec)
        return para

    def em_optim0(self, K):                                       # apa_core.py:846-871
        res = [self.em_algo(self.init_para(K)) for _ in range(10)]
        # bic_arr = np.full(n_trial, np.finfo('f').max) (:849) is a FLOAT32 array: the arg-min over the
        # restarts sees the BICs rounded to f32 (ties -> first restart)
        return res[int(np.argmin(np.array([p.bic for p in res], dtype=np.float32)))]

    def rm_component(self, para):                                 # apa_core.py:832-844, :708-711
        keep = np.array([i for i in range(para.K) if not para.ws[i] < self.min_ws], dtype=int)
        if len(keep) == para.K:
            return para
        para.alpha_arr = para.alpha_arr[keep]
        para.beta_arr = para.beta_arr[keep]
        para.K = len(keep)
        para.ws = self.init_ws(para.K)
        return self.em_algo(para, fixed=True)

    def labels(self, para):                                       # apa_core.py:873-881
        lab = np.zeros(self.N, dtype=np.int64)
        th, be, cnt = _f64(self.all_theta), _f64(self.betas), _f64(self.cnt)
        a, b, w = _f64(para.alpha_arr), _f64(para.beta_arr), _f64(para.ws)
        lib().so_get_label(_p(self.M), ctypes.c_int(len(th)), ctypes.c_int(len(be)), ctypes.c_int(self.N),
                           _p(th), _p(be), _p(cnt), ctypes.c_double(self.unif_ll), ctypes.c_int(para.K),
                           _p(a), _p(b), _p(w), _p(lab, _I64))
        return lab

    def marginal_scipy(self, alpha, beta):
        """ApaModel.loglik_marginal_lxr (apa_core.py:642-651): the numpy/scipy flavour fixed_run uses."""
        from scipy import stats
        from scipy.special import logsumexp
        sel = np.where(np.logical_and(self.all_theta >= alpha - 3 * beta, self.all_theta <= alpha + 3 * beta))[0]
        logp = stats.norm(loc=alpha, scale=beta).logpdf(self.all_theta[sel])
        return logsumexp(self.A[:, sel] + logp[None, :], axis=1) - logsumexp(logp)

    def fixed_run(self, pre_para):                                # apa_core.py:883-928 (rm_comp_flag=False)
        full = np.arange(int(self.min_theta), int(self.L), int(self.theta_step)) + 0.0
        max_b, min_b = np.max(pre_para.beta_arr), np.min(pre_para.beta_arr)
        pieces = []
        for alpha in pre_para.alpha_arr:
            ind, _ = nearest_on_grid(self.all_theta, np.array([alpha - 3 * max_b, alpha + 3 * max_b]))
            pieces.append(full[ind[0]:ind[1]])
        self.all_theta = np.unique(np.concatenate(pieces))
        self.betas = np.arange(min_b, max_b + self.beta_step, self.beta_step) + 0.0
        self.A = phase_a(self.x, self.l, self.r, self.pa, self.all_theta, self.s_dis, self.pmf, self.mu_f, self.sigma_f)
        self.cov = coverage_profile(self.x, self.l, self.cnt, self.L, self.beta_step)
        self.M = np.zeros((len(self.all_theta), len(self.betas), self.N))
        for i, a in enumerate(self.all_theta):
            for j, b in enumerate(self.betas):
                self.M[i][j] = self.marginal_scipy(a, b)
        res = self.em_optim0(int(pre_para.K))
        res.label_arr = self.labels(res)[self.idx]
        res.title = "Final Result (subsample run)"
        return res

    def run(self, skip_lik=False):                                # apa_core.py:930-981
        if self.n_min_apa > self.n_max_apa:
            raise Exception("n_max_apa has to be greater than n_min_apa!")
        if self.max_beta < self.beta_step:
            raise Exception("max_beta has to be greater than beta_step_size!")
        if not skip_lik:
            self.build()
        res = [self.em_optim0(K) for K in range(self.n_max_apa, self.n_min_apa - 1, -1)]
        # bic_arr of run() is float32 as well (:945): first arg-min of the f32-rounded BICs
        best = res[int(np.argmin(np.array([p.bic for p in res], dtype=np.float32)))]
        best = self.rm_component(best)
        best.label_arr = self.labels(best)[self.idx]
        best.title = "Final Result"
        return best


def subsample_run(x, l, r, pa, re_run_mode=True, pre_para=None, **kw):
    """apa_core.py:984-1035; uses the global numpy RNG like the reference.  pre_para: fixed_run_mode."""
    utr_len = max(np.max(x) + np.max(l) + 50, kw.get("utr_length", -1))
    kw = dict(kw)
    kw["utr_length"] = utr_len
    kw.setdefault("n_max_apa", 5)
    if pre_para is not None:                                      # :999-1017
        kw["utr_length"] = max(utr_len, pre_para.L)
        model = Model(x, l, r, pa, **kw)
        return model.fixed_run(pre_para), model
    model = Model(x, l, r, pa, **kw)
    res = model.run()
    while re_run_mode and len(res.alpha_arr) == kw["n_max_apa"]:
        model.n_max_apa = kw["n_max_apa"] + 2
        model.n_min_apa = kw["n_max_apa"]
        kw["n_max_apa"] += 2
        res = model.run(skip_lik=True)
    return res, model
