"""Batched driver of the ``infer_pa`` hot path on one MI355X.

For a list of prepared UTRs it packs the ragged arrays, runs Phase A/B and the EM jobs on
the GPU through the C-ABI and applies the reference's model selection on the host:
``em_optim0`` (apa_core.py:846-871), the K sweep and BIC arg-min of ``run`` (:965-973),
``rm_component`` (:832-844), labels (:976) and the re-run loop of ``subsample_run``
(:1023-1030).  No likelihood or EM arithmetic happens here.

RNG modes
---------
``reference``  one legacy ``RandomState(seed)`` for the whole chunk, consumed UTR by UTR in
               the reference's order (``np.random.seed(1)``, apa_core.py:125).  Results
               follow the reference's random stream exactly; EM jobs of one UTR run
               together, UTRs run one after another because UTR i+1's draws depend on
               UTR i's outcome (prune / re-run).
``per_utr``    every UTR gets ``RandomState(seed + index)``; all UTRs of a wave advance in
               lock-step (one launch per stage).  Statistically equivalent, fully batched.
"""
from __future__ import annotations

import contextlib
import ctypes
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import Params, check, f64, i32, ptr, P_i32, P_i64, P_i8
from . import _hostlib
from .host import N_ROUND, N_TRIAL, FastSampler, Sampler, UtrPrep


@dataclass
class Fit:
    """Result of one em_algo call on the host side (indices into the UTR's grids)."""
    K: int
    a_idx: np.ndarray
    b_idx: np.ndarray
    ws: np.ndarray
    bic: float
    lb: np.ndarray


@dataclass
class UtrResult:
    prep: UtrPrep
    fit: Fit = None
    labels_bin: np.ndarray = None
    n_jobs: int = 0


@dataclass
class _Job:
    u: int              # UTR index within the loaded batch
    K: int
    fixed: bool
    a_idx: np.ndarray
    b_idx: np.ndarray
    ws: np.ndarray
    k_arr: np.ndarray


@dataclass
class PackedJobs:
    """Job tables in the C-ABI's padded layout (scape_hip_batch_em)."""
    ju: np.ndarray          # int32 [n]      UTR index within the batch
    jk: np.ndarray          # int32 [n]      K
    jf: np.ndarray          # int32 [n]      fixed-inference flag
    a: np.ndarray           # int32 [n,kmax] alpha index
    b: np.ndarray           # int32 [n,kmax] beta index
    w: np.ndarray           # f64   [n,kmax+1]
    ka: np.ndarray          # int8  [n,nround]

    def __len__(self):
        return len(self.ju)

    @property
    def kmax(self):
        return self.a.shape[1]


def pack_jobs(jobs):
    n = len(jobs)
    kmax = max(1, max(j.K for j in jobs))
    a = np.zeros((n, kmax), dtype=np.int32)
    b = np.zeros((n, kmax), dtype=np.int32)
    w = np.zeros((n, kmax + 1), dtype=np.float64)
    ka = np.zeros((n, N_ROUND), dtype=np.int8)
    for i, j in enumerate(jobs):
        a[i, :j.K], b[i, :j.K], w[i, :j.K + 1], ka[i] = j.a_idx, j.b_idx, j.ws, j.k_arr
    return PackedJobs(i32([j.u for j in jobs]), i32([j.K for j in jobs]), i32([int(j.fixed) for j in jobs]), a, b, w, ka)


def concat_packed(packs):
    """Stack job tables of different K pitch into one (rows keep their order)."""
    if len(packs) == 1:
        return packs[0]
    n, kmax = sum(len(p) for p in packs), max(p.kmax for p in packs)
    a, b = np.zeros((n, kmax), np.int32), np.zeros((n, kmax), np.int32)
    w = np.zeros((n, kmax + 1), np.float64)
    lo = 0
    for p in packs:
        hi, k = lo + len(p), p.kmax
        a[lo:hi, :k], b[lo:hi, :k], w[lo:hi, :k + 1] = p.a, p.b, p.w
        lo = hi
    cat = lambda name: np.concatenate([getattr(p, name) for p in packs])   # noqa: E731
    return PackedJobs(cat("ju"), cat("jk"), cat("jf"), a, b, w, cat("ka"))


class HipBatch:
    """A batch of UTRs resident on the GPU (scape_hip_batch_* calls)."""

    def __init__(self, ctx, preps):
        self.ctx, self.lib, self.preps = ctx, ctx.lib, preps
        p0 = preps[0]
        for q in preps[1:]:
            if not (np.array_equal(q.betas, p0.betas) and np.array_equal(q.s_dis, p0.s_dis)
                    and q.p["mu_f"] == p0.p["mu_f"] and q.p["sigma_f"] == p0.p["sigma_f"]
                    and q.p["max_unif_ws"] == p0.p["max_unif_ws"]):
                raise ValueError("all UTRs of a batch must share the model constants")
        if len(p0.betas) > _lib.MAX_BETA or len(p0.s_dis) > _lib.MAX_S:
            raise ValueError("beta / polyA-length grid too long for the HIP kernels")
        prm = Params()
        prm.mu_f, prm.sigma_f, prm.max_unif_ws = float(p0.p["mu_f"]), float(p0.p["sigma_f"]), float(p0.p["max_unif_ws"])
        prm.n_beta, prm.n_s, prm.nround = len(p0.betas), len(p0.s_dis), N_ROUND
        for i, v in enumerate(p0.betas):
            prm.betas[i] = v
        for i, (s, q) in enumerate(zip(p0.s_dis, p0.pmf_s)):
            prm.s_dis[i], prm.pmf_s[i] = s, q
        self.params = prm
        self.bin_off = np.zeros(len(preps) + 1, dtype=np.int64)
        self.theta_off = np.zeros(len(preps) + 1, dtype=np.int64)
        np.cumsum([q.N for q in preps], out=self.bin_off[1:])
        np.cumsum([q.T for q in preps], out=self.theta_off[1:])
        cat = lambda name: f64(np.concatenate([getattr(q, name) for q in preps]))
        x, l, r, pa, cnt, th = cat("x"), cat("l"), cat("r"), cat("pa"), cat("cnt"), cat("theta")
        Ls = f64([q.L for q in preps])
        mt = f64([q.min_theta for q in preps])
        ul = f64([q.unif_ll for q in preps])
        check(self.lib.scape_hip_batch_load(self.ctx.h, ctypes.byref(prm), len(preps), ptr(self.bin_off, P_i64),
                                            ptr(x), ptr(l), ptr(r), ptr(pa), ptr(cnt), ptr(self.theta_off, P_i64),
                                            ptr(th), ptr(Ls), ptr(mt), ptr(ul)), "batch_load")
        self.n_bins = int(self.bin_off[-1])

    def build(self):
        check(self.lib.scape_hip_batch_build(self.ctx.h), "batch_build")

    def phase_b_form(self):
        """Which Phase B form the last build() queued (0 one kernel, 1 tables + per-alpha, 2 sliding windows)."""
        form = ctypes.c_int32(-1)
        check(self.lib.scape_hip_batch_phase_b_form(self.ctx.h, ctypes.byref(form)), "batch_phase_b_form")
        return int(form.value)

    def em_packed(self, pj, reuse_buffers=False, want_lb=True):
        """Run em_algo for packed job tables; returns (alpha_idx, beta_idx, ws, bic, n_lb, lb) arrays.
        reuse_buffers: hand out the same host arrays on every call of this shape (the caller must be
        done with the previous result) - saves re-allocating ~0.5 KB per job.
        want_lb=False: the lb_arr rows stay on the device (lb is None; 400 bytes per job, 20 MB per 512-UTR sweep) and
        the caller fetches the rows of the jobs it keeps with :meth:`em_fetch_lb`."""
        n, kmax = len(pj), pj.kmax
        key = (n, kmax, want_lb)
        if reuse_buffers and getattr(self, "_em_buf_key", None) == key:
            ao, bo, wo, bic, nlb, lb = self._em_buf
        else:
            ao, bo = np.empty_like(pj.a), np.empty_like(pj.b)
            wo = np.empty_like(pj.w)
            bic = np.empty(n)
            nlb = np.empty(n, dtype=np.int32)
            lb = np.empty((n, N_ROUND)) if want_lb else None   # rows are valid up to n_lb (the device buffer is zero-filled)
            if reuse_buffers:
                self._em_buf_key, self._em_buf = key, (ao, bo, wo, bic, nlb, lb)
        check(self.lib.scape_hip_batch_em(self.ctx.h, n, kmax, ptr(pj.ju, P_i32), ptr(pj.jk, P_i32),
                                          ptr(pj.jf, P_i32), ptr(pj.a, P_i32), ptr(pj.b, P_i32), ptr(pj.w),
                                          ptr(pj.ka, P_i8), ptr(ao, P_i32), ptr(bo, P_i32), ptr(wo), ptr(bic),
                                          ptr(nlb, P_i32), ptr(lb) if want_lb else None), "batch_em")
        return ao, bo, wo, bic, nlb, lb

    def em_fetch_lb(self, job_idx):
        """lb_arr rows [len(job_idx), nround] of jobs of the LAST em_packed call (scape_hip_batch_em_fetch_lb)."""
        idx = i32(job_idx)
        out = np.empty((len(idx), N_ROUND))
        if len(idx):
            check(self.lib.scape_hip_batch_em_fetch_lb(self.ctx.h, len(idx), ptr(idx, P_i32), ptr(out)), "em_fetch_lb")
        return out

    @staticmethod
    def fit_at(pj, out, i, lb_row=None):
        """lb_row: the job's lb_arr row when `out` was produced with want_lb=False (em_fetch_lb)."""
        ao, bo, wo, bic, nlb, lb = out
        K = int(pj.jk[i])
        row = lb[i] if lb_row is None else lb_row
        return Fit(K=K, a_idx=ao[i, :K].copy(), b_idx=bo[i, :K].copy(), ws=wo[i, :K + 1].copy(),
                   bic=float(bic[i]), lb=row[:nlb[i]].copy())

    def em(self, jobs):
        """Run em_algo for a list of _Job; returns a list of Fit in the same order."""
        if not jobs:
            return []
        pj = pack_jobs(jobs)
        out = self.em_packed(pj)
        return [self.fit_at(pj, out, i) for i in range(len(jobs))]

    def timing(self, which):
        """(sum of HIP-event durations in ms, launches) of kernel kind `which` since the last reset."""
        ms, n = ctypes.c_double(), ctypes.c_int32()
        check(self.lib.scape_hip_timing_get(self.ctx.h, which, ctypes.byref(ms), ctypes.byref(n)), "timing_get")
        return ms.value, n.value

    def em_counters(self):
        """(rounds, tensor elements read by the M-step, Z elements) of the last em call."""
        r, s, z = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(self.lib.scape_hip_em_counters(self.ctx.h, ctypes.byref(r), ctypes.byref(s), ctypes.byref(z)),
              "em_counters")
        return r.value, s.value, z.value

    def em_traffic(self):
        """The M-step kernel's own byte tally over the last em call: dict(tensor_bytes, v_bytes_requested,
        v_bytes_unique, launches) (scape_hip_em_traffic)."""
        t, vr, vu, n = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(self.lib.scape_hip_em_traffic(self.ctx.h, ctypes.byref(t), ctypes.byref(vr), ctypes.byref(vu),
                                            ctypes.byref(n)), "em_traffic")
        return dict(tensor_bytes=t.value, v_bytes_requested=vr.value, v_bytes_unique=vu.value, launches=n.value)

    def labels(self, sel):
        """sel: list of (u, Fit) -> dict u -> int32 labels per bin (get_label, :873-881)."""
        if not sel:
            return {}
        n = len(sel)
        kmax = max(1, max(f.K for _, f in sel))
        su = i32([u for u, _ in sel])
        sk = i32([f.K for _, f in sel])
        a = np.zeros((n, kmax), dtype=np.int32)
        b = np.zeros((n, kmax), dtype=np.int32)
        w = np.zeros((n, kmax + 1), dtype=np.float64)
        for i, (_, f) in enumerate(sel):
            a[i, :f.K], b[i, :f.K], w[i, :f.K + 1] = f.a_idx, f.b_idx, f.ws
        out = np.full(self.n_bins, -1, dtype=np.int32)
        check(self.lib.scape_hip_batch_labels(self.ctx.h, n, kmax, ptr(su, P_i32), ptr(sk, P_i32), ptr(a, P_i32),
                                              ptr(b, P_i32), ptr(w), ptr(out, P_i32)), "batch_labels")
        return {u: out[self.bin_off[u]:self.bin_off[u + 1]].copy() for u, _ in sel}

    def labels_packed(self, su, sk, a, b, w):
        """get_label for models given as padded tables; returns the flat int32 label array of the batch."""
        su, sk, a, b = i32(su), i32(sk), i32(a), i32(b)
        w = f64(w)
        out = np.full(self.n_bins, -1, dtype=np.int32)
        check(self.lib.scape_hip_batch_labels(self.ctx.h, len(su), a.shape[1], ptr(su, P_i32), ptr(sk, P_i32),
                                              ptr(a, P_i32), ptr(b, P_i32), ptr(w), ptr(out, P_i32)), "batch_labels")
        return out

    def fetch_loglik(self, u):
        q = self.preps[u]
        A = np.zeros((q.N, q.T))
        check(self.lib.scape_hip_batch_fetch_loglik(self.ctx.h, u, ptr(A)), "fetch_loglik")
        return A

    def fetch_tensor(self, u):
        q = self.preps[u]
        M = np.zeros((q.T, len(q.betas), q.N))
        check(self.lib.scape_hip_batch_fetch_tensor(self.ctx.h, u, ptr(M)), "fetch_tensor")
        return M

    def free(self):
        check(self.lib.scape_hip_batch_free(self.ctx.h), "batch_free")


def _no_component_left(q):
    """rm_component with every pA component below min_ws: the reference indexes alpha_arr with an empty float
    array (apa_core.py:838-839) and dies with this IndexError; there is no result to reproduce."""
    raise IndexError(f"{q.gene_info_str}: every pA component has weight < min_ws={q.p['min_ws']} "
                     "(the reference's rm_component fails here too: arrays used as indices must be of integer "
                     "(or boolean) type, apa_core.py:838-839)")


class _Sweep:
    """Per-UTR state machine of run() / subsample_run()'s re-run loop."""

    def __init__(self, u, prep, sampler, re_run_mode, trace=None):
        self.u, self.prep, self.sampler, self.re_run = u, prep, sampler, re_run_mode
        self.n_max, self.n_min = prep.p["n_max_apa"], prep.p["n_min_apa"]
        self.stage = "sweep"
        self.best = None
        self.done = False
        self.n_jobs = 0
        self.trace = trace
        self.pred = None        # Engine._predict_outcomes: None / "clean" / ("prune", K') / "stop" - a guess, never a result

    def prune_K(self):
        """K' of rm_component (:836-839): components of the current best fit that stay."""
        f = self.best
        return int(np.count_nonzero(~(f.ws[:f.K] < self.prep.p["min_ws"])))

    def make_jobs(self, predrawn=None):
        """predrawn = (init_ws table, k_arr) of the prune re-fit when they were drawn ahead (Engine._drive_streams)."""
        q = self.prep
        jobs = []
        if self.stage == "sweep":                       # run(): K = n_max .. n_min (:965)
            for K in range(self.n_max, self.n_min - 1, -1):
                for _ in range(N_TRIAL):
                    a, b, w, ka = self.sampler.init_job(q, K)
                    jobs.append(_Job(self.u, K, False, a, b, w, ka))
        elif self.stage == "prune":                     # rm_component -> fixed_inference (:843, :708-711)
            f = self.best
            keep = np.array([i for i in range(f.K) if not f.ws[i] < q.p["min_ws"]], dtype=int)
            Kp = len(keep)
            if Kp == 0:
                _no_component_left(q)
            if predrawn is not None:
                w, ka = predrawn
            else:
                w = self.sampler.init_ws(Kp, q.p["max_unif_ws"])
                ka = self.sampler.k_arr(Kp)
            jobs.append(_Job(self.u, Kp, True, f.a_idx[keep].astype(np.int32), f.b_idx[keep].astype(np.int32), w, ka))
        self.n_jobs += len(jobs)
        return jobs

    def make_packed(self, tables=None, predrawn=None):
        """make_jobs as padded tables (one native call per sweep, engine._drive's fast path); `tables` = the
        sweep tables when they were drawn together with other streams' (make_packed_many)."""
        q = self.prep
        if predrawn is not None:
            return pack_jobs(self.make_jobs(predrawn))
        if self.stage == "sweep" and hasattr(self.sampler, "sweep"):
            jk, a, b, w, ka = tables if tables is not None else self.sampler.sweep(q, self.n_max, self.n_min)
            n = len(jk)
            self.n_jobs += n
            return PackedJobs(np.full(n, self.u, np.int32), jk, np.zeros(n, np.int32), a, b, w, ka)
        return pack_jobs(self.make_jobs())

    @staticmethod
    def make_packed_many(sweeps):
        """make_packed for the heads of several streams: the sweep tables of streams with distinct native samplers
        are drawn in one threaded call (FastSampler.sweep_many)."""
        idx = [i for i, sw in enumerate(sweeps) if sw.stage == "sweep" and isinstance(sw.sampler, FastSampler)]
        if len(idx) < 2 or len({id(sweeps[i].sampler) for i in idx}) != len(idx):
            return [sw.make_packed() for sw in sweeps]
        tabs = FastSampler.sweep_many([(sweeps[i].sampler, sweeps[i].prep, sweeps[i].n_max, sweeps[i].n_min) for i in idx])
        by = dict(zip(idx, tabs))
        return [sw.make_packed(by.get(i)) for i, sw in enumerate(sweeps)]

    def absorb_packed(self, pj, out, lo, hi):
        """absorb for rows [lo, hi) of a packed EM result: only the winner becomes a Fit."""
        if self.stage == "sweep":
            with np.errstate(over="ignore"):
                grid = out[3][lo:hi].astype(np.float32).reshape(-1, N_TRIAL)     # float32 bic_arr (:849, :945)
            tb = np.argmin(grid, axis=1)                                         # em_optim0 (:865)
            kb = int(np.argmin(grid[np.arange(len(tb)), tb]))                    # run (:972)
            self.best = HipBatch.fit_at(pj, out, lo + kb * N_TRIAL + int(tb[kb]))
            if not self.prep.fixed_run and np.any(self.best.ws[:self.best.K] < self.prep.p["min_ws"]):
                self.stage = "prune"
                return
        elif self.stage == "prune":
            self.best = HipBatch.fit_at(pj, out, lo)
        self._after_fit()

    def absorb(self, fits):
        q = self.prep
        if self.trace is not None:
            self.trace.extend(fits)
        if self.stage == "sweep":
            per_k = []
            for i in range(0, len(fits), N_TRIAL):      # em_optim0: first arg-min BIC over trials (:865)
                grp = fits[i:i + N_TRIAL]
                # bic_arr is np.full(n, np.finfo('f').max): a float32 array (:849, :945), so both
                # arg-mins compare BICs rounded to f32 (ties -> first)
                per_k.append(grp[int(np.argmin(np.array([f.bic for f in grp], dtype=np.float32)))])
            self.best = per_k[int(np.argmin(np.array([f.bic for f in per_k], dtype=np.float32)))]     # (:972)
            if not q.fixed_run and any(self.best.ws[i] < q.p["min_ws"] for i in range(self.best.K)):
                self.stage = "prune"
                return
        elif self.stage == "prune":
            self.best = fits[0]
        self._after_fit()

    def set_sweep_winner(self, best, n_jobs):
        """Entry for the vectorised selection: the sweep's BIC winner is already known."""
        self.best = best
        self.n_jobs += n_jobs
        if not self.prep.fixed_run and any(best.ws[i] < self.prep.p["min_ws"] for i in range(best.K)):
            self.stage = "prune"
            return
        self._after_fit()

    def _after_fit(self):
        # re-run rule of subsample_run (:1023-1030); fixed_run returns before it (:1009-1017)
        if self.re_run and not self.prep.fixed_run and self.best.K == self.n_max:
            if self.n_max + 2 > _lib.MAX_K:
                # the reference grows n_max_apa without bound (:1023-1030); the kernels hold K <= MAX_K components.
                # Only this UTR stops re-running - the other UTRs of the wave and file are unaffected.
                import sys
                sys.stderr.write(f"scape_amd: {self.prep.gene_info_str}: K reached n_max_apa={self.n_max} and the "
                                 f"next sweep would exceed the library's K <= {_lib.MAX_K}; keeping K={self.best.K} "
                                 "(the reference would re-run with n_max_apa + 2)\n")
                self.done = True
                return
            self.n_min = self.n_max
            self.n_max = self.n_max + 2
            self.stage = "sweep"
            return
        self.done = True


class Engine:
    """infer_pa for lists of UTRs on one GPU."""
    defer_prunes = True      # reference-stream modes: run the ws-only re-fits of a wave in one launch at its end

    def __init__(self, device=None, mem_fraction=0.7, own_context=False):
        """own_context: create a private library handle (own HIP stream and device buffers) instead of the
        process-wide one per device - required for engines that run on different host threads."""
        if own_context:
            if device is None:
                device = _lib.default_context(None).device
            self.ctx = _lib.Context(device)
        else:
            self.ctx = _lib.default_context(device)
        self.mem_fraction = mem_fraction
        self.own_context = own_context
        self.sweep_lock = None       # optional threading.Lock shared by engines on one device (pipeline.py)

    def _budget(self):
        fb, tb, bb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(self.ctx.lib.scape_hip_batch_bytes(self.ctx.h, ctypes.byref(bb), ctypes.byref(fb), ctypes.byref(tb)),
              "batch_bytes")
        return int(fb.value * self.mem_fraction)

    @staticmethod
    def utr_bytes(q):
        npad = (q.N + 15) // 16 * 16
        return 8 * npad * q.T * (len(q.betas) + 2) + 48 * q.N

    def close(self):
        """Release a private library handle (no-op for the shared per-device handle)."""
        if self.own_context:
            self.ctx.close()

    def load(self, preps):
        """Upload one wave of prepared UTRs (scape_hip_batch_load)."""
        return HipBatch(self.ctx, preps)

    def waves(self, preps):
        """Split into waves whose tensors fit the device (at most 65535 UTRs each)."""
        budget = self._budget()
        out, cur, used = [], [], 0
        for i, q in enumerate(preps):
            b = self.utr_bytes(q)
            if b > budget:
                raise _lib.ScapeHipError(f"UTR {q.gene_info_str}: marginal tensor ({b >> 20} MiB) exceeds device memory")
            if cur and (used + b > budget or len(cur) >= 65535):
                out.append(cur)
                cur, used = [], 0
            cur.append(i)
            used += b
        if cur:
            out.append(cur)
        return out

    def run(self, preps, rng_mode="reference", seed=1, re_run_mode=True, rs=None, keep_trace=False, seeds=None):
        """Returns a list of UtrResult (same order as preps).  per_utr mode: UTR i draws from
        RandomState(seeds[i]) (default seed + i)."""
        if rng_mode not in ("reference", "per_utr"):
            raise ValueError("rng_mode must be 'reference' or 'per_utr'")
        results = [UtrResult(prep=q) for q in preps]
        if seeds is None:
            seeds = [(seed + i) % (2 ** 32) for i in range(len(preps))]
        shared = FastSampler(rs if rs is not None else np.random.RandomState(seed)) if rng_mode == "reference" else None
        self.traces = [[] for _ in preps] if keep_trace else None
        for wave in self.waves(preps):
            wp = [preps[i] for i in wave]
            batch = HipBatch(self.ctx, wp)
            if rng_mode == "reference" or keep_trace or any(q.fixed_run for q in wp):
                batch.build()
                sweeps = []
                for u, gi in enumerate(wave):
                    smp = shared if shared is not None else FastSampler(np.random.RandomState(seeds[gi]))
                    sweeps.append(_Sweep(u, preps[gi], smp, re_run_mode, self.traces[gi] if keep_trace else None))
                if rng_mode == "reference":
                    deferred = []
                    if keep_trace:
                        for sw in sweeps:
                            self._drive(batch, [sw], deferred)
                    else:
                        depth = self._stream_depth(1)
                        if depth > 1:
                            self._predict_outcomes(batch, wp, sweeps, re_run_mode)
                        self._drive_streams(batch, [(shared, sweeps)], deferred, depth=depth)
                    self._finish_deferred(batch, deferred)
                else:
                    self._drive(batch, sweeps)
                labs = batch.labels([(sw.u, sw.best) for sw in sweeps])
                out = [(sw.best, labs[sw.u], sw.n_jobs) for sw in sweeps]
            else:
                plan = self.plan(wp, [seeds[gi] for gi in wave])
                out = self.process(batch, wp, plan, re_run_mode)
            for (fit, lab, nj), gi in zip(out, wave):
                results[gi].fit, results[gi].labels_bin, results[gi].n_jobs = fit, lab, nj
        if shared is not None:
            shared.rs                      # hand the advanced stream back to the caller's RandomState
        return results

    # ---- several reference-RNG streams at once (one stream = one chunk file) --------------------------
    def run_streams(self, streams, re_run_mode=True):
        """streams: list of (preps, seed).  Every stream keeps the reference's serial RandomState(seed)
        semantics (its UTRs are processed one after another, draws in the reference's order), but the
        current UTRs of ALL streams share each EM launch - chunk files are independent random streams
        (np.random.seed(1) per file, apa_core.py:125), so results equal running the files one by one."""
        out = [None] * len(streams)
        ready = [(si, preps, seed) for si, (preps, seed) in enumerate(streams)]

        def take(block, room):
            got = ready[:room]
            del ready[:room]
            return got
        self.run_streams_rolling(take, lambda: bool(ready), lambda si, res: out.__setitem__(si, res), re_run_mode,
                                 streams_in_flight=max(1, len(streams)))
        return out

    stream_wave_utrs = 16    # UTRs one stream brings into a wave at most: early waves stay short, so files that become ready join soon
    stream_wave_min = 512    # ... unless the wave would hold fewer UTRs than this in all (few streams: longer pieces of each)

    def run_streams_rolling(self, take, more, on_done, re_run_mode=True, streams_in_flight=128):
        """run_streams over streams that become available while it runs (chunk files whose prep finishes).
        take(block, room) -> [(tag, preps, seed)] newly available streams, at most `room`; with block=True it waits for
        at least one unless none is left.  more() -> False once take() can return nothing any more.
        on_done(tag, [UtrResult]) is called when a stream's last UTR is finished (not necessarily in admission order).
        Streams join at wave boundaries; a wave takes the next UTRs of every stream in flight, round-robin, up to
        `stream_wave_utrs` per stream and `stream_wave_bytes` of tensors in all."""
        full = self._budget()
        budget = min(full, Engine.stream_wave_bytes)
        active = []       # [tag, preps, sampler, cursor, results]
        while active or more():
            room = streams_in_flight - len(active)
            if room > 0 and more():
                for tag, preps, seed in take(not active, room):
                    st = [tag, preps, FastSampler(np.random.RandomState(seed)), 0, [UtrResult(prep=q) for q in preps]]
                    if len(preps):
                        active.append(st)
                    else:
                        on_done(tag, [])
            if not active:
                continue
            wave, used, stop = [], 0, False           # (stream, index within the stream), stream-major order below
            per_stream = max(Engine.stream_wave_utrs, -(-Engine.stream_wave_min // len(active)))
            for turn in range(per_stream):
                took = False
                for st in active:
                    i = st[3]
                    if i >= len(st[1]) or len(wave) >= 65535:
                        continue
                    bts = self.utr_bytes(st[1][i])
                    if bts > full:
                        raise _lib.ScapeHipError("a UTR's marginal tensor exceeds device memory")
                    if wave and used + bts > budget:
                        stop = True
                        break
                    wave.append((st, i))
                    st[3] = i + 1
                    used += bts
                    took = True
                if stop or not took:
                    break
            wave.sort(key=lambda e: (id(e[0]), e[1]))     # each stream's UTRs contiguous and in file order
            preps = [st[1][i] for st, i in wave]
            batch = HipBatch(self.ctx, preps)
            batch.build()
            queues, wave_sweeps = {}, []
            for u, (st, _i) in enumerate(wave):
                wave_sweeps.append(_Sweep(u, preps[u], st[2], re_run_mode))
                queues.setdefault(id(st), (st, []))[1].append(wave_sweeps[-1])
            done_sweeps, deferred = [], []
            depth = self._stream_depth(len(queues))
            if depth > 1:
                self._predict_outcomes(batch, preps, wave_sweeps, re_run_mode)
            self._drive_streams(batch, [(st[2], q) for st, q in queues.values()], deferred, done_sweeps.append, depth=depth)
            self._finish_deferred(batch, deferred)
            labs = batch.labels([(sw.u, sw.best) for sw in done_sweeps])
            for sw in done_sweeps:
                st, i = wave[sw.u]
                r = st[4][i]
                r.fit, r.labels_bin, r.n_jobs = sw.best, labs[sw.u], sw.n_jobs
            still = []
            for st in active:
                if st[3] < len(st[1]):
                    still.append(st)
                else:
                    on_done(st[0], st[4])
            active = still

    # ---- fully batched path (per-UTR RNG): pre-drawn job tables + vectorised selection -------------
    @staticmethod
    def plan_python(preps, seeds):
        """The plan drawn with numpy's RandomState itself (scape_amd/host.py::Sampler): readable twin of
        :meth:`plan`, used for the UTRs the native sampler declines and by tests/test_host.py."""
        jobs, samplers = [], []
        for u, (q, sd) in enumerate(zip(preps, seeds)):
            smp = Sampler(np.random.RandomState(sd))
            for K in range(q.p["n_max_apa"], q.p["n_min_apa"] - 1, -1):
                for _ in range(N_TRIAL):
                    a, b, w, ka = smp.init_job(q, K)
                    jobs.append(_Job(u, K, False, a, b, w, ka))
            samplers.append(smp)
        spans = np.zeros(len(preps) + 1, dtype=np.int64)
        np.cumsum([(q.p["n_max_apa"] - q.p["n_min_apa"] + 1) * N_TRIAL for q in preps], out=spans[1:])
        states = np.stack([_hostlib.state_from_numpy(s.rs.get_state()) for s in samplers])
        # rm_component draws (init_ws + gen_k_arr, :843 -> :709 -> :720) for every possible K' < n_max,
        # each continuing the UTR's stream from the state the sweep left behind
        kcap = max(q.p["n_max_apa"] for q in preps)
        prune_w = np.zeros((len(preps), kcap, kcap + 1))
        prune_ka = np.zeros((len(preps), kcap, N_ROUND), dtype=np.int8)
        prune_states = np.zeros((len(preps), kcap, _hostlib.STATE_WORDS), dtype=np.uint32)
        rs = np.random.RandomState(0)
        for u, q in enumerate(preps):
            for Kp in range(q.p["n_max_apa"]):
                rs.set_state(_hostlib.state_to_numpy(states[u]))
                smp = Sampler(rs)
                prune_w[u, Kp, :Kp + 1] = smp.init_ws(Kp, q.p["max_unif_ws"])
                prune_ka[u, Kp] = smp.k_arr(Kp)
                prune_states[u, Kp] = _hostlib.state_from_numpy(rs.get_state())
        return dict(main=pack_jobs(jobs), spans=spans, states=states, prune_w=prune_w, prune_ka=prune_ka,
                    prune_states=prune_states)

    @staticmethod
    def plan(preps, seeds, n_threads=None):
        """Draw every restart of the first K sweep (init_para + gen_k_arr, reference order) for each
        UTR from its own RandomState(seed), plus the tables rm_component would draw for every possible
        K' and the generator states left behind (prune / re-run draws continue from them).  Drawn by
        libscape_host.so (include/scape_host.h::scape_host_plan) on the host threads; numbers are
        those of :meth:`plan_python`."""
        H = _hostlib
        lib = H.load_library()
        U = len(preps)
        n_max = i32([q.p["n_max_apa"] for q in preps])
        n_min = i32([q.p["n_min_apa"] for q in preps])
        spans = np.zeros(U + 1, dtype=np.int64)
        np.cumsum((n_max - n_min + 1).astype(np.int64) * N_TRIAL, out=spans[1:])
        J, kmax = int(spans[-1]), max(1, int(n_max.max()))
        kcap = kmax
        peak_off = np.zeros(U + 1, dtype=np.int64)
        np.cumsum([len(q.peaks) for q in preps], out=peak_off[1:])
        theta_off = np.zeros(U + 1, dtype=np.int64)
        np.cumsum([len(q.theta) for q in preps], out=theta_off[1:])
        f64 = lambda xs: np.ascontiguousarray(np.concatenate(xs) if len(xs) else np.zeros(0), dtype=np.float64)  # noqa: E731
        peaks, peak_w = f64([q.peaks for q in preps]), f64([q.peak_w for q in preps])
        theta = f64([q.theta for q in preps])
        Ls, n_beta = i32([q.L for q in preps]), i32([len(q.betas) for q in preps])
        scale = np.array([5 * q.p["beta_step"] for q in preps], dtype=np.float64)
        mxu = np.array([q.p["max_unif_ws"] for q in preps], dtype=np.float64)
        sd = np.array([int(x) % (2 ** 32) for x in seeds], dtype=np.uint32)
        pj = PackedJobs(np.zeros(J, np.int32), np.zeros(J, np.int32), np.zeros(J, np.int32),
                        np.zeros((J, kmax), np.int32), np.zeros((J, kmax), np.int32),
                        np.zeros((J, kmax + 1), np.float64), np.zeros((J, N_ROUND), np.int8))
        states = np.zeros((U, H.STATE_WORDS), dtype=np.uint32)
        prune_w = np.zeros((U, kcap, kcap + 1))
        prune_ka = np.zeros((U, kcap, N_ROUND), dtype=np.int8)
        prune_states = np.zeros((U, kcap, H.STATE_WORDS), dtype=np.uint32)
        status = np.ones(U, dtype=np.int32)
        ok_k = kmax <= H.MAX_K
        if ok_k and U:
            A = H.PlanArgs(U, n_threads or H.host_threads(), N_TRIAL, N_ROUND, kmax, kcap,
                           H.ptr(sd, H.P_u32), H.ptr(peak_off, H.P_i64), H.ptr(peaks, H.P_d), H.ptr(peak_w, H.P_d),
                           H.ptr(theta_off, H.P_i64), H.ptr(theta, H.P_d),
                           H.ptr(Ls, H.P_i32), H.ptr(n_max, H.P_i32), H.ptr(n_min, H.P_i32), H.ptr(n_beta, H.P_i32),
                           H.ptr(scale, H.P_d), H.ptr(mxu, H.P_d), H.ptr(spans, H.P_i64),
                           H.ptr(pj.ju, H.P_i32), H.ptr(pj.jk, H.P_i32), H.ptr(pj.a, H.P_i32), H.ptr(pj.b, H.P_i32),
                           H.ptr(pj.w, H.P_d), H.ptr(pj.ka, H.P_i8), H.ptr(states, H.P_u32), H.ptr(prune_w, H.P_d),
                           H.ptr(prune_ka, H.P_i8), H.ptr(prune_states, H.P_u32), H.ptr(status, H.P_i32))
            if lib.scape_host_plan(ctypes.byref(A)):
                status[:] = 1
        for u in np.nonzero(status)[0]:            # declined by the native path: numpy draws them
            u = int(u)
            one = Engine.plan_python([preps[u]], [int(sd[u])])
            lo, hi, m, k1 = int(spans[u]), int(spans[u + 1]), one["main"], one["main"].kmax
            pj.ju[lo:hi], pj.jk[lo:hi] = u, m.jk
            pj.a[lo:hi], pj.b[lo:hi], pj.w[lo:hi] = 0, 0, 0.0
            pj.a[lo:hi, :k1], pj.b[lo:hi, :k1], pj.w[lo:hi, :k1 + 1], pj.ka[lo:hi] = m.a, m.b, m.w, m.ka
            states[u] = one["states"][0]
            prune_w[u], prune_ka[u], prune_states[u] = 0.0, 0, 0
            prune_w[u, :k1, :k1 + 1], prune_ka[u, :k1] = one["prune_w"][0], one["prune_ka"][0]
            prune_states[u, :k1] = one["prune_states"][0]
        return dict(main=pj, spans=spans, states=states, prune_w=prune_w, prune_ka=prune_ka,
                    prune_states=prune_states)

    def process(self, batch, preps, plan, re_run_mode=True, build=True, labels=True):
        """One pass of the hot path over a resident batch: Phase A/B, the main EM sweep, BIC model
        selection, prune re-fits, re-run sweeps and labels.  Returns [(Fit, labels_bin, n_jobs)]."""
        from time import perf_counter as _now
        t0 = _now()
        pj, spans = plan["main"], plan["spans"]
        # engines that share a device take turns for the long kernels (two EM sweeps at once only slow each
        # other down); uploads, host-side selection and the short re-fit launches overlap freely
        ms0 = batch.timing(2)[0]          # (synchronises: read before the build is queued, not between build and EM)
        if build:
            batch.build()        # queued, not awaited: the EM call's host-side table checks run while the GPU builds
        with (self.sweep_lock if self.sweep_lock is not None else contextlib.nullcontext()):
            out = batch.em_packed(pj, reuse_buffers=True, want_lb=False)   # lb rows of the BIC winners only, below
            self.last_main_em_ms = batch.timing(2)[0] - ms0          # HIP-event time of the sweep launch
        self.last_main_counters = batch.em_counters()
        t1 = _now()
        ao, bo, wo, bic, nlb, lb = out
        U = len(preps)
        # ---- em_optim0 (:865) + run (:972): first arg-min BIC over restarts, then over K (desc) ----
        njobs = np.diff(spans)
        with np.errstate(over="ignore"):
            bic32 = bic.astype(np.float32)      # the reference's bic_arr is float32 (:849, :945)
        if np.all(njobs == njobs[0]):
            nk = int(njobs[0]) // N_TRIAL
            grid = bic32.reshape(U, nk, N_TRIAL)
            tb = np.argmin(grid, axis=2)
            kb = np.argmin(np.take_along_axis(grid, tb[:, :, None], axis=2)[:, :, 0], axis=1)
            win = spans[:-1] + kb * N_TRIAL + tb[np.arange(U), kb]
        else:
            win = np.zeros(U, dtype=np.int64)
            for u in range(U):
                lo, hi = int(spans[u]), int(spans[u + 1])
                g = bic32[lo:hi].reshape(-1, N_TRIAL)
                tbu = np.argmin(g, axis=1)
                kbu = int(np.argmin(g[np.arange(len(tbu)), tbu]))
                win[u] = lo + kbu * N_TRIAL + int(tbu[kbu])
        lb_win = batch.em_fetch_lb(win)             # Parameters.lb_arr of every UTR's winner (:972), before any other EM call
        Kw = pj.jk[win].astype(np.int64)
        min_ws = np.array([q.p["min_ws"] for q in preps])
        n_max = np.array([q.p["n_max_apa"] for q in preps])
        comp = np.arange(pj.kmax)[None, :] < Kw[:, None]
        needs_prune = np.any((wo[win, :pj.kmax] < min_ws[:, None]) & comp, axis=1)
        fits = [None] * U
        njob_out = njobs.astype(np.int64).copy()
        # ---- rm_component (:832-844): ws-only re-fit of the kept components, all pruned UTRs in one launch
        pu = np.nonzero(needs_prune)[0]
        refit = {}
        if len(pu):
            keep = comp[pu] & ~(wo[win[pu], :pj.kmax] < min_ws[pu, None])
            Kp = keep.sum(axis=1)
            if np.any(Kp == 0):
                _no_component_left(preps[int(pu[int(np.argmin(Kp))])])
            order = np.argsort(~keep, axis=1, kind="stable")              # kept components first, in order
            kmp = max(1, int(Kp.max()))
            sel = np.arange(kmp)[None, :] < Kp[:, None]
            ra = np.where(sel, np.take_along_axis(ao[win[pu]], order, axis=1)[:, :kmp], 0).astype(np.int32)
            rb = np.where(sel, np.take_along_axis(bo[win[pu]], order, axis=1)[:, :kmp], 0).astype(np.int32)
            rw = np.ascontiguousarray(plan["prune_w"][pu, Kp, :kmp + 1])
            rka = np.ascontiguousarray(plan["prune_ka"][pu, Kp])
            rpj = PackedJobs(i32(pu), i32(Kp), np.ones(len(pu), dtype=np.int32), ra, rb, rw, rka)
            rout = batch.em_packed(rpj)
            njob_out[pu] += 1
            for i, u in enumerate(pu):
                refit[int(u)] = (rpj, rout, i)
        # ---- re-run rule (:1023-1030): K hit the cap -> continue that UTR with the generic state machine
        sweeps = []
        if re_run_mode:
            Kfin = Kw.copy()
            if len(pu):
                Kfin[pu] = Kp
            for u in np.nonzero(Kfin == n_max)[0]:
                u = int(u)
                rs = np.random.RandomState(0)
                if u in refit:
                    rpj, rout, i = refit.pop(u)
                    best = batch.fit_at(rpj, rout, i)
                    rs.set_state(_hostlib.state_to_numpy(plan["prune_states"][u][best.K]))
                else:
                    best = batch.fit_at(pj, out, int(win[u]), lb_win[u])
                    rs.set_state(_hostlib.state_to_numpy(plan["states"][u]))
                sw = _Sweep(u, preps[u], FastSampler(rs), re_run_mode)
                sw.best, sw.n_jobs = best, int(njob_out[u])
                sw._after_fit()
                sweeps.append(sw)
            self._drive(batch, [sw for sw in sweeps if not sw.done])
            for sw in sweeps:
                fits[sw.u], njob_out[sw.u] = sw.best, sw.n_jobs
        for u, (rpj, rout, i) in refit.items():
            fits[u] = batch.fit_at(rpj, rout, i)
        t2 = _now()
        if not labels:
            return [(fits[u] if fits[u] is not None else batch.fit_at(pj, out, int(win[u]), lb_win[u]), None, int(njob_out[u]))
                    for u in range(U)]
        # ---- labels for every UTR (get_label, :873-881) -------------------------------------------
        kmax_f = max(1, int(Kw.max()), max((f.K for f in fits if f is not None), default=1))
        la = np.zeros((U, kmax_f), dtype=np.int32)
        lbi = np.zeros((U, kmax_f), dtype=np.int32)
        lw = np.zeros((U, kmax_f + 1), dtype=np.float64)
        lk = Kw.astype(np.int32).copy()
        m = min(kmax_f, pj.kmax)
        la[:, :m], lbi[:, :m] = ao[win, :m], bo[win, :m]
        lw[:, :m + 1] = wo[win, :m + 1]
        for u, f in enumerate(fits):
            if f is not None:
                lk[u] = f.K
                la[u], lbi[u], lw[u] = 0, 0, 0.0
                la[u, :f.K], lbi[u, :f.K], lw[u, :f.K + 1] = f.a_idx, f.b_idx, f.ws
        la[la < 0] = 0
        lbi[lbi < 0] = 0
        flat = batch.labels_packed(np.arange(U, dtype=np.int32), lk, la, lbi, lw)
        t3 = _now()
        res = []
        for u in range(U):
            f = fits[u] if fits[u] is not None else batch.fit_at(pj, out, int(win[u]), lb_win[u])
            res.append((f, flat[batch.bin_off[u]:batch.bin_off[u + 1]], int(njob_out[u])))
        t4 = _now()
        self.last_host_ms = dict(build_em=(t1 - t0) * 1e3, select_prune=(t2 - t1) * 1e3, labels=(t3 - t2) * 1e3,
                                 collect=(t4 - t3) * 1e3)
        return res

    @staticmethod
    def _defer_prunes(sweeps, deferred):
        """A sweep that ends in a prune is finished as far as the random stream is concerned: the re-fit's
        tables are drawn now (stream order, rm_component :843 -> :709 -> :720), its outcome feeds nothing but
        the UTR's own result (K' < n_max, so the re-run rule cannot fire) - so the EM call itself can wait and
        share one launch with the other re-fits of the wave.  Returns the sweeps still in progress."""
        rest = []
        for sw in sweeps:
            if sw.done:
                continue
            if sw.stage == "prune" and sw.trace is None and Engine.defer_prunes:
                deferred.append((sw, sw.make_packed()))
            else:
                rest.append(sw)
        return rest

    @staticmethod
    def _finish_deferred(batch, deferred):
        if not deferred:
            return
        pj = concat_packed([pk for _sw, pk in deferred])
        out = batch.em_packed(pj)
        lo = 0
        for sw, pk in deferred:
            sw.absorb_packed(pj, out, lo, lo + len(pk))
            lo += len(pk)
        deferred.clear()

    # ---- the exact-stream modes: UTRs of one random stream, several per EM call --------------------------
    stream_wave_bytes = 32 << 30     # run_streams: tensor bytes of one wave (~850 UTRs of the headline shape)
    spec_depth = 8        # UTRs of ONE stream that share an EM call (the head + its followers); 1 = strictly serial
    spec_max_streams = 8  # more streams in flight than this fill the launches with their heads alone (measured: followers
                          # gain 1.5x at 2 streams, 1.2x at 8, lose 10-25 % at 16-32 - the prediction pass is a second fit)
    spec_seed = 0x5CA9E     # seeds of the outcome-prediction pass (any value: predictions steer speed, never results)
    spec_stats = dict(calls=0, utrs_kept=0, utrs_discarded=0, predicted=0, predicted_right=0, calls_drawn_ahead=0)

    def _stream_depth(self, n_streams):
        return max(1, Engine.spec_depth) if n_streams <= Engine.spec_max_streams else 1

    def _predict_outcomes(self, batch, preps, sweeps, re_run_mode):
        """Guess how each UTR of the wave will leave its random stream - clean, pruned to K' components, or re-run -
        by fitting the whole wave once in the batched per-UTR-seed mode (one resident-batch pass, ~0.4 ms per UTR at
        the headline shape).  The guess only decides which tables the followers of a UTR are drawn from ahead of time
        (_drive_streams); a wrong guess costs a discarded follower, never a different result."""
        if len(preps) < 2 or any(q.fixed_run for q in preps):
            return                                    # fixed_run ends clean by construction (:1009-1017)
        plan = self.plan(preps, [(Engine.spec_seed + 7919 * i) % (2 ** 32) for i in range(len(preps))])
        # The pass is a guess from OTHER seeds: whatever goes wrong in it (every component of some UTR pruned away ->
        # the reference's IndexError; a library error from the extra wave-wide job tables) must not end a stream whose
        # real fits would finish, and it must not overwrite the measurement fields of the real sweeps.
        keep = {k: getattr(self, k, None) for k in ("last_main_em_ms", "last_main_counters", "last_host_ms")}
        try:
            out = self.process(batch, preps, plan, re_run_mode=False, build=False, labels=False)
        except (IndexError, _lib.ScapeHipError):
            for sw in sweeps:
                sw.pred = None
            return
        finally:
            for k, v in keep.items():
                if v is None:
                    self.__dict__.pop(k, None)
                else:
                    setattr(self, k, v)
        for sw, (fit, _lab, nj) in zip(sweeps, out):
            n_sweep = (sw.n_max - sw.n_min + 1) * N_TRIAL
            if re_run_mode and fit.K == sw.n_max:
                sw.pred = "stop"                      # a re-run is likely: nothing is drawn ahead past this UTR
            elif nj > n_sweep:
                sw.pred = ("prune", fit.K)
            else:
                sw.pred = "clean"

    @staticmethod
    def _drive_streams(batch, streams, deferred, on_done=None, depth=1):
        """streams: [(FastSampler, sweeps of that stream in file order)].  Runs every stream to its end, up to `depth`
        UTRs of a stream per EM call.

        UTR i+1's restarts are drawn from wherever UTR i left the generator (apa_core.py:125, :1119-1130), and that
        depends on UTR i's OUTCOME through two events only: a prune (rm_component draws init_ws + gen_k_arr for the
        K' components that stay, :843 -> :709 -> :720) and a re-run (subsample_run sweeps again, :1023-1030).  Given
        the outcome class - clean, or pruned to K' - the generator's state after UTR i is known before any EM runs.
        So the followers of a UTR are drawn from the state its PREDICTED outcome class leaves (sw.pred; the prune
        tables of that K' are drawn in stream order, ahead of their use) and ride in the same EM call.  After the call
        the results are taken in stream order exactly as far as every prediction held; from the first wrong one on,
        the generator goes back to the state saved after that UTR's own sweep draws, the UTR is finished the serial
        way and its followers are drawn again.  Every job therefore runs from the tables the serial loop gives it,
        and a job's bits do not depend on what else shares the call (tests: small-call shapes), so results, job counts
        and the generator's final state are those of the serial loop (depth 1).
        """
        active = [[smp, list(sws), 0] for smp, sws in streams if len(sws)]
        from concurrent.futures import ThreadPoolExecutor
        if len(active) < Engine.pingpong_min_streams and (depth <= 1 or not Engine.draw_ahead):
            stats, base, cur = Engine.spec_stats, dict(Engine.spec_stats), depth
            while active:
                prepared = Engine._streams_prepare(active, cur)
                active, _held = Engine._streams_absorb(active, prepared, batch.em_packed(prepared[0]), deferred, on_done)
                cur = Engine._next_depth(depth, cur, stats, base)
            return
        if len(active) < Engine.pingpong_min_streams:
            # few streams with followers.  While the GPU runs a call (worker thread, GIL released), this thread already
            # draws the tables of the NEXT call as if every prediction of the running one held - exactly what the serial
            # loop would draw then; they are thrown away otherwise (the host has nothing else to do meanwhile).
            stats, base, cur = Engine.spec_stats, dict(Engine.spec_stats), depth
            with ThreadPoolExecutor(1) as gpu:
                prepared = Engine._streams_prepare(active, cur)
                while active:
                    fut = gpu.submit(batch.em_packed, prepared[0])
                    end_state = [st[0].state.copy() for st in active]        # where each stream stands if every prediction holds
                    moved = [len(ch) for ch in prepared[1]]
                    for st, m in zip(active, moved):
                        st[2] += m
                    nxt = [st for st in active if st[2] < len(st[1])]
                    ahead = Engine._streams_prepare(nxt, cur) if nxt else None
                    for st, m in zip(active, moved):
                        st[2] -= m
                    out = fut.result()
                    was = list(active)
                    active, held = Engine._streams_absorb(active, prepared, out, deferred, on_done)
                    new_cur = Engine._next_depth(depth, cur, stats, base)
                    if ahead is not None and all(held) and new_cur == cur:
                        prepared = ahead                   # (all held: `active` is `nxt`, in the same order)
                        stats["calls_drawn_ahead"] += 1
                        continue
                    if ahead is not None:                  # never submitted: the sweeps forget the jobs they were handed,
                        for ch, ps in zip(ahead[1], ahead[2]):
                            for sw, pk in zip(ch, ps):
                                sw.n_jobs -= len(pk)
                        for st, ok, stt in zip(was, held, end_state):
                            if ok:                         # and a stream whose predictions held goes back to where its chain ended
                                st[0].state[:] = stt       # (a stream with a wrong one was put right by _streams_absorb)
                    cur = new_cur
                    if active:
                        prepared = Engine._streams_prepare(active, cur)
            return
        # many streams: two alternating halves - while the GPU runs the EM call of one half (a worker thread inside the
        # library call, GIL released), this thread takes in the other half's results and draws its next tables.  Calls
        # on the handle stay strictly one after another; which streams share a call never changes a result.
        halves = [active[0::2], active[1::2]]
        with ThreadPoolExecutor(1) as gpu:
            prepared = [Engine._streams_prepare(h, depth) for h in halves]
            futs = [gpu.submit(batch.em_packed, pr[0]) for pr in prepared]
            h = 0
            while halves[0] or halves[1]:
                if halves[h]:
                    out = futs[h].result()
                    halves[h], _held = Engine._streams_absorb(halves[h], prepared[h], out, deferred, on_done)
                    if halves[h]:
                        prepared[h] = Engine._streams_prepare(halves[h], depth)
                        futs[h] = gpu.submit(batch.em_packed, prepared[h][0])
                h ^= 1

    pingpong_min_streams = 32    # fewer streams than this: one call at a time (a half would leave the GPU under-filled)
    draw_ahead = True            # few streams with followers: draw the next call's tables while the GPU runs the current one

    @staticmethod
    def _next_depth(depth, cur, stats, base):
        """Followers pay while the predictions hold: with p of them right a chain keeps ~1/(1-p) UTRs, the rest is
        discarded work - the depth follows the share seen so far in this run (speed only)."""
        n_pred = stats["predicted"] - base["predicted"]
        if depth > 2 and n_pred >= 8:
            p_ok = (stats["predicted_right"] - base["predicted_right"]) / n_pred
            return depth if p_ok >= 0.75 else (max(2, depth // 2) if p_ok >= 0.5 else 2)
        return cur

    @staticmethod
    def _streams_prepare(active, depth):
        """Draw the tables of the next EM call: up to `depth` UTRs of every active stream (see _drive_streams).
        With depth > 1 the generator state after every sweep's own draws is kept, and the prune tables of a predicted
        K' are drawn right behind the sweep's (also for the last UTR of a chain: the stream then stands where the
        next call starts if the prediction holds)."""
        chains = [st[1][st[2]:st[2] + depth] for st in active]
        packs = [[] for _ in active]
        marks = [[] for _ in active]      # generator state after each sweep's own draws
        ahead = [[] for _ in active]      # prune tables drawn ahead: (K', init_ws, k_arr) or None
        for pos in range(depth):          # position by position: the streams' sweeps of one position share a threaded call
            col = [(si, ch[pos]) for si, ch in enumerate(chains) if pos < len(ch)]
            if not col:
                break
            for (si, sw), pk in zip(col, _Sweep.make_packed_many([sw for _si, sw in col])):
                smp = active[si][0]
                packs[si].append(pk)
                marks[si].append(smp.state.copy() if depth > 1 else None)
                pre = None
                if depth > 1:
                    pred = sw.pred if sw.best is None else "stop"      # a re-run's sweep has no prediction of its own
                    if pred == "stop":
                        chains[si] = chains[si][:pos + 1]
                    elif isinstance(pred, tuple):
                        Kp = pred[1]
                        pre = (Kp, smp.init_ws(Kp, sw.prep.p["max_unif_ws"]), smp.k_arr(Kp))
                ahead[si].append(pre)
        return concat_packed([pk for ps in packs for pk in ps]), chains, packs, marks, ahead

    @staticmethod
    def _streams_absorb(active, prepared, out, deferred, on_done):
        """Take the results of one EM call in stream order, as far as the predictions held.  Returns (the streams that
        still have UTRs, per stream of `active`: did every prediction of its chain hold - the generator then stands
        where _streams_prepare left it)."""
        pj, chains, packs, marks, ahead = prepared
        stats = Engine.spec_stats
        stats["calls"] += 1
        lo, nxt, held = 0, [], []
        for st, ch, ps, mk, ah in zip(active, chains, packs, marks, ahead):
            smp, live = st[0], True
            for j, (sw, pk) in enumerate(zip(ch, ps)):
                hi = lo + len(pk)
                if not live:
                    sw.n_jobs -= len(pk)               # drawn from a state the stream never reached: discarded
                    stats["utrs_discarded"] += 1
                    lo = hi
                    continue
                sw.absorb_packed(pj, out, lo, hi)
                lo = hi
                stats["utrs_kept"] += 1
                pre = ah[j]
                last = j + 1 == len(ch)
                if not last:
                    stats["predicted"] += 1
                if sw.done:                            # clean end
                    if pre is not None:                # ... but what follows was drawn past a prune's draws
                        smp.state[:] = mk[j]
                        live = False
                elif sw.stage == "prune" and sw.trace is None and Engine.defer_prunes:
                    # the re-fit's tables belong here in the stream; the EM call waits for the wave's other re-fits
                    if pre is not None and pre[0] == sw.prune_K():
                        deferred.append((sw, sw.make_packed(predrawn=pre[1:])))
                    else:
                        if mk[j] is not None:
                            smp.state[:] = mk[j]
                            live = False
                        deferred.append((sw, sw.make_packed()))
                else:                                  # a re-run sweep (or a traced prune): stays the head of its stream
                    if mk[j] is not None:
                        smp.state[:] = mk[j]
                    live = False
                    continue
                if live and not last:
                    stats["predicted_right"] += 1
                st[2] += 1
                if on_done is not None:
                    on_done(sw)
            held.append(live)
            if st[2] < len(st[1]):
                nxt.append(st)
        return nxt, held

    @staticmethod
    def _drive(batch, sweeps, deferred=None):
        pending = list(sweeps)
        while pending and all(sw.trace is None for sw in pending):      # fast path: padded tables end to end
            packs = _Sweep.make_packed_many(pending)
            pj = concat_packed(packs)
            out = batch.em_packed(pj)
            lo = 0
            for sw, pk in zip(pending, packs):
                sw.absorb_packed(pj, out, lo, lo + len(pk))
                lo += len(pk)
            pending = [sw for sw in pending if not sw.done]
            if deferred is not None:
                pending = Engine._defer_prunes(pending, deferred)
        while pending:
            jobs, spans = [], []
            for sw in pending:
                js = sw.make_jobs()
                spans.append((len(jobs), len(jobs) + len(js)))
                jobs.extend(js)
            fits = batch.em(jobs)
            for sw, (a, b) in zip(pending, spans):
                sw.absorb(fits[a:b])
            pending = [sw for sw in pending if not sw.done]
