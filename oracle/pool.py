"""Many ``oracle.subsample_run`` calls on the host's cores (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

The oracle follows the reference in drawing from numpy's GLOBAL random stream (apa_core.py:125), so UTRs cannot
share a process concurrently: every task runs in a worker process that seeds the global stream for its UTR.
Workers come from a fork server (no exec from, and no copy of, a process that may hold a HIP context).

A task is ``(index, n_reads, k_cap_synth, base_seed, seed, re_run_mode, kw)``: the synthetic UTR
``scape_amd.synth.synth_utr(index, n_reads, k_cap_synth, base_seed)`` (inputs only - the generator is not part of
the path under test) fitted by ``subsample_run(..., re_run_mode, **kw)`` after ``np.random.seed(seed)``.
"""
from __future__ import annotations

import multiprocessing as mp
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np


def fit_one(task):
    """-> dict of the final Parameters fields the reference's consumers read."""
    index, n_reads, k_cap_synth, base_seed, seed, re_run_mode, kw = task
    from oracle import scape_oracle as so
    from scape_amd.synth import synth_utr
    g, df, _ = synth_utr(index, n_reads, k_cap=k_cap_synth, base_seed=base_seed)
    np.random.seed(seed)
    res, model = so.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values,
                                  re_run_mode=re_run_mode, **kw)
    return dict(gene=g, K=int(res.K), alpha=np.asarray(res.alpha_arr), beta=np.asarray(res.beta_arr),
                ws=np.asarray(res.ws), bic=float(res.bic), lb=np.asarray(res.lb_arr, dtype=np.float64),
                labels=np.asarray(res.label_arr), n_calls=len(model.calls), n_max_apa=int(model.n_max_apa))


def fit_many(tasks, workers=None):
    workers = workers or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    workers = max(1, min(workers, len(tasks)))
    with ProcessPoolExecutor(workers, mp_context=mp.get_context("forkserver")) as ex:
        return list(ex.map(fit_one, tasks))
