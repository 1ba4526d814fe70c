"""Rate of the CLI-default 'reference' RNG mode (one RandomState per chunk, UTR-serial by construction) and of
run_streams (several chunk files' streams sharing launches) at the headline shape.  GPU box: python tools/reference_mode_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scape_amd.engine import Engine
from scape_amd.host import prepare_utr
from scape_amd.synth import synth_utr

U, F = 48, 32
kw = dict(n_max_apa=10, n_min_apa=1)
files = []
for f in range(F):
    preps = []
    for i in range(U):
        g, df, _ = synth_utr(f * U + i, 2000, k_cap=10, base_seed=20250225)
        preps.append(prepare_utr(df, gene_info_str=g, **kw))
    files.append(preps)
eng = Engine(0)
eng.run(files[0][:8], rng_mode="reference", seed=1)           # warm-up
t = time.perf_counter(); res = eng.run(files[0], rng_mode="reference", seed=1, re_run_mode=True); dt = time.perf_counter() - t
print(f"reference mode, one chunk of {U} UTRs: {dt:.2f} s = {U / dt:.1f} UTRs/s; EM jobs/UTR {np.mean([r.n_jobs for r in res]):.1f}")
t = time.perf_counter(); out = eng.run_streams([(p, 1) for p in files], re_run_mode=True); dt = time.perf_counter() - t
print(f"run_streams, {F} chunks x {U} UTRs: {dt:.2f} s = {F * U / dt:.1f} UTRs/s")
t = time.perf_counter(); res2 = eng.run(files[0], rng_mode="per_utr", seed=1, re_run_mode=True); dt = time.perf_counter() - t
print(f"per_utr mode, {U} UTRs (re_run on): {dt:.2f} s = {U / dt:.1f} UTRs/s")
