"""Single-chunk rate of the exact-stream mode against the number of UTRs of the stream per EM call
(Engine.spec_depth; 1 = strictly serial), with the kept / discarded follower counts.
GPU box: python tools/stream_followers_rate.py [n_utr] [reads] [kcap]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scape_amd.engine import Engine
from scape_amd.host import prepare_utr
from scape_amd.synth import synth_utr

U = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
KC = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kw = dict(n_max_apa=KC, n_min_apa=1)
preps = []
for i in range(U):
    g, df, _ = synth_utr(2 * 10 ** 6 + i, R, k_cap=KC, base_seed=20250225)
    preps.append(prepare_utr(df, gene_info_str=g, **kw))
eng = Engine(0)
eng.run(preps, rng_mode="reference", seed=1)           # warm-up
ref = None
for depth in (1, 2, 4, 8, 12, 16, 24, 32):
    Engine.spec_depth = depth
    b = dict(Engine.spec_stats)
    dt = 1e30
    for _ in range(2):
        t = time.perf_counter(); res = eng.run(preps, rng_mode="reference", seed=1, re_run_mode=True); dt = min(dt, time.perf_counter() - t)
    d = {k: (Engine.spec_stats[k] - b[k]) // 2 for k in b}
    sig = [(r.fit.K, tuple(r.fit.a_idx), tuple(r.fit.ws), r.n_jobs) for r in res]
    ref = ref or sig
    print(f"depth {depth:2d}: {U / dt:7.1f} UTRs/s  {dt * 1e3 / U:.2f} ms/UTR  calls {d['calls']}  kept {d['utrs_kept']}  discarded {d['utrs_discarded']}  predictions {d['predicted_right']}/{d['predicted']}  drawn ahead {d['calls_drawn_ahead']}  identical {sig == ref}", flush=True)
nj = np.array([r.n_jobs for r in res])
print("job counts per UTR:", dict(zip(*np.unique(nj, return_counts=True))))
