"""cProfile of Engine.run_streams (reference RNG streams of several chunk files sharing launches) at the
headline shape + per-kernel event totals.  GPU box: python tools/profile_streams.py"""
import cProfile
import ctypes
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scape_amd.engine import Engine          # noqa: E402
from scape_amd.host import prepare_utr       # noqa: E402
from scape_amd.synth import synth_utr        # noqa: E402

kw = dict(n_max_apa=10, n_min_apa=1)
files = []
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for f in range(NF):
    preps = []
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 32):
        g, df, _ = synth_utr(f * 32 + i, 2000, k_cap=10, base_seed=20250225)
        preps.append(prepare_utr(df, gene_info_str=g, **kw))
    files.append(preps)
eng = Engine(0)
eng.run(files[0][:8], rng_mode="reference", seed=1)
pr = cProfile.Profile()
pr.enable()
eng.run_streams([(p, 1) for p in files], re_run_mode=True)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
os.environ["SCAPE_HIP_ROUND_TIMING"] = "1"
eng.ctx.lib.scape_hip_timing_reset(eng.ctx.h)
t = time.perf_counter()
eng.run_streams([(p, 1) for p in files], re_run_mode=True)
print("with per-round events", time.perf_counter() - t, "s")
for w in (2, 4, 5):
    ms, n = ctypes.c_double(), ctypes.c_int32()
    eng.ctx.lib.scape_hip_timing_get(eng.ctx.h, w, ctypes.byref(ms), ctypes.byref(n))
    print("kind", w, round(ms.value, 2), "ms", n.value, "launches")
