"""Where a reference-stream UTR spends its time: wall time of the EM calls vs the kernels inside them.
GPU box: python tools/reference_mode_breakdown.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scape_amd.engine import Engine          # noqa: E402
from scape_amd.host import prepare_utr       # noqa: E402
from scape_amd.synth import synth_utr        # noqa: E402

kw = dict(n_max_apa=10, n_min_apa=1)
preps = []
NU = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for i in range(NU):
    g, df, _ = synth_utr(i, 2000, k_cap=10, base_seed=20250225)
    preps.append(prepare_utr(df, gene_info_str=g, **kw))
eng = Engine(0)
eng.run(preps[:4], rng_mode="reference", seed=1)
lib, h = eng.ctx.lib, eng.ctx.h
for fine in (False, True):
    if fine:
        os.environ["SCAPE_HIP_ROUND_TIMING"] = "1"
    lib.scape_hip_timing_reset(h)
    t = time.perf_counter()
    eng.run(preps, rng_mode="reference", seed=1, re_run_mode=True)
    dt = time.perf_counter() - t
    print(f"per-round events {'on' if fine else 'off'}: {dt * 1e3 / len(preps):.2f} ms wall per UTR")
    for w, name in ((0, "phase_a"), (1, "phase_b"), (2, "em calls (event around the whole call)"), (4, "k2_estep"), (5, "k2_mstep")):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        lib.scape_hip_timing_get(h, w, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            print(f"   {name}: {ms.value / len(preps):.3f} ms per UTR, {n.value} launches, {1e3 * ms.value / n.value:.1f} us each")
