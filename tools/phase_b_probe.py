#!/usr/bin/env python3
"""Where Phase B's time goes: times scape_hip_batch_build on the bench's default batch with parts of
k_phase_b switched off (SCAPE_HIP_PB_PROBE bit mask; results are wrong with it, timing only)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scape_amd.engine import Engine, HipBatch   # noqa: E402
from scape_amd.host import prepare_utr          # noqa: E402
from scape_amd.synth import synth_utr           # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 512
preps = [prepare_utr(df, gene_info_str=g, n_max_apa=10, n_min_apa=1)
         for g, df, _ in (synth_utr(i, 2000, k_cap=10, base_seed=20250225) for i in range(U))]
eng = Engine()
batch = HipBatch(eng.ctx, preps)
lib, h = eng.ctx.lib, eng.ctx.h
for probe in (0, 1, 2, 3, 0):
    os.environ["SCAPE_HIP_PB_PROBE"] = str(probe)
    batch.build()
    lib.scape_hip_timing_reset(h)
    for _ in range(3):
        batch.build()
    out = []
    for which in (0, 1):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        lib.scape_hip_timing_get(h, which, ctypes.byref(ms), ctypes.byref(n))
        out.append(ms.value / max(n.value, 1))
    print(f"probe {probe}: phase A {out[0]:.2f} ms, phase B (+extents) {out[1]:.2f} ms", flush=True)
