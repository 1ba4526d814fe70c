"""Engine.run_streams (exact per-chunk reference random streams) vs number of chunk files in flight.
GPU box: python tools/streams_scaling.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scape_amd.engine import Engine          # noqa: E402
from scape_amd.host import prepare_utr       # noqa: E402
from scape_amd.synth import synth_utr        # noqa: E402

kw = dict(n_max_apa=10, n_min_apa=1)
U = 12
preps = []
for i in range(256 * U):
    g, df, _ = synth_utr(i, 2000, k_cap=10, base_seed=20250225)
    preps.append(prepare_utr(df, gene_info_str=g, **kw))
eng = Engine(0)
eng.run(preps[:4], rng_mode="reference", seed=1)
for rep in range(2):
    for depth, pp in ((1, 10 ** 9), (8, 10 ** 9), (8, 2)):   # UTRs of one stream per EM call (1 = heads only); two-halves schedule from pp streams on
        Engine.spec_depth, Engine.pingpong_min_streams = depth, pp
        for F in (2, 4, 8, 16, 32, 64, 128, 256):
            if F * U > len(preps) or (pp == 2 and F < 16):
                continue
            streams = [(preps[f * U:(f + 1) * U], 1) for f in range(F)]
            b = dict(Engine.spec_stats)
            t = time.perf_counter()
            eng.run_streams(streams, re_run_mode=True)
            dt = time.perf_counter() - t
            d = {k: Engine.spec_stats[k] - b[k] for k in b}
            print(f"spec_depth={depth} halves={pp == 2} (per-stream depth {eng._stream_depth(F)}) {F} chunks x {U} UTRs in flight: {dt:.2f} s = "
                  f"{F * U / dt:.0f} UTRs/s  calls {d['calls']} discarded {d['utrs_discarded']}", flush=True)
