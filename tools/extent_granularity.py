"""How many tensor bytes finer finite extents would save the M-step: for a few headline UTRs, the bins a full pass
streams with one extent per 64-row tile (what k2_mstep uses), per 16-row group and per row.
GPU box: python tools/extent_granularity.py [n_utr]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scape_amd.engine import Engine, HipBatch
from scape_amd.host import prepare_utr
from scape_amd.synth import synth_utr

U = int(sys.argv[1]) if len(sys.argv) > 1 else 6
SENT = float(np.finfo("f").min)
preps = [prepare_utr(df, gene_info_str=g, n_max_apa=10, n_min_apa=1)
         for g, df, _ in (synth_utr(i, 2000, k_cap=10, base_seed=20250225) for i in range(U))]
eng = Engine(0)
batch = HipBatch(eng.ctx, preps)
batch.build()
tot = dict(tile=0, group=0, row=0, full=0)
for u, q in enumerate(preps):
    M = batch.fetch_tensor(u).reshape(-1, q.N)                    # [T*B, N]
    fin = M != SENT
    ext = np.where(fin.any(axis=1), q.N - np.argmax(fin[:, ::-1], axis=1), 0)     # one past the last finite bin of each row
    ext16 = (ext + 15) // 16 * 16
    R = len(ext)
    tot["full"] += R * q.N
    tot["row"] += int(ext16.sum())
    for name, h in (("tile", 64), ("group", 16)):
        pad = (-R) % h
        e = np.concatenate([ext16, np.zeros(pad, ext16.dtype)]).reshape(-1, h)
        rows = np.concatenate([np.ones(R), np.zeros(pad)]).reshape(-1, h).sum(axis=1)
        tot[name] += int((e.max(axis=1) * rows).sum())
for k in ("tile", "group", "row"):
    print(f"extent per {k:5s}: {tot[k] / tot['full']:.4f} of the tensor, {tot[k] / tot['tile']:.4f} of the per-tile scheme")
