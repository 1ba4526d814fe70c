"""Synthetic UTR pile-ups drawn from the model's own generative story.

The reference ships no generator; reads are produced the way its likelihood assumes them
(``taichi_core.py:56-157``): a pA component picks the cleavage position theta ~ N(alpha_k,
beta_k), the polyA length s is uniform on {20,30,...,140}, the read-2 start is
x ~ N(theta + s - mu_f, sigma_f) and its aligned length l is uniform up to the room left
before theta; a small fraction of reads carries the observed cleavage position ``pa``;
``r`` is always NaN (10x data, ``input_processor.py:425-426``).  Seed = base_seed + utr index.
"""
from __future__ import annotations

import numpy as np
import pandas as pd


def synth_utr(index, n_reads, k_cap=5, base_seed=0, pa_rate=0.015, noise=0.05, r_rate=0.0):
    rng = np.random.default_rng(base_seed + index)
    L_true = int(rng.integers(1500, 4001))
    K = int(rng.integers(1, min(5, k_cap) + 1))
    # sorted alphas on [300, L_true-200] with gaps >= 150
    span = L_true - 200 - 300 - 150 * (K - 1)
    while span <= 0:
        K -= 1
        span = L_true - 200 - 300 - 150 * (K - 1)
    alphas = 300 + np.sort(rng.random(K)) * span + 150 * np.arange(K)
    betas = rng.integers(10, 51, size=K).astype(np.float64)
    ws = rng.dirichlet(np.full(K, 2.0))
    comp = rng.choice(K, size=n_reads, p=ws)
    theta = rng.normal(alphas[comp], betas[comp])
    s = rng.choice(np.arange(20, 150, 10), size=n_reads)
    x = np.rint(rng.normal(theta + s - 300, 50))
    x = np.clip(x, 0, np.maximum(theta - 31, 0))
    room = np.maximum(theta - x, 31)
    l = np.floor(31 + rng.random(n_reads) * (np.minimum(132, room) - 31 + 1))
    l = np.clip(l, 31, 132)
    is_noise = rng.random(n_reads) < noise
    nn = int(is_noise.sum())
    x[is_noise] = np.floor(rng.random(nn) * (L_true - 132))
    l[is_noise] = rng.integers(31, 133, size=nn)
    pa = np.full(n_reads, np.nan)
    has_pa = (rng.random(n_reads) < pa_rate) & ~is_noise
    pa[has_pa] = np.rint(theta[has_pa])
    r = np.full(n_reads, np.nan)
    if r_rate > 0:
        has_r = (rng.random(n_reads) < r_rate) & ~has_pa & ~is_noise
        r[has_r] = np.minimum(np.floor(rng.random(int(has_r.sum())) * s[has_r]) + 1, s[has_r])
    # the three columns merge_pa reads (input_processor.py:636): a few junction reads with their segment ends in
    # genomic coordinates; drawn from a generator of their own, so the model columns above do not depend on them
    jrng = np.random.default_rng([base_seed + index, 7])
    junction = (jrng.random(n_reads) < 0.03).astype(np.int64)
    seg1 = np.where(junction == 1, 1.0 + x + np.floor(jrng.random(n_reads) * l), np.nan)
    seg2 = np.where(junction == 1, 1.0 + x + l, np.nan)
    df = pd.DataFrame({"x": x.astype(np.int64), "l": l.astype(np.int64), "r": r, "pa": pa,
                       "cb_id": np.arange(n_reads, dtype=np.int64),
                       "read_id": np.arange(n_reads, dtype=np.int64),
                       "junction": junction, "seg1_en": seg1, "seg2_en": seg2})
    gene = f"syn:G{index:07d}:1:1-{L_true}:+"
    truth = dict(alphas=alphas, betas=betas, ws=ws, L_true=L_true)
    return gene, df, truth


def synth_chunk(n_utr, n_reads, k_cap=5, base_seed=0, start=0, **kw):
    return [synth_utr(start + i, n_reads, k_cap=k_cap, base_seed=base_seed, **kw)[:2] for i in range(n_utr)]
