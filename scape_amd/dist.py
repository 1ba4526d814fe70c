"""Multi-GPU sharding of ``infer_pa``: UTRs are independent problems (reference
``apa_core.py:1119-1130``), so a node with G GPUs runs G processes (one per GPU), each on its own
shard of the UTR list; there is no data-path collective.  Rank 0 gathers the per-UTR results on
the host and restores the input order (the reference writes one Parameters per input tuple, in
order: ``apa_core.py:1134-1137``).  ``torch.distributed`` is used only as rendezvous plumbing for
that object gather (RCCL is never asked to move tensor data).
"""
from __future__ import annotations

import os

import numpy as np


def utr_cost(n_frag, n_theta, n_beta, n_max_apa, n_min_apa=1, n_trial=10):
    """Work estimate of one UTR: tensor elements x restarts x sum of K."""
    ksum = sum(range(n_min_apa, n_max_apa + 1))
    return float(n_frag) * n_theta * n_beta * n_trial * ksum


def lpt_partition(costs, world):
    """Greedy longest-processing-time assignment; returns `world` sorted index lists."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    shards = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += costs[i]
    return [sorted(s) for s in shards]


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def gather_in_order(local_items, n_total, group=None):
    """local_items: list of (global_index, obj).  Returns the ordered list of n_total objects on
    rank 0 and None elsewhere.  Single-process (no initialised process group): plain reorder."""
    try:
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized()
    except Exception:
        live = False
    if not live:
        parts = [local_items]
        rank = 0
    else:
        rank = dist.get_rank(group)
        world = dist.get_world_size(group)
        parts = [None] * world if rank == 0 else None
        dist.gather_object(local_items, parts, dst=0, group=group)
    if rank != 0:
        return None
    out = [None] * n_total
    seen = 0
    for part in parts:
        for gi, obj in part:
            if out[gi] is not None:
                raise RuntimeError(f"UTR {gi} was processed by two ranks")
            out[gi] = obj
            seen += 1
    if seen != n_total:
        raise RuntimeError(f"gather: {seen} of {n_total} UTRs arrived")
    return out


def run_sharded(items, costs, run_fn, rank=None, world=None, group=None):
    """Shard `items` by LPT over `costs`, run `run_fn(list_of_items) -> list_of_results` on this
    rank's shard and gather on rank 0 in input order."""
    if rank is None or world is None:
        rank, _local, world = env_rank_world()
    shards = lpt_partition(costs, world)
    mine = shards[rank]
    res = run_fn([items[i] for i in mine]) if mine else []
    if len(res) != len(mine):
        raise RuntimeError("run_fn must return one result per item")
    return gather_in_order(list(zip(mine, res)), len(items), group=group)


def infer_chunk_sharded(pkl_input_file, pickle_output_file, device=None, **kwargs):
    """infer() of one chunk across the ranks of an initialised process group (one rank per GPU):
    every rank reads the chunk, prepares and processes its shard in 'per_utr' RNG mode (seed + UTR
    index, so the result of a UTR does not depend on the sharding), rank 0 writes the pickle."""
    import pickle

    from .apa_core import read_input_chunk, to_parameters
    from .engine import Engine
    from .host import prepare_utr
    rank, local, world = env_rank_world()
    utrs = list(read_input_chunk(pkl_input_file, frames="columns"))
    # The shards are cut BEFORE binning, from what the raw columns tell at a glance (reads ~ bins, the read span ~ the
    # theta grid), so that a rank bins and prepares only its own UTRs - preparing all of them on every rank was host
    # work x world size.  Every rank computes the same estimate, hence the same partition.
    from .host import model_params
    p0 = model_params(kwargs)
    n_beta = max(1, int(round(p0["max_beta"] / p0["beta_step"])) - 1)

    def estimate(df):
        x, l = np.asarray(df["x"]), np.asarray(df["l"])
        L = max(2000, int(x.max()) + int(l.max()) + 50, int(kwargs.get("utr_length", -1) or -1))
        T = max(1, (L - int(l.min())) // max(1, int(p0["theta_step"])))
        return utr_cost(len(x), T, n_beta, p0["n_max_apa"], p0["n_min_apa"])
    costs = [estimate(df) for _g, df in utrs]
    seed = int(kwargs.get("seed", 1))
    eng = Engine(device=local if device is None else device)

    def run_fn(idx_utrs):
        # per-UTR seed follows the global index, so results do not depend on the sharding
        preps = [prepare_utr(df, gene_info_str=g, **kwargs) for _gi, (g, df) in idx_utrs]
        res = eng.run(preps, rng_mode="per_utr",
                      seeds=[(seed + gi) % (2 ** 32) for gi, _u in idx_utrs],
                      re_run_mode=bool(kwargs.get("re_run_mode", True)))
        return [to_parameters(r) for r in res]

    res = run_sharded(list(enumerate(utrs)), costs, run_fn, rank=rank, world=world)
    if rank == 0:
        with open(pickle_output_file, "wb") as fh:
            for para in res:
                pickle.dump(para, fh)
    return res
