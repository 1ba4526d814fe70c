#!/usr/bin/env python3
"""Benchmark of the infer_pa hot path on MI355X (contract: see the round driver's README).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the headline): synthetic UTRs x 2,000 reads, K swept 10..1,
10 restarts each (1,000 em_algo jobs... 100 per UTR), drawn from the 50k-UTR synthetic stream
(scape_amd/synth.py, seed = base + index).  One *step* = one pass of the hot path over the batch
resident in HBM: Phase A -> Phase B -> EM sweep -> BIC selection -> prune re-fits -> labels, results
back on the host.  Binned inputs and the restart tables are uploaded / drawn before the timed region.
With N > 1 every rank processes its own batch of the stream (weak scaling, no data-path collective);
value = UTRs all ranks processed / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utrs", type=int, default=256, help="UTRs per GPU per step")
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--kcap", type=int, default=10)
    ap.add_argument("--base-seed", type=int, default=20250225)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU work for the baseline sample")
    return ap.parse_args()


def cpu_baseline(preps, plan, target_s, cores):
    """The oracle (CPU restatement of the reference algorithm, 'port') on a bounded sample of the
    same workload: Phase A + Phase B + the same EM jobs per UTR, UTRs spread over `cores` threads."""
    from oracle import scape_oracle as so
    lib = so.lib()
    pj, spans = plan["main"], plan["spans"]
    D, I64, I32 = so._D, so._I64, so._I32

    def one(u):
        q = preps[u]
        lo, hi = int(spans[u]), int(spans[u + 1])
        nj, kmax = hi - lo, pj.kmax
        K = pj.jk[lo:hi].astype(np.int32)
        alpha = np.zeros((nj, kmax))
        beta = np.zeros((nj, kmax))
        for j in range(nj):
            alpha[j, :K[j]] = q.theta[pj.a[lo + j, :K[j]]]
            beta[j, :K[j]] = q.betas[pj.b[lo + j, :K[j]]]
        ws = np.ascontiguousarray(pj.w[lo:hi]).copy()
        ka = np.ascontiguousarray(pj.ka[lo:hi].astype(np.int64))
        fixed = np.zeros(nj, dtype=np.int32)
        bic = np.zeros(nj)
        lb = np.zeros((nj, 50))
        nlb = np.zeros(nj, dtype=np.int32)
        x, l, r, pa, cnt, th, be = (np.ascontiguousarray(v, dtype=np.float64) for v in
                                    (q.x, q.l, q.r, q.pa, q.cnt, q.theta, q.betas))
        s, pmf = np.ascontiguousarray(q.s_dis, dtype=np.float64), np.ascontiguousarray(q.pmf_s, dtype=np.float64)
        p = lambda a, t=D: a.ctypes.data_as(t)
        rc = lib.so_utr_jobs(p(x), p(l), p(r), p(pa), p(cnt), ctypes.c_int(q.N), p(th), ctypes.c_int(q.T),
                             p(be), ctypes.c_int(len(be)), p(s), p(pmf), ctypes.c_int(len(s)),
                             ctypes.c_double(q.p["mu_f"]), ctypes.c_double(q.p["sigma_f"]),
                             ctypes.c_double(q.unif_ll), ctypes.c_double(q.L), ctypes.c_double(q.min_theta),
                             ctypes.c_double(q.p["max_unif_ws"]), ctypes.c_int(nj), ctypes.c_int(kmax),
                             p(K, I32), p(fixed, I32), p(alpha), p(beta), p(ws), p(ka, I64), ctypes.c_int(50),
                             p(bic), p(lb), p(nlb, I32))
        assert rc == 0
        return u, alpha, beta, ws, bic, nlb

    t0 = time.perf_counter()
    first = one(0)
    t1 = time.perf_counter() - t0
    n = int(min(len(preps) - 1, max(cores, round(cores * target_s / max(t1, 1e-3)))))
    n = max(n, 1)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        outs = list(ex.map(one, range(1, 1 + n)))
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="UTRs/s", cores=cores, kind="port",
                sample=f"{n} UTRs of the same batch (Phase A+B + the same {int(spans[1] - spans[0])} EM jobs/UTR) "
                       f"on {cores} threads, {dt:.1f} s wall; 1 UTR on 1 thread = {t1:.2f} s"), [first] + outs


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        try:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            tdev = torch.device("cuda", local_rank)
        except Exception:
            dist.init_process_group(backend="gloo")
            tdev = torch.device("cpu")

    from scape_amd.engine import Engine, HipBatch
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr

    U = args.utrs
    kw = dict(n_max_apa=args.kcap, n_min_apa=1)
    t_prep = time.perf_counter()
    preps = []
    for i in range(U):
        gene, df, _truth = synth_utr(rank * U + i, args.reads, k_cap=args.kcap, base_seed=args.base_seed)
        preps.append(prepare_utr(df, gene_info_str=gene, **kw))
    eng = Engine(device=local_rank)
    plan = eng.plan(preps, [(args.base_seed + rank * U + i) % 2 ** 32 for i in range(U)])
    t_prep = time.perf_counter() - t_prep
    t_h2d = time.perf_counter()
    batch = HipBatch(eng.ctx, preps)            # inputs resident in HBM from here on
    t_h2d = time.perf_counter() - t_h2d

    def sync_all():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        return eng.process(batch, preps, plan, re_run_mode=False)

    for _ in range(args.warmup):
        res = step()
    lib, h = eng.ctx.lib, eng.ctx.h
    lib.scape_hip_timing_reset(h)
    sync_all()
    em_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        em_ms += eng.last_main_em_ms
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel timing (HIP events on the library's stream) and algorithmic bytes of the EM launch
    kern = {}
    for which, name in enumerate(("phase_a", "phase_b", "em_all", "labels")):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        lib.scape_hip_timing_get(h, which, ctypes.byref(ms), ctypes.byref(n))
        kern[name] = dict(ms_total=ms.value, launches=n.value)
    rounds, slab, zel = eng.last_main_counters
    alg_bytes = 8.0 * slab + 16.0 * zel                   # slab reads + Z/log_z traffic
    em_avg_ms = em_ms / args.steps
    achieved = alg_bytes / (em_avg_ms * 1e-3) / 1e9

    if rank == 0:
        N = np.array([q.N for q in preps])
        T = np.array([q.T for q in preps])
        out = {
            "metric": "UTR regions/sec (infer_pa EM to convergence), 50k UTRs x 2k reads synthetic",
            "value": world * U * args.steps / elapsed, "unit": "UTRs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"50k-UTR synthetic stream x {args.reads} reads, K=1..{args.kcap} x 10 restarts; "
                                   f"step = one resident batch of {U} UTRs per GPU",
                       "utrs_per_gpu_per_step": U, "reads_per_utr": args.reads, "k_cap": args.kcap,
                       "em_jobs_per_step": int(len(plan["main"])), "n_frag_mean": float(N.mean()),
                       "n_frag_max": int(N.max()), "n_theta_mean": float(T.mean()), "re_run_mode": False,
                       "rng_mode": "per_utr", "parallelism": f"utr-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "k_em (main K sweep launch)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": em_avg_ms,
                         "em_rounds_per_launch": int(rounds)},
            "kernels_ms": kern,
            "host": {"prep_s": t_prep, "h2d_s": t_h2d},
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = min(16, os.cpu_count() or 1)
            cb, outs = cpu_baseline(preps, plan, args.cpu_seconds, cores)
            out["cpu_baseline"] = cb
            # parity of the sampled UTRs' winners (GPU step result vs the oracle's jobs + same selection)
            same = 0
            for u, alpha, beta, ws, bic, nlb in outs:
                lo = int(plan["spans"][u])
                grid = bic.reshape(-1, 10)
                tb = np.argmin(grid, axis=1)
                kb = int(np.argmin(grid[np.arange(len(tb)), tb]))
                j = kb * 10 + int(tb[kb])
                K = int(plan["main"].jk[lo + j])
                fit = res[u][0]
                q = preps[u]
                pruned = bool(np.any(ws[j, :K] < q.p["min_ws"]))
                if pruned:
                    keep = ~(ws[j, :K] < q.p["min_ws"])
                    same += (fit.K == int(keep.sum()) and np.array_equal(q.theta[fit.a_idx], alpha[j, :K][keep])
                             and np.array_equal(q.betas[fit.b_idx], beta[j, :K][keep]))
                else:
                    same += (fit.K == K and np.array_equal(q.theta[fit.a_idx], alpha[j, :K])
                             and np.array_equal(q.betas[fit.b_idx], beta[j, :K])
                             and np.allclose(fit.ws, ws[j, :K + 1], rtol=1e-4, atol=1e-9))
            out["cpu_baseline"]["parity_sample"] = f"{same}/{len(outs)} sampled UTRs: GPU pA calls identical to the CPU port"
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / cb["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
