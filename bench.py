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
# HBM bytes per k2_mstep launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) at
# the default workload (512 UTRs x 2k reads per step); see profiles/README.md.
PMC_TRAFFIC_BYTES_PER_LAUNCH = 11.2e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utrs", type=int, default=512, help="UTRs per GPU per step")
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--kcap", type=int, default=10)
    ap.add_argument("--base-seed", type=int, default=20250225)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-utrs", type=int, default=16384,
                    help="UTRs of the untimed-by-the-metric end-to-end leg (chunk files -> .res.pkl), 0 = skip")
    ap.add_argument("--e2e-workers", type=int, default=0, help="prep processes of the end-to-end leg (0 = auto)")
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


def parity_count(preps, plan, res, outs):
    """How many of the UTRs the CPU port ran (`outs` of cpu_baseline) got the same pA calls from the GPU step
    (`res` of Engine.process): the port's jobs go through the reference's selection (float32 BIC arg-mins,
    apa_core.py:849/:865/:945/:972) and pruning rule (:832-844) and are compared with the GPU's final fit."""
    same = 0
    for u, alpha, beta, ws, bic, nlb in outs:
        lo = int(plan["spans"][u])
        grid = bic.astype(np.float32).reshape(-1, 10)      # the reference's bic_arr is float32
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
    return int(same)


def end_to_end(args, pool, dev):
    """The whole `infer_pa` job as a user runs it, on a bounded sample of the same synthetic stream:
    prepare_input-style chunk files on disk -> non-executing unpickle -> binning / coverage peaks (process
    pool) -> restart tables (native sampler) -> GPU -> Parameters -> pkl_output/*.res.pkl.  SURVEY.md 8(d):
    the rate WITH host pre/post-processing, reported beside `value` (which is the resident-batch rate)."""
    import shutil
    import tempfile
    from scape_amd.apa_core import infer_files
    from scape_amd.pipeline import synth_chunk_file
    root = tempfile.mkdtemp(prefix="scape_e2e_")
    try:
        os.makedirs(os.path.join(root, "pkl_input"))
        per_file = 128
        tasks = [(os.path.join(root, "pkl_input", f"synth.{per_file}.{i}.input.pkl"), 10 ** 6 + i * per_file,
                  min(per_file, args.e2e_utrs - i * per_file), args.reads, args.kcap, args.base_seed)
                 for i in range((args.e2e_utrs + per_file - 1) // per_file)]
        t0 = time.perf_counter()
        files = list(pool.ex.map(synth_chunk_file, tasks))
        t_write = time.perf_counter() - t0
        in_bytes = sum(os.path.getsize(f) for f in files)
        st = {}
        t0 = time.perf_counter()
        written = infer_files(files, root, device=dev, stats=st, rng_mode="per_utr", seed=args.base_seed,
                              re_run_mode=False, n_max_apa=args.kcap, n_min_apa=1)
        dt = time.perf_counter() - t0
        assert len(written) == len(files)
        out_bytes = sum(os.path.getsize(f) for f in written)

        def digests():
            import hashlib
            out = {}
            for f in written:
                h = hashlib.sha1()
                with open(f, "rb") as fh:
                    for blk in iter(lambda: fh.read(1 << 22), b""):
                        h.update(blk)
                out[os.path.basename(f)] = h.hexdigest()
            return out
        first = digests()
        # the same job again from pre-binned columnar chunks (scape prebin, scape_amd/binned.py)
        from scape_amd.pipeline import prebin_chunk_file
        t0 = time.perf_counter()
        bfiles = list(pool.ex.map(prebin_chunk_file, files))
        t_prebin = time.perf_counter() - t0
        st2 = {}
        t0 = time.perf_counter()
        infer_files(files, root, device=dev, stats=st2, rng_mode="per_utr", seed=args.base_seed,
                    re_run_mode=False, n_max_apa=args.kcap, n_min_apa=1)
        dt2 = time.perf_counter() - t0
        # full-size property: same seeds -> byte-identical result files, whatever the input form, batching and
        # thread interleaving of the run
        same = digests() == first
        prebinned = dict(value=args.e2e_utrs / dt2, seconds=dt2, prebin_s=t_prebin, outputs_identical_to_pickle_run=same,
                         binned_bytes=sum(os.path.getsize(f) for f in bfiles),
                         stages_s={k: round(v, 3) for k, v in st2.items() if k.endswith("_s")})
        # and in the CLI's default mode: every chunk file keeps the reference's own random stream (results equal
        # the reference's run of that file), the files' current UTRs share the launches (Engine.run_streams)
        t0 = time.perf_counter()
        infer_files(files, root, device=dev, rng_mode="reference", seed=1, re_run_mode=False,
                    n_max_apa=args.kcap, n_min_apa=1)
        dt3 = time.perf_counter() - t0
        reference_streams = dict(value=args.e2e_utrs / dt3, seconds=dt3, chunk_files_in_flight=min(128, len(files)))
        return dict(value=args.e2e_utrs / dt, unit="UTRs/s", utrs=args.e2e_utrs, seconds=dt, prep_workers=pool.workers,
                    from_prebinned_chunks=prebinned, reference_streams=reference_streams,
                    chunk_files=len(files), input_bytes=in_bytes, output_bytes=out_bytes,
                    stages_s={k: round(v, 3) for k, v in st.items() if k.endswith("_s")}, gpu_batches=st.get("n_batch"),
                    includes="read+unpickle chunk files, binning, coverage peaks, restart sampling, H2D, Phase A/B, "
                             "EM sweep, BIC selection, prune re-fits, labels, Parameters, pickle write",
                    synth_write_s=t_write)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    pool = None
    if world == 1 and args.e2e_utrs > 0:
        # the end-to-end leg's prep workers: started before this process touches the GPU, idle until then
        from scape_amd.pipeline import shared_pool
        pool = shared_pool(args.e2e_workers or None)
    from scape_amd import _lib as _sl
    dev = local_rank % max(1, _sl.device_count())      # one rank per GPU (modulo only matters on a 1-GPU test box)
    if world > 1:
        # The data path has no collective (UTR shards are independent), so the process group is only used for
        # the barrier and the max-over-ranks time: gloo on the host.  torch.cuda is deliberately never
        # initialised - torch bundles its own HIP runtime and this library links the system one.
        import torch
        import torch.distributed as dist
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
    eng = Engine(device=dev)
    plan = eng.plan(preps, [(args.base_seed + rank * U + i) % 2 ** 32 for i in range(U)])
    t_prep = time.perf_counter() - t_prep
    t_h2d = time.perf_counter()
    batch = HipBatch(eng.ctx, preps)            # inputs resident in HBM from here on
    t_h2d = time.perf_counter() - t_h2d

    def sync_all():
        # every C-ABI call returns with its stream synchronised, so the device is idle here
        if dist is not None:
            dist.barrier()

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

    # kernel timing (HIP events on the library's stream) of the timed steps
    kern = {}
    for which, name in enumerate(("phase_a", "phase_b", "em_sweep_and_refits", "labels")):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        lib.scape_hip_timing_get(h, which, ctypes.byref(ms), ctypes.byref(n))
        kern[name] = dict(ms_total=ms.value, launches=n.value)
    rounds, slab, zel = eng.last_main_counters
    em_avg_ms = em_ms / args.steps
    # one extra, untimed step with an event pair around every per-round kernel: average duration of the
    # dominant kernel (k2_mstep, one launch per EM round) measured live on the launch stream
    os.environ["SCAPE_HIP_ROUND_TIMING"] = "1"
    lib.scape_hip_timing_reset(h)
    batch.build()
    batch.em_packed(plan["main"])
    del os.environ["SCAPE_HIP_ROUND_TIMING"]
    per = {}
    for which, name in ((4, "k2_estep"), (5, "k2_mstep")):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        lib.scape_hip_timing_get(h, which, ctypes.byref(ms), ctypes.byref(n))
        per[name] = (ms.value, n.value)
        kern[name + "_profiled_step"] = dict(ms_total=ms.value, launches=n.value)
    m_ms, m_n = per["k2_mstep"]
    m_avg_ms = m_ms / max(m_n, 1)
    alg_bytes = 8.0 * slab / max(m_n, 1)                # SURVEY.md 8(d) slab term, per M-step launch
    achieved = alg_bytes / (m_avg_ms * 1e-3) / 1e9
    tensor_bytes = 8.0 * sum(q.T * len(q.betas) * ((q.N + 15) // 16 * 16) for q in preps)

    # the PMC figure was measured on the default workload only
    pmc_traffic = PMC_TRAFFIC_BYTES_PER_LAUNCH if (U, args.reads, args.kcap) == (512, 2000, 10) else None
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
            "roofline": {"bound": "hbm", "kernel": "k2_mstep (tile-stationary f64-MFMA M-step, one launch per EM round)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "hbm_actual_gbs": pmc_traffic / (m_avg_ms * 1e-3) / 1e9 if pmc_traffic else None,
                         "hbm_actual_frac": pmc_traffic / (m_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pmc_traffic else None,
                         "launch_ms": m_avg_ms, "launches_per_sweep": m_n,
                         "tensor_bytes_streamed_once": tensor_bytes,
                         "note": "algorithmic bytes = 8*N*B*|window| summed over the jobs and rounds (what a "
                                 "job-at-a-time M-step reads); the kernel streams each tensor tile once per round "
                                 "for all jobs that need it, so achieved exceeds the HBM peak by the reuse factor; "
                                 "hbm_actual_* = PMC-measured bytes per launch (traffic, default workload) / launch_ms",
                         "em_sweep_ms": em_avg_ms, "em_rounds_per_sweep": int(rounds),
                         "mfma_f64_tflops": 2.0 * slab / (m_ms * 1e-3) / 1e12},
            "kernels_ms": kern,
            "host": {"prep_s": t_prep, "h2d_s": t_h2d, "process_ms": eng.last_host_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = min(16, os.cpu_count() or 1)
            cb, outs = cpu_baseline(preps, plan, args.cpu_seconds, cores)
            out["cpu_baseline"] = cb
            same = parity_count(preps, plan, res, outs)
            out["cpu_baseline"]["parity_sample"] = f"{same}/{len(outs)} sampled UTRs: GPU pA calls identical to the CPU port"
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / cb["value"]
        if pool is not None:
            batch.free()
            out["end_to_end"] = end_to_end(args, pool, dev)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
