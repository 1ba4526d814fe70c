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

Launch: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` every rank reads
RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Started plainly with `--gpus N` (N > 1, no WORLD_SIZE)
this process starts the N ranks itself as fresh child processes BEFORE touching the GPU, runs the multi-GPU
end-to-end leg, merges it into rank 0's JSON line and prints that one line; a failed rank is a non-zero exit.

roofline (the dominant kernel, k2_mstep): `traffic` = HBM-side bytes per launch as tallied by the kernel
itself (tensor-tile bytes streamed + each job's v vector once; scape_hip_em_traffic), `achieved` = traffic /
HIP-event launch time, `frac` = achieved / 8 TB/s.  SURVEY.md 8(d)'s job-at-a-time slab bytes are kept beside
it as `algorithmic_bytes_per_launch` with `reuse_factor` = algorithmic / traffic.
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
    ap.add_argument("--utrs", type=int, default=512, help="UTRs per GPU per step")
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--kcap", type=int, default=10)
    ap.add_argument("--base-seed", type=int, default=20250225)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the two-step legs at BASELINE configs #2 / #4 / #5")
    ap.add_argument("--no-cli-leg", action="store_true", help="skip the fresh-process `scape infer_pa` leg")
    ap.add_argument("--e2e-utrs", type=int, default=50000,
                    help="UTRs of the untimed-by-the-metric end-to-end leg (chunk files -> .res.pkl), 0 = skip; "
                         "default = the whole headline job")
    ap.add_argument("--e2e-ref-utrs", type=int, default=16384,
                    help="UTRs of the end-to-end leg in the exact reference-stream mode (a prefix of the same files)")
    ap.add_argument("--e2e-multi-utrs", type=int, default=1024,
                    help="N > 1: UTRs PER GPU of the multi-GPU end-to-end leg (infer_all(gpus=N) on chunk files of "
                         "BASELINE config #4's shape), 0 = skip")
    ap.add_argument("--e2e-multi-reads", type=int, default=10000)
    ap.add_argument("--e2e-multi-repeat", type=int, default=4,
                    help="N > 1: every chunk file of the multi-GPU leg is processed this many times (hard links under other "
                         "names: more work per worker without writing more synthetic input)")
    ap.add_argument("--e2e-workers", type=int, default=0, help="prep processes of the end-to-end leg (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU work for the baseline sample")
    return ap.parse_args()


def cpu_baseline(preps, plan, target_s, cores, max_utrs=None):
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
    if max_utrs is not None:
        n = min(n, int(max_utrs))
    n = max(n, 1)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        outs = list(ex.map(one, range(1, 1 + n)))
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="UTRs/s", cores=cores, kind="port", single_thread_value=1.0 / t1,
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
                     and np.allclose(fit.ws, ws[j, :K + 1], rtol=1e-4, atol=1e-9)
                     and abs(fit.bic - bic[j]) <= 1e-8 * abs(bic[j]))
    return int(same)


def physical_cores():
    """(physical cores among the CPUs this process may run on, threads per core) - /proc/cpuinfo's (physical id,
    core id) pairs; falls back to usable CPUs / 2 when SMT is active and the pairs are not listed."""
    try:
        usable = set(os.sched_getaffinity(0))
    except AttributeError:
        usable = set(range(os.cpu_count() or 1))
    pairs, cur = set(), {}
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if ":" in line:
                    k, v = (t.strip() for t in line.split(":", 1))
                    cur[k] = v
                elif not line.strip():
                    if cur.get("processor", "").isdigit() and int(cur["processor"]) in usable and "core id" in cur:
                        pairs.add((cur.get("physical id", "0"), cur["core id"]))
                    cur = {}
        if cur.get("processor", "").isdigit() and int(cur["processor"]) in usable and "core id" in cur:
            pairs.add((cur.get("physical id", "0"), cur["core id"]))
    except OSError:
        pass
    if pairs:
        return len(pairs), max(1, round(len(usable) / len(pairs)))
    smt = False
    try:
        with open("/sys/devices/system/cpu/smt/active") as fh:
            smt = fh.read().strip() == "1"
    except OSError:
        pass
    return max(1, len(usable) // (2 if smt else 1)), 2 if smt else 1


def cgroup_cpu_limit():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def host_info():
    """What the CPU-side numbers were measured on (printed with cpu_baseline and the end-to-end legs)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    cores, tpc = physical_cores()
    return dict(cpu_model=model, nproc=os.cpu_count() or 1, nproc_usable=usable, physical_cores=cores,
                threads_per_core=tpc, cgroup_cpu_limit=cgroup_cpu_limit())


def write_synth_chunks(pool, root, n_utrs, reads, kcap, base_seed, per_file=128, first_index=10 ** 6):
    os.makedirs(os.path.join(root, "pkl_input"), exist_ok=True)
    from scape_amd.pipeline import synth_chunk_file
    tasks = [(os.path.join(root, "pkl_input", f"synth.{per_file}.{i}.input.pkl"), first_index + i * per_file,
              min(per_file, n_utrs - i * per_file), reads, kcap, base_seed)
             for i in range((n_utrs + per_file - 1) // per_file)]
    t0 = time.perf_counter()
    files = list(pool.ex.map(synth_chunk_file, tasks))
    return files, time.perf_counter() - t0


def end_to_end(args, pool, dev):
    """The whole `infer_pa` job as a user runs it, on the same synthetic stream (default: all 50,000 UTRs of
    the headline job): prepare_input-style chunk files on disk -> non-executing unpickle -> binning /
    coverage peaks (process pool) -> restart tables (native sampler) -> GPU -> Parameters ->
    pkl_output/*.res.pkl.  SURVEY.md 8(d): the rate WITH host pre/post-processing, reported beside `value`
    (which is the resident-batch rate)."""
    import shutil
    import tempfile
    from scape_amd.apa_core import infer_files
    root = tempfile.mkdtemp(prefix="scape_e2e_")
    try:
        files, t_write = write_synth_chunks(pool, root, args.e2e_utrs, args.reads, args.kcap, args.base_seed)
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
        # the reference's run of that file), the files' current UTRs share the launches (Engine.run_streams);
        # re_run_mode on, as the reference's default parameter file has it (input_processor.py:97-112)
        n_ref_files = max(1, min(len(files), (args.e2e_ref_utrs + 127) // 128))
        ref_files = files[:n_ref_files]
        n_ref = min(args.e2e_utrs, n_ref_files * 128)
        t0 = time.perf_counter()
        infer_files(ref_files, root, device=dev, rng_mode="reference", seed=1, re_run_mode=True,
                    n_max_apa=args.kcap, n_min_apa=1)
        dt3 = time.perf_counter() - t0
        reference_streams = dict(value=n_ref / dt3, utrs=n_ref, seconds=dt3, re_run_mode=True,
                                 chunk_files_in_flight=min(256, len(ref_files)),
                                 note="every file its own random stream (the CLI default); files join the running set of "
                                      "streams as their prep finishes (Engine.run_streams_rolling)")
        return dict(value=args.e2e_utrs / dt, unit="UTRs/s", utrs=args.e2e_utrs, seconds=dt, prep_workers=pool.workers,
                    from_prebinned_chunks=prebinned, reference_streams=reference_streams,
                    chunk_files=len(files), input_bytes=in_bytes, output_bytes=out_bytes,
                    stages_s={k: round(v, 3) for k, v in st.items() if k.endswith("_s")}, gpu_batches=st.get("n_batch"),
                    includes="read+unpickle chunk files, binning, coverage peaks, restart sampling, H2D, Phase A/B, "
                             "EM sweep, BIC selection, prune re-fits, labels, Parameters, pickle write",
                    synth_write_s=t_write)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cli_single_chunk(args):
    """The reference's actual usage (input_processor.py:31, tutorial: one `scape infer_pa` process per chunk of 100
    UTRs): a FRESH child process - interpreter start, imports, HIP context, device buffers, unpickling, binning, the
    exact-stream fit, printing and pickling 100 Parameters - timed from outside, with the child's own stage clock
    (SCAPE_TIMING_JSON) beside it.  apa_core.py:107-147."""
    import shutil
    import subprocess
    import tempfile
    from scape_amd.pipeline import synth_chunk_file
    if args.no_cli_leg:
        return None
    root = tempfile.mkdtemp(prefix="scape_cli_")
    try:
        os.makedirs(os.path.join(root, "pkl_input"))
        f = os.path.join(root, "pkl_input", "synth.100.1.1.input.pkl")
        synth_chunk_file((f, 5 * 10 ** 6, 100, args.reads, args.kcap, args.base_seed))
        with open(os.path.join(root, "parameters.toml"), "w") as fh:
            fh.write(f"n_max_apa = {args.kcap}\nn_min_apa = 1\nre_run_mode = true\n")
        runs = []
        for _ in range(2):                          # second run: page cache and code objects warm
            tj = os.path.join(root, "timing.json")
            env = dict(os.environ, SCAPE_TIMING_JSON=tj, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
            t0 = time.perf_counter()
            cp = subprocess.run([sys.executable, "-m", "scape", "infer_pa", "--pkl_input_file", f, "--output_dir", root],
                                env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, cwd=root)
            dt = time.perf_counter() - t0
            if cp.returncode != 0:
                return dict(error=cp.stderr[-2000:])
            with open(tj) as fh:
                stages = json.load(fh)
            runs.append((dt, stages))
        res = os.path.join(root, "pkl_output", "synth.100.1.1.res.pkl")
        dt, stages = min(runs, key=lambda r: r[0])
        inside = sum(v for k, v in stages.items() if k.endswith("_s"))
        return dict(seconds=dt, utrs=100, value=100 / dt, unit="UTRs/s", first_run_seconds=runs[0][0],
                    stages_s={k: round(v, 3) for k, v in stages.items()},
                    interpreter_start_and_exit_s=round(dt - inside, 3), output_bytes=os.path.getsize(res),
                    command="python -m scape infer_pa --pkl_input_file <100 UTRs x %d reads> --output_dir D "
                            "(parameters.toml: n_max_apa = %d, re_run_mode = true; rng_mode = reference, the default)" % (args.reads, args.kcap),
                    note="fresh child process per run, wall clock from outside (best of 2; the first run also pays cold file "
                         "and code-object caches); stages_s is the child's own clock: imports, HIP context + library, "
                         "reading and preparing the chunk, the fit, printing + pickling the results")
    finally:
        shutil.rmtree(root, ignore_errors=True)


def single_chunk_reference(args, dev):
    """`scape infer_pa` as the reference is used - ONE chunk file, its own random stream (np.random.seed(1),
    apa_core.py:125), re_run_mode on: the small-call latency path."""
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    n = 128
    kw = dict(n_max_apa=args.kcap, n_min_apa=1)
    preps = []
    for i in range(n):
        g, df, _ = synth_utr(2 * 10 ** 6 + i, args.reads, k_cap=args.kcap, base_seed=args.base_seed)
        preps.append(prepare_utr(df, gene_info_str=g, **kw))
    eng = Engine(device=dev)
    eng.run(preps, rng_mode="reference", seed=1, re_run_mode=True)              # warm-up (device buffers of this size)
    out = {}
    for name, depth in (("followers", Engine.spec_depth), ("serial", 1)):
        keep, Engine.spec_depth = Engine.spec_depth, depth
        try:
            dt, before = 1e30, dict(Engine.spec_stats)
            for _ in range(2):
                t0 = time.perf_counter()
                res = eng.run(preps, rng_mode="reference", seed=1, re_run_mode=True)
                dt = min(dt, time.perf_counter() - t0)
        finally:
            Engine.spec_depth = keep
        out[name] = dict(rate=n / dt, seconds=dt, sig=[(r.fit.K, r.fit.a_idx.tobytes(), r.fit.ws.tobytes(), r.n_jobs) for r in res],
                         stats={k: (Engine.spec_stats[k] - before[k]) // 2 for k in before})
    f, s1 = out["followers"], out["serial"]
    return dict(value=f["rate"], unit="UTRs/s", utrs=n, seconds=f["seconds"], rng_mode="reference", re_run_mode=True,
                utrs_of_the_stream_per_em_call=Engine.spec_depth, em_calls=f["stats"]["calls"],
                followers_kept=f["stats"]["utrs_kept"] - f["stats"]["calls"], followers_discarded=f["stats"]["utrs_discarded"],
                outcome_predictions_right=f"{f['stats']['predicted_right']}/{f['stats']['predicted']}",
                strictly_serial=dict(value=s1["rate"], seconds=s1["seconds"], em_calls=s1["stats"]["calls"]),
                identical_to_strictly_serial=f["sig"] == s1["sig"],
                note="one chunk file of 128 UTRs in the CLI's default exact-stream mode (prepared UTRs resident on the host; "
                     "best of two runs): up to 8 UTRs of the stream per EM call, the followers drawn from the generator "
                     "state their predecessors' predicted outcome leaves (Engine._drive_streams; the prediction pass is "
                     "inside the timed region); strictly_serial = one UTR per call")


def end_to_end_multi(args, pool, gpus):
    """The product path on N GPUs: `infer_pa_all --gpus N` (scape_amd.apa_core.infer_all) over chunk files of
    BASELINE config #4's shape (10k reads per UTR), one worker process + prep pool per GPU, chunk files sharded
    by size, no collective.  Reports the per-worker pipeline stage seconds so the host-side limit is visible.
    Runs in a process that has not initialised the GPU (or whose fork server was started before it did)."""
    import shutil
    import tempfile
    from scape_amd.apa_core import infer_all
    n_utrs = args.e2e_multi_utrs * gpus
    root = tempfile.mkdtemp(prefix="scape_e2e_multi_")
    try:
        files, t_write = write_synth_chunks(pool, root, n_utrs, args.e2e_multi_reads, args.kcap, args.base_seed,
                                            per_file=64, first_index=3 * 10 ** 6)
        rep = max(1, args.e2e_multi_repeat)
        base_files = list(files)         # the same chunks again under other names: independent files as far as infer_pa_all can tell
        for k in range(1, rep):
            for f in base_files:
                g = os.path.join(os.path.dirname(f), f"rep{k}." + os.path.basename(f))
                os.link(f, g)
                files.append(g)
        n_utrs *= rep
        in_bytes = sum(os.path.getsize(f) for f in files)
        st = {}
        t0 = time.perf_counter()
        infer_all(root, gpus=gpus, stats=st, rng_mode="per_utr", seed=args.base_seed, re_run_mode=False,
                  n_max_apa=args.kcap, n_min_apa=1)
        dt = time.perf_counter() - t0
        outs = [os.path.join(root, "pkl_output", os.path.basename(f)[:-10] + ".res.pkl") for f in files]
        assert all(os.path.exists(o) for o in outs), "a chunk has no result file"
        busy = max((w.get("wall_s", 0.0) for w in st.get("workers") or []), default=0.0)
        return dict(value=(n_utrs / busy) if busy else n_utrs / dt, unit="UTRs/s", n_gpus=gpus, utrs=n_utrs,
                    seconds_incl_worker_startup=dt, value_incl_worker_startup=n_utrs / dt,
                    worker_startup_s=(dt - busy) if busy else None, slowest_worker_s=busy,
                    note="value = UTRs / the slowest worker's own pipeline time; starting the worker processes (python + "
                         "numpy imports, HIP context, first device buffers) is reported beside it",
                    workload=f"{n_utrs} synthetic UTRs x {args.e2e_multi_reads} reads, K=1..{args.kcap} "
                             f"(BASELINE config #4 shape; {len(base_files)} distinct chunk files x {rep}), {len(files)} chunk files "
                             f"-> .res.pkl, rng_mode per_utr",
                    chunk_files=len(files), input_bytes=in_bytes, synth_write_s=t_write,
                    workers=st.get("workers"), prep_workers_per_gpu=st.get("prep_workers_per_gpu"), host=host_info())
    finally:
        shutil.rmtree(root, ignore_errors=True)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process
    never touches the GPU), then run the multi-GPU end-to-end leg here and print ONE JSON line."""
    import socket
    import subprocess
    pool = None
    if args.e2e_multi_utrs > 0:
        from scape_amd.pipeline import shared_pool
        pool = shared_pool(args.e2e_workers or None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SCAPE_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n{out0}\n")
        sys.exit(1)
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n" + out0 + "\n")
        sys.exit(1)
    out = json.loads(line)
    if pool is not None:
        out["end_to_end_multi_gpu"] = end_to_end_multi(args, pool, args.gpus)
    print(json.dumps(out))


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a wrong n_gpus\n")
        sys.exit(2)
    child = os.environ.get("SCAPE_BENCH_CHILD") == "1"       # ranks started by launch_ranks: the parent runs the e2e legs
    dist = None
    pool = None
    want_e2e = (world == 1 and args.e2e_utrs > 0) or (world > 1 and rank == 0 and not child and args.e2e_multi_utrs > 0)
    if want_e2e:
        # the end-to-end legs' prep workers (and the fork server the multi-GPU workers come from): started before
        # this process touches the GPU, idle until then
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
        import datetime
        # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."); the contract is ONE JSON line
        # there, so file descriptor 1 points at stderr until the group is up and has done its first collective
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(hours=1))   # ranks wait while rank 0 runs the e2e leg
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        tdev = torch.device("cpu")

    from scape_amd.engine import Engine, HipBatch
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr

    def sync_all():
        # every C-ABI call returns with its stream synchronised, so the device is idle here
        if dist is not None:
            dist.barrier()

    def timed(fn, steps):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, r

    eng = Engine(device=dev)
    lib, h = eng.ctx.lib, eng.ctx.h

    def measure(U, reads, kcap, steps, warmup, index0, base_seed, rerun_leg=True):
        """One workload shape: U synthetic UTRs x `reads` reads resident in HBM, K = 1..kcap x 10 restarts; `steps` timed
        passes of the hot path, then one extra sweep with an event pair around every per-round kernel (roofline)."""
        kw = dict(n_max_apa=kcap, n_min_apa=1)
        t_prep = time.perf_counter()
        preps = []
        for i in range(U):
            gene, df, _truth = synth_utr(index0 + i, reads, k_cap=kcap, base_seed=base_seed)
            preps.append(prepare_utr(df, gene_info_str=gene, **kw))
        plan = eng.plan(preps, [(base_seed + index0 + i) % 2 ** 32 for i in range(U)])
        t_prep = time.perf_counter() - t_prep
        t_h2d = time.perf_counter()
        batch = HipBatch(eng.ctx, preps)            # inputs resident in HBM from here on
        t_h2d = time.perf_counter() - t_h2d
        em_ms = [0.0]

        def step():
            r = eng.process(batch, preps, plan, re_run_mode=False)
            em_ms[0] += eng.last_main_em_ms
            return r

        for _ in range(warmup):
            res = step()
        lib.scape_hip_timing_reset(h)
        em_ms[0] = 0.0
        elapsed, res = timed(step, steps)

        # kernel timing (HIP events on the library's stream) of the timed steps
        kern = {}
        for which, name in enumerate(("phase_a", "phase_b", "em_sweep_and_refits", "labels")):
            ms, n = ctypes.c_double(), ctypes.c_int32()
            lib.scape_hip_timing_get(h, which, ctypes.byref(ms), ctypes.byref(n))
            kern[name] = dict(ms_total=ms.value, launches=n.value)
        rounds, slab, zel = eng.last_main_counters
        em_avg_ms = em_ms[0] / steps
        host_ms = dict(eng.last_host_ms)

        # the same timed region with re_run_mode on (the reference's default): UTRs whose K hits the cap
        # get a second, larger sweep through the generic state machine
        rr = None
        if rerun_leg:
            rr_steps = max(1, min(steps, 2))
            rr_elapsed, _ = timed(lambda: eng.process(batch, preps, plan, re_run_mode=True), rr_steps)
            rr = {"value": world * U * rr_steps / rr_elapsed, "unit": "UTRs/s", "steps": rr_steps,
                  "ms_per_step": 1e3 * rr_elapsed / rr_steps,
                  "note": "same resident batch and timed region with the re-run rule of subsample_run "
                          "(apa_core.py:1023-1030) on"}

        # one extra, untimed step with an event pair around every per-round kernel: average duration of the
        # dominant kernel (the M-step, one launch per EM round) measured live on the launch stream, and the
        # kernel's own byte tally of that sweep
        os.environ["SCAPE_HIP_ROUND_TIMING"] = "1"
        lib.scape_hip_timing_reset(h)
        batch.build()
        batch.em_packed(plan["main"])
        del os.environ["SCAPE_HIP_ROUND_TIMING"]
        traffic = batch.em_traffic()
        per = {}
        for which, name in ((4, "k2_estep"), (5, "k2_mstep")):
            ms, n = ctypes.c_double(), ctypes.c_int32()
            lib.scape_hip_timing_get(h, which, ctypes.byref(ms), ctypes.byref(n))
            per[name] = (ms.value, n.value)
            kern[name + "_profiled_step"] = dict(ms_total=ms.value, launches=n.value)
        m_ms, m_n = per["k2_mstep"]
        m_avg_ms = m_ms / max(m_n, 1)
        assert traffic["launches"] == m_n, (traffic, m_n)
        alg_bytes = 8.0 * slab / max(m_n, 1)                # SURVEY.md 8(d) slab term (job-at-a-time), per M-step launch
        hbm_bytes = (traffic["tensor_bytes"] + traffic["v_bytes_unique"]) / max(m_n, 1)
        achieved = hbm_bytes / (m_avg_ms * 1e-3) / 1e9
        tensor_bytes = 8.0 * sum(q.T * len(q.betas) * ((q.N + 15) // 16 * 16) for q in preps)
        N = np.array([q.N for q in preps])
        T = np.array([q.T for q in preps])
        roofline = {"bound": "hbm", "kernel": "k3_mstep (tile-stationary f64-MFMA M-step on LDS-DMA rings, one launch per EM round)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": hbm_bytes,
                    "traffic_source": "tallied by the kernel itself (scape_hip_em_traffic): tensor-tile bytes "
                                      "streamed + each job's v vector once, per launch; the rocprofv3 FETCH_SIZE "
                                      "pass of the same command is in profiles/ (README: how they compare)",
                    "traffic_tensor_bytes": traffic["tensor_bytes"] / max(m_n, 1),
                    "traffic_v_bytes_unique": traffic["v_bytes_unique"] / max(m_n, 1),
                    "v_bytes_requested_incl_l2_hits": traffic["v_bytes_requested"] / max(m_n, 1),
                    "launch_ms": m_avg_ms, "launches_per_sweep": m_n,
                    "algorithmic_bytes_per_launch": alg_bytes, "reuse_factor": alg_bytes / max(hbm_bytes, 1.0),
                    "tensor_bytes_streamed_once": tensor_bytes,
                    "note": "achieved = traffic / launch_ms (HIP events on the launch stream).  "
                            "algorithmic_bytes_per_launch = SURVEY.md 8(d)'s 8*N*B*|window| summed over jobs "
                            "(what a job-at-a-time M-step reads); the kernel streams each live tensor tile once "
                            "per round for all jobs that need it: reuse_factor = algorithmic / traffic",
                    "em_sweep_ms": em_avg_ms, "em_rounds_per_sweep": int(rounds),
                    "mfma_f64_tflops": 2.0 * slab / (m_ms * 1e-3) / 1e12}
        return dict(preps=preps, plan=plan, res=res, batch=batch, elapsed=elapsed, kern=kern, host_ms=host_ms, rr=rr,
                    roofline=roofline, t_prep=t_prep, t_h2d=t_h2d, n_frag_mean=float(N.mean()), n_frag_max=int(N.max()),
                    n_theta_mean=float(T.mean()), value=world * U * steps / elapsed, ms_per_step=1e3 * elapsed / steps)

    U = args.utrs
    m = measure(U, args.reads, args.kcap, args.steps, args.warmup, rank * U, args.base_seed)
    preps, plan, res, batch = m["preps"], m["plan"], m["res"], m["batch"]

    if rank == 0:
        out = {
            "metric": "UTR regions/sec (infer_pa EM to convergence), 50k UTRs x 2k reads synthetic",
            "value": m["value"], "unit": "UTRs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"50k-UTR synthetic stream x {args.reads} reads, K=1..{args.kcap} x 10 restarts; "
                                   f"step = one resident batch of {U} UTRs per GPU",
                       "utrs_per_gpu_per_step": U, "reads_per_utr": args.reads, "k_cap": args.kcap,
                       "em_jobs_per_step": int(len(plan["main"])), "n_frag_mean": m["n_frag_mean"],
                       "n_frag_max": m["n_frag_max"], "n_theta_mean": m["n_theta_mean"], "re_run_mode": False,
                       "rng_mode": "per_utr", "parallelism": f"utr-shard x{world}"},
            "roofline": m["roofline"],
            "kernels_ms": m["kern"],
            "re_run_mode_true": m["rr"],
            "host": {"prep_s": m["t_prep"], "h2d_s": m["t_h2d"], "process_ms": m["host_ms"], **host_info()},
        }
        if world == 1 and not args.no_cpu_baseline:
            hi = host_info()
            cores = hi["physical_cores"]
            if hi["cgroup_cpu_limit"]:
                cores = max(1, min(cores, int(hi["cgroup_cpu_limit"] + 0.5)))
            cb, outs = cpu_baseline(preps, plan, args.cpu_seconds, cores)
            out["cpu_baseline"] = cb
            same = parity_count(preps, plan, res, outs)
            out["cpu_baseline"]["parity_sample"] = f"{same}/{len(outs)} sampled UTRs: GPU pA calls identical to the CPU port"
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / cb["value"]
            out["cpu_baseline"]["gpu_over_cpu_single_thread"] = out["value"] / cb["single_thread_value"]
            out["cpu_baseline"].update(hi)
    batch.free()
    if rank == 0 and world == 1 and not args.no_other_configs:
        # The other single-GPU BASELINE shapes, two steps each through the same timed region: config #2 (1k x 500 reads,
        # K <= 5), and - one GPU's share of - config #4 (10k reads, K <= 10) and config #5 (5k reads, K = 1..12)
        oc = []
        for name, (U2, reads2, kcap2) in (("config #2: 500 reads, K<=5", (2048, 500, 5)),
                                          ("config #4 shape: 10,000 reads, K<=10", (256, 10000, 10)),
                                          ("config #5 shape: 5,000 reads, K=1..12", (256, 5000, 12))):
            m2 = measure(U2, reads2, kcap2, 2, 1, 4 * 10 ** 6, args.base_seed, rerun_leg=False)
            e = {"workload": f"{name}; step = one resident batch of {U2} UTRs", "value": m2["value"], "unit": "UTRs/s",
                 "steps": 2, "ms_per_step": m2["ms_per_step"], "n_frag_mean": m2["n_frag_mean"],
                 "em_jobs_per_step": int(len(m2["plan"]["main"])),
                 "roofline": {k: m2["roofline"][k] for k in ("achieved", "frac", "traffic", "launch_ms", "launches_per_sweep",
                                                            "em_sweep_ms", "mfma_f64_tflops", "reuse_factor")}}
            if not args.no_cpu_baseline:
                hi = host_info()
                cores = hi["physical_cores"]
                if hi["cgroup_cpu_limit"]:
                    cores = max(1, min(cores, int(hi["cgroup_cpu_limit"] + 0.5)))
                cb2, outs2 = cpu_baseline(m2["preps"], m2["plan"], args.cpu_seconds / 4, cores, max_utrs=cores)
                same2 = parity_count(m2["preps"], m2["plan"], m2["res"], outs2)
                e["parity_sample"] = f"{same2}/{len(outs2)} sampled UTRs: GPU pA calls identical to the CPU port"
                e["cpu_port"] = {k: cb2[k] for k in ("value", "cores", "single_thread_value", "sample")}
            m2["batch"].free()
            oc.append(e)
        out["other_configs"] = oc
    if rank == 0:
        if world == 1 and pool is not None:
            out["single_chunk_reference"] = single_chunk_reference(args, dev)
            out["cli_single_chunk"] = cli_single_chunk(args)
            out["end_to_end"] = end_to_end(args, pool, dev)
    if dist is not None:
        dist.barrier()                           # every rank has released its batch
        if rank == 0 and pool is not None:
            out["end_to_end_multi_gpu"] = end_to_end_multi(args, pool, world)
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
