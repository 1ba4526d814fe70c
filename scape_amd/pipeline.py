"""Host pipeline around the GPU hot path: keeps one MI355X busy while the CPU side reads chunks.

The reference runs ``infer_pa`` once per chunk file, one UTR after another on one core
(apa_core.py:1104-1137).  With the EM on the GPU the per-UTR host work (unpickling the chunk,
binning, coverage peaks: ~3 ms on one core) is ~8x the GPU time per UTR, so the stages overlap:

    prep workers (processes)  ->  planner (native sampler, host threads)  ->  GPU thread  ->  sink
       read chunk + prepare_utr      Engine.plan                              Engine.process    Parameters + pickle

UTR order, per-UTR seeds (seed + index inside the task) and therefore every result are independent of
the batching, the number of workers and the number of GPUs.  Only the 'per_utr' RNG mode can be
pipelined this way (engine.py: the reference's single stream makes UTR i+1 wait for UTR i).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import queue
import threading
from collections import deque
from concurrent.futures import ProcessPoolExecutor
from time import perf_counter

from . import _hostlib

_STOP = object()


def _worker_init(owner_pid):
    # prep workers never touch the GPU; keep numpy's own thread pools out of the way
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

    def watch():                                      # the workers are children of the fork server, not of the
        import time                                   # process that owns the pool: leave when that one is gone
        while True:
            time.sleep(2.0)
            try:
                os.kill(owner_pid, 0)
            except OSError:
                os._exit(0)
    threading.Thread(target=watch, name="scape-owner-watch", daemon=True).start()
    from . import apa_core, binned  # noqa: F401  (numpy / pandas / scipy imported now, not inside the first task)


def prep_chunk_file(args):
    """Worker task: (path, kwargs) -> prepared UTRs of one prepare_input chunk, in file order."""
    path, kwargs = args
    from .apa_core import load_preps
    return load_preps(path, kwargs)


def prebin_chunk_file(path):
    """Worker task: write <stem>.binned.npz for one chunk (scape_amd/binned.py)."""
    from .binned import prebin_chunk
    return prebin_chunk(path)


def prep_items(args):
    """Worker task: ([(gene_info_str, DataFrame)], kwargs) -> prepared UTRs."""
    items, kwargs = args
    from .host import prepare_utr
    return [prepare_utr(df, gene_info_str=g, **kwargs) for g, df in items]


class PrepPool:
    """Process pool for the per-UTR host preparation.  Start it BEFORE the process initialises the GPU
    (workers come from a fork server that never sees a HIP context)."""

    def __init__(self, workers=None):
        self.workers = workers if workers else max(1, _hostlib.host_threads() - 4)   # planner, two GPU threads, writer
        self.ex = ProcessPoolExecutor(self.workers, mp_context=mp.get_context("forkserver"),
                                      initializer=_worker_init, initargs=(os.getpid(),))
        list(self.ex.map(int, range(self.workers)))            # bring the workers up now

    def close(self):
        self.ex.shutdown(wait=True, cancel_futures=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_shared = None


def shared_pool(workers=None):
    """The process-wide PrepPool (created on first use, closed at interpreter exit).  Call it before the
    first HIP call of the process when you can: the fork server is then born without a GPU context."""
    global _shared
    if _shared is None:
        import atexit
        _shared = PrepPool(workers)
        atexit.register(_shared.close)
    return _shared


def close_shared_pool():
    """Shut the process-wide pool down now.  Needed at the end of multiprocessing children: they leave through
    os._exit() after joining their own child processes, which never end unless the executor is shut down."""
    global _shared
    if _shared is not None:
        _shared.close()
        _shared = None


def synth_chunk_file(args):
    """Worker task for benchmarks/tests: write UTRs [start, start+count) of the synthetic stream
    (scape_amd/synth.py) as one prepare_input-style chunk file; returns the path."""
    path, start, count, reads, k_cap, base_seed = args
    import pickle
    from .synth import synth_utr
    with open(path, "wb") as fh:
        for i in range(start, start + count):
            gene, df, _truth = synth_utr(i, reads, k_cap=k_cap, base_seed=base_seed)
            pickle.dump((gene, df), fh)
    return path


def run_pipeline(tasks, prep_fn, sink, engine_factory, pool, *, seed=1, re_run_mode=True, batch_utrs=512,
                 min_batch_utrs=96, gpu_streams=2, ahead=None, stats=None):
    """tasks: picklable task descriptors; ``prep_fn(task) -> [UtrPrep]`` runs in the pool;
    ``sink(task_index, [UtrResult])`` is called on the caller's thread once per task, in task order.
    UTR j of a task draws from RandomState(seed + j).  Returns the number of UTRs processed.

    gpu_streams engines (one library handle and HIP stream each, same device) take batches in turn, so
    one batch's upload / Phase A/B / host-side selection overlaps the other's EM rounds.  A batch is cut
    at batch_utrs, or already at min_batch_utrs when a GPU thread is waiting for work."""
    import sys
    ahead = ahead or 3 * pool.workers
    q_gpu, q_out = queue.Queue(maxsize=gpu_streams), queue.Queue()
    err = []
    st = stats if stats is not None else {}
    st.update(prep_wait_s=0.0, plan_s=0.0, gpu_s=0.0, gpu_idle_s=0.0, sink_s=0.0, n_utr=0, n_batch=0)
    abort = threading.Event()                        # set when the sink fails: stop feeding / skip queued batches
    waiting = [0]                                    # GPU threads blocked on an empty queue
    sweep_lock = threading.Lock()
    # the GPU threads re-take the GIL after every C-ABI call; with two busy Python threads beside them the
    # default 5 ms switch interval would add up to more than the kernels take
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)

    def producer():
        try:
            futs, nxt = deque(), 0
            cur, meta = [], []                       # preps of the batch being filled, (task, j, seed)
            sizes = {}

            def flush():
                if not cur:
                    return
                t0 = perf_counter()
                from .engine import Engine
                plan = None
                if not any(q.fixed_run for q in cur):
                    plan = Engine.plan(cur, [m[2] for m in meta])
                st["plan_s"] += perf_counter() - t0
                q_gpu.put((list(cur), list(meta), plan))
                cur.clear()
                meta.clear()

            n_task = len(tasks)
            while nxt < n_task and len(futs) < ahead:
                futs.append((nxt, pool.ex.submit(prep_fn, tasks[nxt])))
                nxt += 1
            while futs and not abort.is_set():
                ti, fu = futs.popleft()
                t0 = perf_counter()
                preps = fu.result()
                st["prep_wait_s"] += perf_counter() - t0
                if nxt < n_task and not abort.is_set():
                    futs.append((nxt, pool.ex.submit(prep_fn, tasks[nxt])))
                    nxt += 1
                sizes[ti] = len(preps)
                q_out.put(("size", ti, len(preps)))
                for j, q in enumerate(preps):
                    cur.append(q)
                    meta.append((ti, j, (seed + j) % (2 ** 32)))
                    if len(cur) >= batch_utrs or (len(cur) >= min_batch_utrs and waiting[0] and q_gpu.empty()):
                        flush()
            for _ti, fu in futs:                      # aborted: nothing more is wanted from the workers
                fu.cancel()
            if not abort.is_set():
                flush()
        except BaseException as e:                    # noqa: BLE001 - handed to the caller's thread
            err.append(e)
        finally:
            for _ in range(gpu_streams):
                q_gpu.put(_STOP)

    def gpu_stage():
        try:
            eng = engine_factory()
            eng.sweep_lock = sweep_lock
            while True:
                t0 = perf_counter()
                waiting[0] += 1
                item = q_gpu.get()
                waiting[0] -= 1
                st["gpu_idle_s"] += perf_counter() - t0
                if item is _STOP:
                    break
                preps, meta, plan = item
                if abort.is_set():
                    continue
                t0 = perf_counter()
                if plan is None:
                    res = eng.run(preps, rng_mode="per_utr", seeds=[m[2] for m in meta], re_run_mode=re_run_mode)
                    out = [(r.fit, r.labels_bin, r.n_jobs) for r in res]
                else:
                    waves = eng.waves(preps)
                    out = [None] * len(preps)
                    for wave in waves:
                        wp = [preps[i] for i in wave]
                        wplan = plan if len(waves) == 1 else eng.plan(wp, [meta[i][2] for i in wave])
                        got = eng.process(eng.load(wp), wp, wplan, re_run_mode)
                        for i, g in zip(wave, got):
                            out[i] = g
                st["gpu_s"] += perf_counter() - t0
                st["n_batch"] += 1
                q_out.put(("res", preps, meta, out))
        except BaseException as e:                    # noqa: BLE001
            err.append(e)
            while q_gpu.get() is not _STOP:           # let the producer finish
                pass
        finally:
            q_out.put(_STOP)
            try:
                eng.close()                           # a private library handle is released with its thread
            except Exception:                         # noqa: BLE001 - engine_factory itself may have failed
                pass

    tp = threading.Thread(target=producer, name="scape-plan", daemon=True)
    tgs = [threading.Thread(target=gpu_stage, name=f"scape-gpu{i}", daemon=True) for i in range(gpu_streams)]
    tp.start()
    for tg in tgs:
        tg.start()

    from .engine import UtrResult
    sizes, done, next_task, n_utr = {}, {}, 0, 0
    stops = 0
    while True:
        msg = q_out.get()
        if msg is _STOP:
            stops += 1
            if stops == gpu_streams:
                break
            continue
        if msg[0] == "size":
            sizes[msg[1]] = msg[2]
            done.setdefault(msg[1], [None] * msg[2])
        else:
            _tag, preps, meta, out = msg
            for q, (ti, j, _sd), (fit, lab, nj) in zip(preps, meta, out):
                done[ti][j] = UtrResult(prep=q, fit=fit, labels_bin=lab, n_jobs=nj)
            n_utr += len(preps)
        t0 = perf_counter()
        try:
            while not err and next_task in sizes and all(r is not None for r in done[next_task]):
                sink(next_task, done.pop(next_task))
                next_task += 1
        except BaseException as e:                    # noqa: BLE001 - a failing sink stops the other stages too
            err.append(e)
            abort.set()
        st["sink_s"] += perf_counter() - t0
    tp.join()
    for tg in tgs:
        tg.join()
    sys.setswitchinterval(old_switch)
    if err:
        raise err[0]
    if next_task != len(tasks):
        raise RuntimeError("pipeline ended before every task was written")
    st["n_utr"] = n_utr
    return n_utr
