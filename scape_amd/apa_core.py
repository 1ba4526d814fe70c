"""``scape infer_pa`` on MI355X - host mirror of the reference's ``scape/apa_core.py`` surface.

Same command line, same ``parameters.toml`` handling, same input chunk format and the
same output stream of ``scape.apa_core.Parameters`` pickles as the reference
(``/root/reference/src/scape/apa_core.py:40-147``, ``:236-258``, ``:984-1035``,
``:1104-1137``); the per-UTR inference itself runs through ``scape_amd.engine`` on the
GPU.  There is no CPU fallback.
"""
from __future__ import annotations

import os
import pickle
from pathlib import Path
from timeit import default_timer as timer

import click
import numpy as np

from . import safe_pickle
from .engine import Engine
from .host import prepare_utr

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover - 3.10 images
    import tomli as _toml


class Parameters:
    """Result record, field-compatible with the reference class (apa_core.py:236-258);
    pickles as ``scape.apa_core.Parameters`` so merge_pa & co. load it unchanged."""

    def __init__(self, title='', alpha_arr=None, beta_arr=None, ws=None, L=None, cb_id_arr=None,
                 readID_arr=None, K=None):
        self.title = title
        self.alpha_arr = alpha_arr
        self.beta_arr = beta_arr
        self.ws = ws
        self.K = len(self.alpha_arr)
        self.L = L
        self.cb_id_arr = cb_id_arr
        self.readID_arr = readID_arr

    def __str__(self):
        out = '-' * 10 + f'{self.title} K={self.K}' + '-' * 10 + '\n'
        if hasattr(self, 'gene_info_str'):
            out += f'gene info: {self.gene_info_str}\n'
        out += f'K={self.K} L={self.L} Last component is uniform component.\n'
        out += f'alpha_arr={self.alpha_arr}\n'
        out += f'beta_arr={self.beta_arr}\n'
        out += f'ws={np.around(self.ws, decimals=2)}\n'
        if hasattr(self, 'bic'):
            out += f'bic={np.around(self.bic, decimals=2)}\n'
        out += '-' * 30 + '\n'
        return out


Parameters.__module__ = "scape.apa_core"
Parameters.__qualname__ = "Parameters"


def to_parameters(res):
    """UtrResult -> Parameters with the reference's field set and dtypes."""
    q, f = res.prep, res.fit
    para = Parameters(title='Final Result (subsample run)' if q.fixed_run else 'Final Result',
                      alpha_arr=np.rint(q.theta[f.a_idx]).astype('int'),
                      beta_arr=q.betas[f.b_idx].astype(np.float64),
                      ws=np.array(f.ws, dtype=np.float64), L=int(q.L),
                      cb_id_arr=q.cb_id, readID_arr=q.read_id)
    para.bic = np.float64(f.bic)
    para.lb_arr = [np.float64(v) for v in f.lb]
    para.label_arr = res.labels_bin.astype(np.int64)[q.idx]
    para.gene_info_str = q.gene_info_str
    return para


def _dump_toml(d, fh):
    """flat key = value writer (the reference uses tomli_w.dump, apa_core.py:98-99)."""
    lines = []
    for k, v in d.items():
        if v is None:
            continue
        if isinstance(v, bool):
            s = "true" if v else "false"
        elif isinstance(v, (int, float)):
            s = repr(v)
        else:
            s = '"' + str(v).replace("\\", "\\\\").replace('"', '\\"') + '"'
        lines.append(f"{k} = {s}\n")
    fh.write("".join(lines).encode())


def read_input_chunk(path, frames="pandas"):
    """Stream of (gene_info_str, DataFrame) tuples (input_processor.py:223-259).  Uses the
    non-executing reader; SCAPE_TRUST_PICKLE=1 switches to pickle.load like the reference.
    frames="columns": the frames come back as plain column mappings (safe_pickle.Columns), no pandas import."""
    if os.environ.get("SCAPE_TRUST_PICKLE") == "1":
        with open(path, "rb") as fh:
            while True:
                try:
                    yield pickle.load(fh)
                except EOFError:
                    return
    else:
        yield from safe_pickle.iter_pickles(path, frames=frames)


def load_preps(pkl_input_file, kwargs):
    """Prepared UTRs of one chunk, in file order: from <stem>.binned.npz when it is present and was made
    from this pickle (scape_amd/binned.py), else by decoding and binning the pickle stream."""
    from .binned import binned_path, is_current, read_binned
    from .host import prepare_binned
    pre_para = None
    if kwargs.get("fixed_run_mode", False):                  # subsample_run (:999-1007)
        assert kwargs["pre_para_pkl_file"]
        assert os.path.exists(kwargs["pre_para_pkl_file"])
        pre_para = next(iter(read_input_chunk(kwargs["pre_para_pkl_file"])))   # first Parameters only (:1002-1003)
    bp = binned_path(pkl_input_file)
    if is_current(bp, pkl_input_file):
        return [prepare_binned(b, gene_info_str=g, pre_para=pre_para, **kwargs) for g, b, _j in read_binned(bp)]
    return [prepare_utr(df, gene_info_str=gene, pre_para=pre_para, **kwargs)
            for gene, df in read_input_chunk(pkl_input_file, frames="columns")]


def infer(pickle_input_file, pickle_output_file, **kwargs):
    """Reference ``infer`` (apa_core.py:1104-1137): one Parameters per input tuple, same order.

    The HIP context and library handle come up on a side thread (the runtime calls release the GIL) while this thread
    reads and prepares the chunk.  SCAPE_TIMING_JSON=<path>: the stage seconds of this call are written there
    (bench.py's `cli_single_chunk` leg)."""
    import threading
    print(f"start inferring APA events from input pickle file = {pickle_input_file}. "
          f"Output file = {pickle_output_file}")
    start_t = timer()
    box = {}

    def bring_up():
        t0 = timer()
        try:
            box["engine"] = Engine(device=kwargs.get("device"))
        except BaseException as e:                # re-raised on the calling thread
            box["error"] = e
        box["context_s"] = timer() - t0
    th = threading.Thread(target=bring_up, name="scape-hip-context")
    th.start()
    t0 = timer()
    try:
        preps = load_preps(pickle_input_file, kwargs)
    finally:
        th.join()
    t_prep = timer() - t0
    if "error" in box:
        raise box["error"]
    engine = box["engine"]
    t0 = timer()
    results = engine.run(preps, rng_mode=kwargs.get("rng_mode", "reference"), seed=int(kwargs.get("seed", 1)),
                         re_run_mode=bool(kwargs.get("re_run_mode", True)))
    t_fit = timer() - t0
    t0 = timer()
    res_lst = [to_parameters(r) for r in results]
    for res in res_lst:
        print(res)
    print(f"Done {len(res_lst)} UTRs in {(timer() - start_t) / 60} min.")
    with open(pickle_output_file, 'wb') as fh:
        for res in res_lst:
            print(f"save result of {res.gene_info_str}")
            pickle.dump(res, fh)
    t_out = timer() - t0
    tj = os.environ.get("SCAPE_TIMING_JSON")
    if tj:
        import json
        import sys
        t_imp = getattr(sys.modules.get("scape"), "_IMPORT_S", None)
        with open(tj, "w") as fh:
            json.dump({"imports_s": t_imp or 0.0, "read_prepare_s": t_prep, "hip_context_s_overlapped_with_read_prepare": box["context_s"],
                       "fit_s": t_fit, "print_pickle_s": t_out, "utrs": len(res_lst)}, fh)
    return res_lst


def _infer_pa(pkl_input_file: str, output_dir: str, **kwargs):
    """Reference ``_infer_pa`` (apa_core.py:107-147)."""
    if not os.path.exists(pkl_input_file):
        raise Exception("Given input file does not exists")
    os.makedirs(os.path.join(output_dir, "pkl_output"), exist_ok=True)
    filename = os.path.basename(pkl_input_file)[:-10]          # strip ".input.pkl"
    if ".tmp." in filename:
        raise Exception("The input file " + filename + " is incomplete. Please re-run prepare_input() on " +
                        filename.split(".")[0] + ".bam")
    out_pkl_file = os.path.join(output_dir, "pkl_output", filename + ".res.pkl")
    if os.path.exists(out_pkl_file):
        os.remove(out_pkl_file)
    # watch_dog_flag (apa_core.py:140-145) is accepted and ignored: it only logs host CPU/memory
    return infer(pkl_input_file, out_pkl_file, **kwargs)


def _write_results(output_dir, pkl_input_file, results):
    out = os.path.join(output_dir, "pkl_output", os.path.basename(pkl_input_file)[:-10] + ".res.pkl")
    tmp = out + ".part"                                  # complete files only ever appear under the final name
    with open(tmp, "wb") as fh:
        for r in results:
            pickle.dump(to_parameters(r), fh)
    os.replace(tmp, out)
    return out


def infer_files(files, output_dir, device=None, files_in_flight=256, workers=None, stats=None, **kwargs):
    """Several chunk files on one GPU; chunk reading / binning runs in a process pool beside the GPU.

    rng_mode 'reference' (default): each file keeps the reference's own random stream (np.random.seed(1)
    per file, apa_core.py:125) and UTR order, so every <stem>.res.pkl equals what `scape infer_pa` writes
    for that file; the files only share the GPU launches (Engine.run_streams).
    rng_mode 'per_utr': UTR j of a file draws from RandomState(seed + j); files stream through
    scape_amd.pipeline (prep workers -> native planner -> GPU -> writer), the throughput mode."""
    from .engine import Engine
    from .pipeline import prep_chunk_file, run_pipeline, shared_pool
    os.makedirs(os.path.join(output_dir, "pkl_output"), exist_ok=True)
    seed, re_run = int(kwargs.get("seed", 1)), bool(kwargs.get("re_run_mode", True))
    tasks = [(f, kwargs) for f in files]
    written = []
    pool = shared_pool(workers)                        # before this function initialises the GPU
    if kwargs.get("rng_mode", "reference") == "per_utr":
        run_pipeline(tasks, prep_chunk_file, lambda ti, res: written.append(_write_results(output_dir, files[ti], res)),
                     lambda: Engine(device=device, mem_fraction=0.35, own_context=True), pool, seed=seed, re_run_mode=re_run,
                     stats=stats)             # two engines share the device (pipeline.run_pipeline)
        return written
    engine = Engine(device=device)
    # every file is a stream of its own; files join the running set of streams as their prep finishes (at wave
    # boundaries, Engine.run_streams_rolling), at most files_in_flight at a time; results are written by a side thread
    # as files finish.  Prep tasks are handed to the pool a bounded number ahead of the GPU.
    from concurrent.futures import FIRST_COMPLETED, ThreadPoolExecutor, wait
    ahead = files_in_flight + 2 * max(1, getattr(pool, "workers", 8))
    state = dict(next=0, futs={})                         # task index -> future, submitted and not yet admitted

    def top_up():
        while state["next"] < len(tasks) and len(state["futs"]) < ahead:
            state["futs"][state["next"]] = pool.ex.submit(prep_chunk_file, tasks[state["next"]])
            state["next"] += 1

    def more():
        return bool(state["futs"]) or state["next"] < len(tasks)

    def take(block, room):
        top_up()
        if block and state["futs"] and not any(f.done() for f in state["futs"].values()):
            wait(list(state["futs"].values()), return_when=FIRST_COMPLETED)
        got = []
        for ti in sorted(state["futs"]):
            if len(got) >= room:
                break
            if state["futs"][ti].done():
                got.append((ti, state["futs"].pop(ti).result(), seed))
        top_up()
        return got

    with ThreadPoolExecutor(1) as writer:
        writes = {}
        engine.run_streams_rolling(take, more, lambda ti, res: writes.__setitem__(
            ti, writer.submit(_write_results, output_dir, files[ti], res)), re_run_mode=re_run, streams_in_flight=files_in_flight)
        written = [writes[ti].result() for ti in sorted(writes)]
    return written


def _infer_files_worker(rank, files, output_dir, kwargs, workers, stats_file=None):
    os.environ["LOCAL_RANK"] = str(rank)
    from . import _lib
    from .pipeline import close_shared_pool, shared_pool
    shared_pool(workers)                                   # prep workers first: they never see a HIP context
    try:
        device = rank % max(1, _lib.device_count())        # more workers than GPUs only happens in tests
        st = {}
        t0 = timer()
        infer_files(files, output_dir, device=device, workers=workers, stats=st, **kwargs)
        if stats_file:
            import json
            st.update(rank=rank, device=device, chunk_files=len(files), wall_s=timer() - t0)
            with open(stats_file, "w") as fh:
                json.dump({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}, fh)
    finally:
        close_shared_pool()                                # or this process never gets past its exit join


def infer_all(output_dir, gpus=1, resume=False, stats=None, **kwargs):
    """Every complete chunk of <output_dir>/pkl_input on `gpus` GPUs of this node: chunk files are sharded
    over one worker process per GPU (largest-first by file size), no communication between workers
    (merge_pa expects one .res.pkl per .input.pkl, reference junction_handler.py:59-64).
    resume: leave out the chunks whose .res.pkl already exists and is not older than the chunk (result files
    are written under a temporary name and renamed, so an existing one is complete).
    stats: optional dict; receives `workers` = the pipeline stage seconds of every worker process and
    `prep_workers_per_gpu`."""
    import glob
    import multiprocessing as mp

    from .dist import lpt_partition
    files = sorted(f for f in glob.glob(os.path.join(output_dir, "pkl_input", "*.input.pkl"))
                   if ".tmp." not in os.path.basename(f))
    if not files:
        raise Exception("no *.input.pkl under " + os.path.join(output_dir, "pkl_input"))
    if resume:
        def done(f):
            out = os.path.join(output_dir, "pkl_output", os.path.basename(f)[:-10] + ".res.pkl")
            return os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(f)
        todo = [f for f in files if not done(f)]
        print(f"resume: {len(files) - len(todo)} of {len(files)} chunks already have results")
        files = todo
        if not files:
            return []
    shards = lpt_partition([os.path.getsize(f) for f in files], gpus)
    if gpus == 1:
        st = {}
        infer_files(files, output_dir, device=kwargs.pop("device", None), stats=st, **kwargs)
        if stats is not None:
            stats.update(workers=[dict(st, rank=0, chunk_files=len(files))], prep_workers_per_gpu=None)
        return files
    # fork-server children: no exec from this process (which may already hold a HIP context, e.g. under pytest),
    # and the workers are born without one
    ctx = mp.get_context("forkserver")
    try:
        n_cpu = len(os.sched_getaffinity(0))                      # the cores this process may use, not the host's
    except AttributeError:
        n_cpu = os.cpu_count() or 1
    workers = max(1, min(14, n_cpu // gpus - 2))               # prep processes per GPU worker
    import json
    import shutil
    import tempfile
    sdir = tempfile.mkdtemp(prefix="scape_stats_") if stats is not None else None
    try:
        procs = [ctx.Process(target=_infer_files_worker,
                             args=(r, [files[i] for i in shards[r]], output_dir, kwargs, workers,
                                   os.path.join(sdir, f"rank{r}.json") if sdir else None))
                 for r in range(gpus) if shards[r]]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
        bad = [p.exitcode for p in procs if p.exitcode != 0]
        if bad:
            raise Exception(f"infer_pa_all: {len(bad)} worker(s) failed")
        if stats is not None:
            per = []
            for r in range(gpus):
                f = os.path.join(sdir, f"rank{r}.json")
                if os.path.exists(f):
                    with open(f) as fh:
                        per.append(json.load(fh))
            stats.update(workers=per, prep_workers_per_gpu=workers)
    finally:
        if sdir:
            shutil.rmtree(sdir, ignore_errors=True)
    return files


@click.command(name="infer_pa_all")
@click.option('--output_dir', type=str, help='output directory of prepare_input (holds pkl_input/ and parameters.toml)',
              required=True)
@click.option('--toml_para_file', type=str, default=None, required=False)
@click.option('--gpus', type=int, default=1, help='number of GPUs of this node to shard the chunk files over')
@click.option('--resume', is_flag=True, default=False,
              help='skip the chunks whose .res.pkl already exists and is not older than the chunk')
def infer_pa_all(output_dir: str, toml_para_file: str = None, gpus: int = 1, resume: bool = False):
    """infer_pa for every chunk under <output_dir>/pkl_input (not in the reference CLI: its tutorial
    loops `scape infer_pa` over the files in a shell, SCAPE-example-with-DE.ipynb cell 8)."""
    assert Path(output_dir).exists()
    para_dict = {"n_max_apa": 5}
    if toml_para_file is None:
        toml_para_file = Path(output_dir) / "parameters.toml"
    assert os.path.exists(toml_para_file)
    with open(toml_para_file, "rb") as fh:
        para_dict.update(_toml.load(fh))
    para_dict.pop("output_dir", None)
    infer_all(output_dir, gpus=gpus, resume=resume, **para_dict)


@click.command(name="prebin")
@click.option('--output_dir', type=str, required=True,
              help='output directory of prepare_input (holds pkl_input/)')
@click.option('--workers', type=int, default=0, help='processes (0 = one per host core, at most 14)')
def prebin(output_dir: str, workers: int = 0):
    """Write <stem>.binned.npz beside every complete chunk of <output_dir>/pkl_input: the binned, columnar form
    infer_pa / infer_pa_all / merge_pa read instead of the pickles when present (scape_amd/binned.py)."""
    import glob

    from .pipeline import prebin_chunk_file, shared_pool
    files = sorted(f for f in glob.glob(os.path.join(output_dir, "pkl_input", "*.input.pkl"))
                   if ".tmp." not in os.path.basename(f))
    if not files:
        raise Exception("no *.input.pkl under " + os.path.join(output_dir, "pkl_input"))
    for path in shared_pool(workers or None).ex.map(prebin_chunk_file, files):
        print("wrote", path)


@click.command(name="infer_pa")
@click.option('--pkl_input_file', type=str, help='input pickle file (result of prepare_input)', required=True)
@click.option('--output_dir', type=str, help='output directory', required=True)
@click.option('--toml_para_file', type=str, help='a TOML file specifies user-defined parameters', default=None,
              required=False)
@click.option('--pre_para_pkl_file', type=str,
              help='a pickle file with pre-specified pA sites and utr length, result file of scape analysis',
              default=None, required=False)
def infer_pa(pkl_input_file: str, output_dir: str, toml_para_file: str = None, pre_para_pkl_file=None):
    """Infer pA sites for every UTR of one prepare_input chunk (reference apa_core.py:40-104)."""
    assert Path(output_dir).exists()
    para_dict = {"n_max_apa": 5, "pre_para_pkl_file": pre_para_pkl_file}
    if toml_para_file is None:
        toml_para_file = Path(output_dir) / "parameters.toml"
    if toml_para_file:
        assert os.path.exists(toml_para_file)
        with open(toml_para_file, "rb") as fh:
            user_para_dict = _toml.load(fh)
            para_dict.update(user_para_dict)
        print(f"Parameter file {toml_para_file} loaded.")
        for k, v in user_para_dict.items():
            print(f"{k} = {v}")
        print()
    if pre_para_pkl_file:
        assert os.path.exists(pre_para_pkl_file)
        para_dict["fixed_run_mode"] = True
        para_dict["pre_para_pkl_file"] = pre_para_pkl_file
        with open(toml_para_file, 'wb') as fh:
            _dump_toml(para_dict, fh)
    if "output_dir" in para_dict:
        del para_dict["output_dir"]
    _infer_pa(pkl_input_file, output_dir, **para_dict)
