"""merge_pa over a synthetic output directory (8,192 UTRs in 64 chunk files, results from this build's infer_pa_all in
per_utr mode): in-process vs spread over the prep pool, from pickles and from pre-binned chunks.
GPU box: python tools/merge_pa_rate.py"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from scape_amd.pipeline import prebin_chunk_file, shared_pool, synth_chunk_file
    pool = shared_pool()
    from scape_amd.apa_core import infer_files
    from scape_amd.junction_handler import _merge_pa
    root = tempfile.mkdtemp(prefix="scape_merge_")
    try:
        os.makedirs(os.path.join(root, "pkl_input"))
        n, per = 8192, 128
        tasks = [(os.path.join(root, "pkl_input", f"synth.{per}.{i}.input.pkl"), i * per, per, 2000, 10, 20250225)
                 for i in range(n // per)]
        files = list(pool.ex.map(synth_chunk_file, tasks))
        infer_files(files, root, rng_mode="per_utr", seed=1, re_run_mode=False, n_max_apa=10, n_min_apa=1)
        for label, workers in (("in-process, pickles", 0), ("pool, pickles", None)):
            t = time.perf_counter()
            m = _merge_pa(root, True, workers=workers)
            dt = time.perf_counter() - t
            print(f"merge_pa {label}: {m} genes in {dt:.1f} s = {m / dt:.0f} genes/s", flush=True)
        list(pool.ex.map(prebin_chunk_file, files))
        for label, workers in (("in-process, pre-binned", 0), ("pool, pre-binned", None)):
            t = time.perf_counter()
            m = _merge_pa(root, True, workers=workers)
            dt = time.perf_counter() - t
            print(f"merge_pa {label}: {m} genes in {dt:.1f} s = {m / dt:.0f} genes/s", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)
