"""End-to-end rate of the exact reference-stream mode over chunk files (the bench's `reference_streams` leg alone),
four times in a row: the leg is noisy (12.6 s and 16.5 s alternate on one box for the same 16,384 UTRs).
GPU box: python tools/streams_e2e.py [n_files]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scape_amd.pipeline import shared_pool, synth_chunk_file   # noqa: E402

if __name__ == "__main__":
    pool = shared_pool(None)
    from scape_amd.apa_core import infer_files                   # noqa: E402
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    root = tempfile.mkdtemp(prefix="scape_streams_")
    try:
        os.makedirs(os.path.join(root, "pkl_input"))
        tasks = [(os.path.join(root, "pkl_input", f"s.128.{i}.input.pkl"), 10 ** 6 + 128 * i, 128, 2000, 10, 20250225)
                 for i in range(n_files)]
        files = list(pool.ex.map(synth_chunk_file, tasks))
        for rep in range(4):
            t0 = time.perf_counter()
            infer_files(files, root, rng_mode="reference", seed=1, re_run_mode=True, n_max_apa=10, n_min_apa=1)
            dt = time.perf_counter() - t0
            print(f"run {rep}: {n_files * 128 / dt:.0f} UTRs/s ({dt:.2f} s)", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)
