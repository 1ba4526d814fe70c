"""cProfile of the exact-stream chunk-directory run (infer_files, rng_mode reference) on the main thread.
GPU box: python tools/streams_e2e_profile.py [n_files]"""
import cProfile
import os
import pstats
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
        infer_files(files[:16], root, rng_mode="reference", seed=1, re_run_mode=True, n_max_apa=10, n_min_apa=1)
        st = {}
        t0 = time.perf_counter()
        pr = cProfile.Profile()
        pr.enable()
        infer_files(files, root, rng_mode="reference", seed=1, re_run_mode=True, n_max_apa=10, n_min_apa=1, stats=st)
        pr.disable()
        dt = time.perf_counter() - t0
        print(f"{n_files * 128 / dt:.0f} UTRs/s ({dt:.2f} s)  stages {st}", flush=True)
        pstats.Stats(pr).sort_stats('tottime').print_stats(22)
        pstats.Stats(pr).sort_stats('cumtime').print_stats(30)
    finally:
        shutil.rmtree(root, ignore_errors=True)
