"""CPU tests of the host pipeline (scape_amd/pipeline.py) with a stand-in for the GPU stage: task order,
per-UTR seeds, batches that straddle files, empty chunks, error propagation.  The real GPU stage is
covered by tests/test_gpu_parity.py::test_pipelined_per_utr_mode_equals_engine_run."""
import os
import pickle

import numpy as np
import pytest


class FakeEngine:
    """Engine look-alike: 'fits' are digests of the plan rows the UTR was given."""
    ctx = None

    def __init__(self):
        from scape_amd.engine import Engine
        self.plan = Engine.plan

    def waves(self, preps):
        half = max(1, len(preps) // 2)                      # exercise the multi-wave branch too
        idx = list(range(len(preps)))
        return [idx[:half], idx[half:]] if len(preps) > 3 else [idx]

    def load(self, wp):
        return None

    def process(self, batch, preps, plan, re_run):
        pj, sp = plan["main"], plan["spans"]
        return [((q.gene_info_str, int(pj.a[sp[u]:sp[u + 1]].sum()), float(pj.w[sp[u], 0])),
                 np.zeros(q.N, np.int32), int(sp[u + 1] - sp[u])) for u, q in enumerate(preps)]


def _write_chunks(tmp_path, sizes):
    from scape_amd.synth import synth_chunk
    files = []
    for i, n in enumerate(sizes):
        f = os.path.join(tmp_path, f"s.{i}.input.pkl")
        with open(f, "wb") as fh:
            for it in synth_chunk(n, 250, base_seed=100 * i, k_cap=3):
                pickle.dump(it, fh)
        files.append(f)
    return files


@pytest.fixture(scope="module")
def pool():
    from scape_amd.pipeline import PrepPool
    with PrepPool(2) as p:
        yield p


def test_pipeline_order_seeds_and_batching(tmp_path, pool):
    from scape_amd import pipeline
    from scape_amd.engine import Engine
    files = _write_chunks(str(tmp_path), (5, 0, 7, 3, 6))
    kw = dict(n_max_apa=3, n_min_apa=1)
    order, got, stats = [], {}, {}

    def sink(ti, res):
        order.append(ti)
        got[ti] = res
    n = pipeline.run_pipeline([(f, kw) for f in files], pipeline.prep_chunk_file, sink, FakeEngine, pool,
                              seed=2 ** 32 - 2, batch_utrs=4, stats=stats)
    assert n == 21 and stats["n_utr"] == 21 and order == [0, 1, 2, 3, 4]
    assert [len(got[i]) for i in range(5)] == [5, 0, 7, 3, 6]
    for ti, f in enumerate(files):
        preps = pipeline.prep_chunk_file((f, kw))
        if not preps:
            continue
        plan = Engine.plan(preps, [(2 ** 32 - 2 + j) % 2 ** 32 for j in range(len(preps))])
        want = FakeEngine().process(None, preps, plan, True)
        assert [r.fit for r in got[ti]] == [w[0] for w in want]
        assert [r.prep.gene_info_str for r in got[ti]] == [q.gene_info_str for q in preps]


def test_pipeline_propagates_worker_and_gpu_errors(tmp_path, pool):
    from scape_amd import pipeline
    files = _write_chunks(str(tmp_path), (3, 3))
    kw = dict(n_max_apa=3, n_min_apa=1)
    with pytest.raises(FileNotFoundError):
        pipeline.run_pipeline([(files[0], kw), (os.path.join(str(tmp_path), "missing.input.pkl"), kw)],
                              pipeline.prep_chunk_file, lambda ti, res: None, FakeEngine, pool, batch_utrs=2)

    def bad_sink(ti, res):
        raise OSError("disk full")
    with pytest.raises(OSError, match="disk full"):
        pipeline.run_pipeline([(f, kw) for f in files], pipeline.prep_chunk_file, bad_sink, FakeEngine, pool, batch_utrs=2)

    class Boom(FakeEngine):
        def process(self, *a):
            raise RuntimeError("device lost")
    with pytest.raises(RuntimeError, match="device lost"):
        pipeline.run_pipeline([(f, kw) for f in files], pipeline.prep_chunk_file, lambda ti, res: None, Boom, pool,
                              batch_utrs=2)


def _child_uses_and_closes_pool():
    from scape_amd import pipeline
    pool = pipeline.shared_pool(2)
    assert list(pool.ex.map(int, [1, 2, 3])) == [1, 2, 3]
    pipeline.close_shared_pool()


def test_worker_process_exits_after_closing_its_pool():
    """A multiprocessing child joins its own children before os._exit(): without close_shared_pool() the prep
    workers keep it alive for ever (what `infer_pa_all --gpus N` workers do in their finally clause)."""
    import multiprocessing as mp
    p = mp.get_context("forkserver").Process(target=_child_uses_and_closes_pool)
    p.start()
    p.join(60)
    alive = p.is_alive()
    if alive:
        p.kill()
    assert not alive and p.exitcode == 0
