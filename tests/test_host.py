"""CPU tests of the product's host logic (no GPU compute): chunk reader, binning, coverage,
restart sampling stream, CLI error behaviour, and that libscape_hip.so loads and exports
every symbol include/scape_hip.h declares."""
import os
import pickle
import re

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT, load_npz, trace_params, utr_df


# ---------------------------------------------------------------- C-ABI surface
def test_library_exports_every_declared_symbol():
    from scape_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "scape_hip.h")).read()
    declared = set(re.findall(r"\b(scape_hip_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.load_library()
    for name in declared:
        assert hasattr(lib, name), f"libscape_hip.so lacks {name}"
    assert declared == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.scape_hip_abi_version() == 4


def test_host_library_exports_every_declared_symbol():
    from scape_amd import _hostlib
    hdr = open(os.path.join(ROOT, "include", "scape_host.h")).read()
    declared = set(re.findall(r"\b(scape_host_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _hostlib.load_library()
    for name in declared:
        assert hasattr(lib, name), f"libscape_host.so lacks {name}"
    assert declared == set(_hostlib.SIGNATURES), "ctypes table and header disagree"


def test_no_product_import_of_oracle():
    """The product path must never route through the oracle."""
    for root, _dirs, files in os.walk(os.path.join(ROOT, "scape_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, fn)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{fn} mentions the oracle"
    for fn in os.listdir(os.path.join(ROOT, "scape")):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "scape", fn)).read()


# ---------------------------------------------------------------- chunk reader
def _chunk_bytes(utrs):
    out = b""
    for gene, df in utrs:
        out += pickle.dumps((gene, df), protocol=4)
    return out


def test_safe_pickle_reads_prepare_input_layout():
    from scape_amd import safe_pickle
    rng = np.random.default_rng(3)
    utrs = []
    for i in range(3):
        n = 50 + 10 * i
        df = pd.DataFrame({"x": rng.integers(0, 1500, n), "l": rng.integers(31, 133, n),
                           "r": np.full(n, np.nan), "pa": np.where(rng.random(n) < 0.1, 800.0, np.nan),
                           "cb_id": rng.integers(0, 9999, n), "read_id": np.arange(n),
                           "junction": rng.integers(0, 2, n), "seg1_en": np.full(n, np.nan),
                           "seg2_en": rng.random(n)})
        utrs.append((f"chr1:G{i}:1:10-2000:+", df))
    got = safe_pickle.load_all(_chunk_bytes(utrs))
    assert len(got) == 3
    for (g0, d0), (g1, d1) in zip(utrs, got):
        assert g0 == g1 and list(d0.columns) == list(d1.columns)
        for c in d0.columns:
            assert np.array_equal(d0[c].values, d1[c].values, equal_nan=True)
            assert d0[c].dtype == d1[c].dtype


def test_safe_pickle_refuses_code_execution(tmp_path):
    from scape_amd import safe_pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("touch " + str(tmp_path / "pwned"),))
    data = pickle.dumps(("g", Evil()))
    with pytest.raises(safe_pickle.UnsafePickleError):
        safe_pickle.load_all(data)
    assert not (tmp_path / "pwned").exists()


def test_parameters_pickles_under_reference_class_path():
    from scape.apa_core import Parameters
    from scape_amd import safe_pickle
    p = Parameters(title="Final Result", alpha_arr=np.array([623]), beta_arr=np.array([10.0]),
                   ws=np.array([0.9, 0.1]), L=2000, cb_id_arr=np.arange(3), readID_arr=np.arange(3))
    p.bic = np.float64(1.5)
    p.lb_arr = [np.float64(-3.0), np.float64(-2.0)]
    p.label_arr = np.array([0, 0, 1])
    p.gene_info_str = "chr1:G:1:1-2:+"
    blob = pickle.dumps(p)
    assert b"scape.apa_core" in blob and b"Parameters" in blob
    q = pickle.loads(blob)
    assert type(q) is Parameters and q.K == 1 and "K=1" in str(q)
    s = safe_pickle.load_all(blob)[0]
    assert s.gene_info_str == p.gene_info_str and np.array_equal(s.label_arr, p.label_arr)
    assert [float(v) for v in s.lb_arr] == [-3.0, -2.0]


# ---------------------------------------------------------------- binning / coverage / grids
def test_binning_matches_oracle_on_random_reads(oracle):
    from scape_amd.host import bin_reads
    rng = np.random.default_rng(5)
    for case in range(6):
        n = int(rng.integers(1, 3000))
        x = rng.integers(0, 2500, n).astype(float)
        l = rng.integers(31, 133, n).astype(float)
        r = np.full(n, np.nan)
        pa = np.full(n, np.nan)
        if case % 2:
            m = rng.random(n) < 0.2
            pa[m] = rng.integers(100, 2600, m.sum())
        if case >= 3:
            m = rng.random(n) < 0.3
            r[m] = rng.integers(1, 140, m.sum())
        got = bin_reads(x, l, r, pa)
        want = oracle.bin_reads(x, l, r, pa)
        for g, w in zip(got, want):
            assert np.array_equal(g, w, equal_nan=True)


def test_prepare_utr_matches_reference_trace():
    from scape_amd.host import coverage_profile, prepare_utr
    for name in ("synA", "synB", "chr19"):
        f = load_npz(f"trace_{name}.npz")
        p = trace_params(f)
        for i in range(int(f["n_utr"])):
            gene, df = utr_df(f, i)
            q = prepare_utr(df, gene_info_str=gene, **p)
            for k, v in (("x", q.x), ("l", q.l), ("r", q.r), ("pa", q.pa)):
                assert np.array_equal(v, f[f"u{i}_bin_{k}"], equal_nan=True)
            assert np.array_equal(q.cnt, f[f"u{i}_bin_cnt"]) and np.array_equal(q.idx, f[f"u{i}_bin_idx"])
            assert np.array_equal(q.theta, f[f"u{i}_all_theta"]) and np.array_equal(q.betas, f[f"u{i}_betas"])
            assert q.L == int(f[f"u{i}_L"]) and q.min_theta == float(f[f"u{i}_min_theta"])
            assert q.unif_ll == float(f[f"u{i}_unif_ll"])
            _, ys = coverage_profile(q.x, q.l, q.cnt, q.L, q.p["beta_step"])
            assert np.array_equal(ys, f[f"u{i}_cov_y"])         # bit-equal to ker_smooth's loops


def test_sampler_consumes_reference_stream():
    """init_para + gen_k_arr draws equal the reference's, call for call (first sweep of each UTR)."""
    from scape_amd.host import Sampler, prepare_utr
    for name in ("synA", "synB", "chr17"):
        f = load_npz(f"trace_{name}.npz")
        p = trace_params(f)
        for i in range(int(f["n_utr"])):
            gene, df = utr_df(f, i)
            q = prepare_utr(df, gene_info_str=gene, **p)
            rs = np.random.RandomState(0)
            st = ("MT19937", f[f"u{i}_rng_keys"], int(f[f"u{i}_rng_pos"]), 0, 0.0)
            rs.set_state(st)
            smp, c = Sampler(rs), 0
            for K in range(p["n_max_apa"], p["n_min_apa"] - 1, -1):
                for _ in range(10):
                    a, b, w, ka = smp.init_job(q, K)
                    assert np.array_equal(q.theta[a], f[f"u{i}_call_a0"][c, :K])
                    assert np.array_equal(q.betas[b], f[f"u{i}_call_b0"][c, :K])
                    assert np.array_equal(w, f[f"u{i}_call_w0"][c, :K + 1])
                    assert np.array_equal(ka, f[f"u{i}_call_k_arr"][c].astype(int))
                    c += 1


def test_native_sampler_equals_numpy_stream():
    """libscape_host.so draws the numbers numpy's legacy RandomState draws (both init_para branches:
    K <= #peaks -> choice with p, K > #peaks -> permutation(L)), and leaves the same generator state."""
    from scape_amd.host import FastSampler, Sampler, prepare_utr
    from scape_amd.synth import synth_chunk
    from scape_amd import _hostlib
    lib = _hostlib.load_library()
    st = np.zeros(625, dtype=np.uint32)
    for seed in (0, 1, 12345, 2 ** 32 - 1):
        lib.scape_host_mt_seed(seed, _hostlib.ptr(st, _hostlib.P_u32))
        ref = np.random.RandomState(seed)
        g = ref.get_state()
        assert np.array_equal(st[:624], g[1]) and st[624] == g[2]
        assert [lib.scape_host_mt_double(_hostlib.ptr(st, _hostlib.P_u32)) for _ in range(700)] == list(ref.random_sample(700))
    preps = [prepare_utr(df, g, n_max_apa=12, n_min_apa=1) for g, df in synth_chunk(10, 600, base_seed=5)]
    f = load_npz("trace_chr17.npz")
    preps.append(prepare_utr(utr_df(f, 0)[1], gene_info_str="chr17", **trace_params(f)))
    for u, q in enumerate(preps):
        r1, r2 = np.random.RandomState(100 + u), np.random.RandomState(100 + u)
        s1, s2 = Sampler(r1), FastSampler(r2)
        for K in list(range(14, 0, -1)) + [25]:
            for _ in range(3):
                for x, y in zip(s1.init_job(q, K), s2.init_job(q, K)):
                    assert np.array_equal(x, y) and x.dtype == y.dtype, (u, K)
            assert np.array_equal(s1.init_ws(K - 1, 0.15), s2.init_ws(K - 1, 0.15))
            assert np.array_equal(s1.k_arr(K - 1), s2.k_arr(K - 1))
        g1, g2 = r1.get_state(), s2.rs.get_state()
        assert np.array_equal(g1[1], g2[1]) and g1[2] == g2[2]
    # what numpy refuses, the native path declines and numpy then raises as usual
    q = preps[0]
    with pytest.raises(ValueError):
        FastSampler(np.random.RandomState(1)).init_job(q, q.L + len(q.peaks) + 5)


def test_native_sweep_equals_job_by_job_draws():
    """FastSampler.sweep (one native call per K sweep, the reference-stream fast path) == init_job per restart."""
    from scape_amd.host import FastSampler, Sampler, prepare_utr
    from scape_amd.synth import synth_chunk
    for u, (g, df) in enumerate(synth_chunk(6, 500, base_seed=21)):
        q = prepare_utr(df, g, n_max_apa=12, n_min_apa=1)
        r1, r2 = np.random.RandomState(9 + u), np.random.RandomState(9 + u)
        s1, s2 = Sampler(r1), FastSampler(r2)
        for n_max, n_min in ((12, 1), (5, 2), (14, 12), (3, 3)):
            jk, a, b, w, ka = s2.sweep(q, n_max, n_min)
            i = 0
            for K in range(n_max, n_min - 1, -1):
                for _ in range(10):
                    ea, eb, ew, eka = s1.init_job(q, K)
                    assert jk[i] == K and np.array_equal(a[i, :K], ea) and np.array_equal(b[i, :K], eb)
                    assert np.array_equal(w[i, :K + 1], ew) and np.array_equal(ka[i], eka)
                    assert not a[i, K:].any() and not w[i, K + 1:].any()
                    i += 1
            assert i == len(jk)
        g1, g2 = r1.get_state(), s2.rs.get_state()
        assert np.array_equal(g1[1], g2[1]) and g1[2] == g2[2]


def test_native_sweep_many_equals_single_sweeps():
    """The threaded batch call for the heads of several streams draws what the per-stream calls draw."""
    from scape_amd.host import FastSampler, prepare_utr
    from scape_amd.synth import synth_chunk
    preps = [prepare_utr(df, g, n_max_apa=10, n_min_apa=1) for g, df in synth_chunk(7, 400, base_seed=31)]
    one = [FastSampler(np.random.RandomState(40 + i)) for i in range(7)]
    many = [FastSampler(np.random.RandomState(40 + i)) for i in range(7)]
    a = [s.sweep(q, 10 - i % 3, 1 + i % 2) for i, (s, q) in enumerate(zip(one, preps))]
    b = FastSampler.sweep_many([(s, q, 10 - i % 3, 1 + i % 2) for i, (s, q) in enumerate(zip(many, preps))], n_threads=3)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
    for s, t in zip(one, many):
        assert np.array_equal(s.state, t.state)
    with pytest.raises(ValueError):
        FastSampler.sweep_many([(one[0], preps[0], 3, 1), (one[0], preps[1], 3, 1)])


def test_native_plan_equals_python_plan():
    """Engine.plan (scape_host_plan, threaded) == Engine.plan_python (numpy RandomState), table for table,
    for mixed K ranges and seeds at the 32-bit edge."""
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_chunk
    chunk = synth_chunk(12, 500, base_seed=9)
    preps = [prepare_utr(df, g, n_max_apa=(10 if i % 3 else 4), n_min_apa=(1 if i % 2 else 2))
             for i, (g, df) in enumerate(chunk)]
    seeds = [(2 ** 32 - 4 + i) % 2 ** 32 for i in range(len(preps))]
    P1 = Engine.plan_python(preps, seeds)
    for nt in (1, 3):
        P2 = Engine.plan(preps, seeds, n_threads=nt)
        for k in ("ju", "jk", "jf", "a", "b", "w", "ka"):
            assert np.array_equal(getattr(P1["main"], k), getattr(P2["main"], k)), k
        for k in ("spans", "states", "prune_w", "prune_ka", "prune_states"):
            assert np.array_equal(P1[k], P2[k]), k


def test_snap_to_grid_ties_go_up(oracle):
    from scape_amd.host import snap_to_grid
    grid = np.arange(31, 2000, 9) + 0.0
    vals = np.array([-5.0, 31.0, 35.5, 35.4, 35.6, 40.0, 1999.0, 5000.0, 44.5])
    assert np.array_equal(snap_to_grid(grid, vals), oracle.nearest_on_grid(grid, vals)[0])


def test_k_arr_rules():
    from scape_amd.host import Sampler
    s = Sampler(np.random.RandomState(4))
    st = s.rs.get_state()[2]
    assert np.array_equal(s.k_arr(1), np.zeros(50)) and np.array_equal(s.k_arr(0), np.zeros(50))
    assert s.rs.get_state()[2] == st                       # K <= 1 draws nothing (apa_core.py:671-672)
    ka = s.k_arr(4)
    for t0 in range(0, 48, 4):
        assert sorted(ka[t0:t0 + 4]) == [0, 1, 2, 3]


# ---------------------------------------------------------------- pre-binned columnar chunks
def test_binned_chunk_gives_identical_preps_and_detects_staleness(tmp_path):
    from scape_amd import binned
    from scape_amd.apa_core import load_preps
    from scape_amd.synth import synth_chunk
    f = load_npz("fixture_chr17.npz")
    utrs = [utr_df(f, i) for i in range(int(f["n_utr"]))] + synth_chunk(5, 400, base_seed=8, pa_rate=0.05)
    rng = np.random.default_rng(1)
    full = []
    for g, df in utrs:                                  # the three columns merge_pa reads besides read_id
        df = df.copy()
        n = len(df)
        df["junction"] = (rng.random(n) < 0.1).astype(np.int64)
        df["seg1_en"] = np.where(df["junction"] == 1, rng.integers(1000, 9000, n).astype(float), np.nan)
        df["seg2_en"] = np.where(df["junction"] == 1, rng.integers(1000, 9000, n).astype(float), np.nan)
        full.append((g, df))
    path = str(tmp_path / "c.100.1.1.input.pkl")
    with open(path, "wb") as fh:
        for it in full:
            pickle.dump(it, fh)
    kw = dict(n_max_apa=4, n_min_apa=1)
    want = load_preps(path, kw)                         # no binned file yet: decodes the pickle
    bp = binned.prebin_chunk(path)
    assert bp == str(tmp_path / "c.100.1.1.binned.npz") and binned.is_current(bp, path)
    with np.load(bp, allow_pickle=False) as z:          # plain arrays only
        assert int(z["version"]) == binned.VERSION and len(z["gene_info"]) == len(full)
    got = load_preps(path, kw)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        for k in ("x", "l", "r", "pa", "cnt", "idx", "cb_id", "read_id", "theta", "betas", "peaks", "peak_w", "s_dis"):
            assert np.array_equal(getattr(a, k), getattr(b, k), equal_nan=True), k
        assert (a.gene_info_str, a.L, a.min_theta, a.unif_ll, a.p) == (b.gene_info_str, b.L, b.min_theta, b.unif_ll, b.p)
    for (g, b, cols), (g0, df) in zip(binned.read_binned(bp), full):
        assert g == g0 and np.array_equal(cols["junction"], df["junction"])
        assert np.array_equal(cols["seg1_en"], df["seg1_en"], equal_nan=True)
        assert np.array_equal(cols["seg2_en"], df["seg2_en"], equal_nan=True)
    with open(path, "ab") as fh:                        # the chunk changes -> the binned copy is stale
        pickle.dump(full[0], fh)
    assert not binned.is_current(bp, path)
    assert len(load_preps(path, kw)) == len(full) + 1


# ---------------------------------------------------------------- CLI behaviour (reference apa_core.py:78-132)
def test_cli_error_behaviour(tmp_path):
    from click.testing import CliRunner
    from scape.cli import cli
    out = tmp_path / "out"
    out.mkdir()
    run = CliRunner()
    # parameters.toml must exist (assert at apa_core.py:85)
    r = run.invoke(cli, ["infer_pa", "--pkl_input_file", str(tmp_path / "a.100.1.1.input.pkl"), "--output_dir", str(out)])
    assert isinstance(r.exception, AssertionError)
    (out / "parameters.toml").write_text("n_max_apa = 3\noutput_dir = \"x\"\n")
    r = run.invoke(cli, ["infer_pa", "--pkl_input_file", str(tmp_path / "missing.input.pkl"), "--output_dir", str(out)])
    assert "does not exists" in str(r.exception)
    tmp_in = tmp_path / "b.tmp.100.1.1.input.pkl"
    tmp_in.write_bytes(b"")
    r = run.invoke(cli, ["infer_pa", "--pkl_input_file", str(tmp_in), "--output_dir", str(out)])
    assert "incomplete" in str(r.exception)
    r = run.invoke(cli, ["infer_pa", "--pkl_input_file", str(tmp_in), "--output_dir", str(tmp_path / "nope")])
    assert isinstance(r.exception, AssertionError)


def test_infer_all_resume_skips_finished_chunks(tmp_path, monkeypatch):
    from scape_amd import apa_core
    d = tmp_path / "o"
    (d / "pkl_input").mkdir(parents=True)
    (d / "pkl_output").mkdir()
    for i in range(4):
        (d / "pkl_input" / f"c.100.4.{i + 1}.input.pkl").write_bytes(b"x")
    (d / "pkl_input" / "c.tmp.100.4.9.input.pkl").write_bytes(b"x")          # incomplete chunk: never processed
    (d / "pkl_output" / "c.100.4.2.res.pkl").write_bytes(b"y")                 # finished
    (d / "pkl_output" / "c.100.4.3.res.pkl").write_bytes(b"y")                 # older than its chunk -> redo
    os.utime(d / "pkl_output" / "c.100.4.3.res.pkl", (1, 1))
    seen = []
    monkeypatch.setattr(apa_core, "infer_files", lambda files, out, **kw: seen.append([os.path.basename(f) for f in files]))
    apa_core.infer_all(str(d), gpus=1, resume=True, n_max_apa=3)
    assert seen == [["c.100.4.1.input.pkl", "c.100.4.3.input.pkl", "c.100.4.4.input.pkl"]]
    apa_core.infer_all(str(d), gpus=1, n_max_apa=3)
    assert len(seen[1]) == 4


def test_wave_split_respects_budget(monkeypatch):
    from scape_amd import engine

    class Q:
        def __init__(self, N, T):
            self.N, self.T, self.betas, self.gene_info_str = N, T, np.zeros(13), "g"
    eng = engine.Engine.__new__(engine.Engine)
    eng.mem_fraction = 1.0
    preps = [Q(100, 218), Q(1000, 300), Q(50, 218), Q(1200, 400)]
    sizes = [engine.Engine.utr_bytes(q) for q in preps]
    monkeypatch.setattr(engine.Engine, "_budget", lambda self: sizes[1] + sizes[2])
    waves = eng.waves(preps[:3])
    assert waves == [[0], [1, 2]]
    with pytest.raises(Exception):
        eng.waves(preps)


# ---------------------------------------------------------------- exact-stream driver (host logic only, no GPU)
class _FakeBatch:
    """Stands in for HipBatch.em_packed in the stream-driver tests: every job's "fit" is a pure function of that job's
    own table row (as on the GPU, where a job's bits do not depend on what shares its call), made up so that sweeps end
    clean, pruned to various K' or at K = n_max (re-run)."""
    calls = 0

    def em_packed(self, pj, reuse_buffers=False, want_lb=True):
        import zlib
        from scape_amd.host import N_ROUND
        type(self).calls += 1
        n, kmax = len(pj), pj.kmax
        ao, bo, wo = pj.a.copy(), pj.b.copy(), np.zeros_like(pj.w)
        bic, nlb, lb = np.zeros(n), np.full(n, 3, np.int32), np.zeros((n, N_ROUND))
        for i in range(n):
            K = int(pj.jk[i])
            h = zlib.crc32(pj.a[i, :K].tobytes() + pj.b[i, :K].tobytes() + pj.w[i, :K + 1].tobytes() + pj.ka[i].tobytes() +
                           bytes([K, int(pj.jf[i])]))          # (the row up to K: the padding width depends on the call)
            r = np.random.RandomState(h)
            w = r.dirichlet(np.full(K + 1, 0.6))
            w[0] += 0.06                                   # (one component always stays: rm_component needs K' >= 1)
            w /= w.sum()
            wo[i, :K + 1] = w
            bic[i] = 2.0 * abs(K - 2.3) + 30.0 * r.random_sample()
            lb[i, :3] = r.random_sample(3)
        return ao, bo, wo, bic, nlb, lb


def _stream_preps(n, base):
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    return [prepare_utr(df, gene_info_str=g, n_max_apa=3, n_min_apa=1)
            for g, df, _ in (synth_utr(i, 150, k_cap=3, base_seed=base) for i in range(n))]


def _drive(chunks, depth, preds=None, seed=7):
    from scape_amd.engine import Engine, _Sweep
    from scape_amd.host import FastSampler
    streams, all_sw = [], []
    for ci, preps in enumerate(chunks):
        smp = FastSampler(np.random.RandomState(seed + ci))
        sws = [_Sweep(u, q, smp, True) for u, q in enumerate(preps)]
        if preds is not None:
            for sw, p in zip(sws, preds[ci]):
                sw.pred = p
        streams.append((smp, sws))
        all_sw.append(sws)
    deferred = []
    Engine._drive_streams(_FakeBatch(), streams, deferred, depth=depth)
    Engine._finish_deferred(_FakeBatch(), deferred)
    sig = [[(sw.best.K, sw.best.a_idx.tobytes(), sw.best.b_idx.tobytes(), sw.best.ws.tobytes(), sw.best.bic, sw.n_jobs)
            for sw in sws] for sws in all_sw]
    return sig, [smp.state.copy() for smp, _ in streams]


def test_stream_driver_results_do_not_depend_on_depth_or_predictions(monkeypatch):
    """Engine._drive_streams: several UTRs of a stream per EM call, drawn from the state their predecessors' PREDICTED
    outcome leaves.  Whatever the predictions say - right, wrong, absent, 'stop' - every UTR must get the fit, the job
    count and the generator state of the strictly serial loop (depth 1); also with the look-ahead draws on and off and in the
    two-halves schedule of many streams."""
    from scape_amd.engine import Engine
    chunks = [_stream_preps(14, 9100), _stream_preps(9, 9200), _stream_preps(5, 9300)]
    ref, ref_states = _drive(chunks, 1)
    outcomes = {(k, nj) for ch in ref for (k, *_rest, nj) in ch}
    assert len({nj for _k, nj in outcomes}) >= 3, outcomes          # clean (30 jobs), pruned (+1) and re-run sweeps all occur
    rs = np.random.RandomState(3)
    true_pred = [[("clean" if nj == 30 else ("prune", k) if nj == 31 else "stop") for (k, *_r, nj) in ch] for ch in ref]
    for trial in range(6):
        if trial == 0:
            preds = None                                              # no predictor: followers assume a clean end
        elif trial == 1:
            preds = true_pred                                         # an oracle predictor: (almost) nothing discarded
        else:                                                         # right, wrong and missing predictions mixed
            preds = [[p if rs.random_sample() < 0.6 else rs.choice(["clean", "stop", None, "p1", "p2"]) for p in ch]
                     for ch in true_pred]
            preds = [[("prune", int(p[1])) if isinstance(p, str) and p.startswith("p") else p for p in ch] for ch in preds]
        for depth, ahead, pp in ((3, True, 32), (8, True, 32), (8, False, 32), (4, True, 2)):
            monkeypatch.setattr(Engine, "draw_ahead", ahead)
            monkeypatch.setattr(Engine, "pingpong_min_streams", pp)
            before = dict(Engine.spec_stats)
            got, states = _drive(chunks, depth, preds)
            assert got == ref, (trial, depth, ahead, pp)
            for a, b in zip(states, ref_states):
                assert np.array_equal(a, b), (trial, depth, ahead, pp)
            if trial == 1 and pp == 32:
                assert Engine.spec_stats["utrs_kept"] - before["utrs_kept"] >= sum(len(c) for c in chunks)


def test_prediction_pass_failure_is_not_a_stream_failure():
    """Engine._predict_outcomes fits the wave from OTHER seeds; when that fit prunes every component of some UTR
    (the reference's IndexError, apa_core.py:838-839) or the library refuses its job tables, the real stream must go
    on without predictions and the measurement fields of the real sweeps must stay as they were."""
    from scape_amd import _lib
    from scape_amd.engine import Engine, _Sweep
    from scape_amd.host import FastSampler
    preps = _stream_preps(4, 9400)
    for exc in (IndexError("index 0 is out of bounds for axis 0 with size 0"), _lib.ScapeHipError("job table refused")):
        eng = Engine.__new__(Engine)              # no GPU: only the prediction wrapper is exercised
        eng.last_main_em_ms, eng.last_main_counters = 12.5, (1, 2, 3)

        def boom(*a, _e=exc, **k):
            eng.last_main_em_ms, eng.last_host_ms = -1.0, dict(x=1)      # what a half-finished process() leaves behind
            raise _e
        eng.plan = lambda preps, seeds: None
        eng.process = boom
        smp = FastSampler(np.random.RandomState(1))
        sweeps = [_Sweep(u, q, smp, True) for u, q in enumerate(preps)]
        for sw in sweeps:
            sw.pred = "clean"
        eng._predict_outcomes(None, preps, sweeps, True)
        assert all(sw.pred is None for sw in sweeps)
        assert eng.last_main_em_ms == 12.5 and eng.last_main_counters == (1, 2, 3)
        assert not hasattr(eng, "last_host_ms")


def test_find_peaks_equals_scipy():
    """scape_amd.host.find_peaks restates scipy.signal.find_peaks(x, distance=d) (apa_core.py:784) so that the CLI
    does not import scipy.signal; it must return exactly scipy's peaks - also with plateaus, ties between peak
    heights and peaks closer than the distance."""
    from scipy.signal import find_peaks as sp
    from scape_amd.host import find_peaks
    rs = np.random.RandomState(5)
    cases = []
    for trial in range(300):
        n = int(rs.randint(3, 400))
        kind = trial % 4
        if kind == 0:
            x = rs.random_sample(n)
        elif kind == 1:
            x = rs.randint(0, 4, n).astype(float)                     # plateaus and equal heights everywhere
        elif kind == 2:
            x = np.convolve(rs.poisson(2.0, n).astype(float), np.ones(7) / 7, mode="same")
        else:
            x = np.repeat(rs.randint(0, 6, n // 3 + 1), 3)[:n].astype(float)
        cases.append(x)
    cases += [np.zeros(5), np.array([0., 1, 0]), np.array([1., 1, 1]), np.array([0., 2, 2, 0, 2, 2, 2, 0]), np.arange(10.0)]
    for x in cases:
        for d in (1, 2, 3.5, 10, 100):
            want = sp(x, distance=d)[0]
            got = find_peaks(x, d)[0]
            assert np.array_equal(got, want), (x.tolist(), d, got, want)
    assert len(find_peaks(np.array([1.0, 2.0]), 5)[0]) == 0
