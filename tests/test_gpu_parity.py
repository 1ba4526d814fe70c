"""GPU parity tests: the HIP path (through the ctypes C-ABI) against
  - the reference's own outputs recorded in tests/golden/trace_*.npz,
  - the CPU oracle on seeded inputs,
  - size-independent properties at the BASELINE sizes.
Tolerances: K / alpha / beta / labels / round counts exact; f64 values to 1e-9 relative or better
(the north star allows 1e-4 relative on positions and weights)."""
import math
import os
import pickle

import numpy as np
import pandas as pd
import pytest

from conftest import FIXTURES, TRACES, load_npz, trace_params, trace_pre_para, utr_df

pytestmark = pytest.mark.gpu
SENT = float(np.finfo("f").min)


def _same_sent(a, b):
    return np.array_equal(a == SENT, b == SENT)


# ---------------------------------------------------------------- operator level (taichi_core seam)
def test_known_answers_through_hip():
    """Inputs of the reference's taichi_code_test.py:514-593 pushed through the HIP operators."""
    from scape import taichi_core as tc
    one = lambda v: np.array([float(v)])
    # loglik_xlr_t_pa = loglik_l_xt(x,l,theta) + logN(pa-theta; 0, sigma_f)
    got = tc.loglik_xlr_t_pa(one(5), one(50), one(187), 70.0, 50)[0]
    z = (187 - 70) / 50
    assert got == pytest.approx(-math.log(65.0) - 0.5 * z * z - math.log(50) - 0.5 * math.log(2 * math.pi), rel=1e-14)
    got = tc.loglik_xlr_t_pa(one(30), one(50), one(187), 70.0, 50)[0]          # l > theta - x
    assert got == pytest.approx(SENT, rel=1e-15)
    s = np.arange(20, 150, 10).astype(float)
    pmf = np.full(13, 1 / 13)
    # r_unknown: sum_j 1/s_j N(x; theta+s_j-mu_f, sigma_f) 1/(theta-x) pmf_j
    x, l, th = 44.0, 50.0, 460.0
    v = sum(1 / sj * math.exp(-0.5 * ((x - (th + sj - 300)) / 50) ** 2) / math.sqrt(2 * math.pi) / 50 / (th - x) / 13 for sj in s)
    got = tc.loglik_xlr_t_r_unknown(one(x), one(l), None, s, pmf, th, 300, 50)[0]
    assert got == pytest.approx(math.log(v), rel=1e-13)
    # r_known with r=20 > s_0: first term dropped, normalised by the kept mass
    r = 25.0
    terms = [-math.log(sj) - 0.5 * ((x - (th + sj - 300)) / 50) ** 2 - math.log(50) - 0.5 * math.log(2 * math.pi)
             - math.log(th - x) + math.log(1 / 13) for sj in s if sj >= r]
    want = math.log(sum(math.exp(t - max(terms)) for t in terms)) + max(terms) - math.log(len(terms) / 13)
    got = tc.loglik_xlr_t_r_known(one(x), one(l), one(r), s, pmf, th, 300, 50)[0]
    assert got == pytest.approx(want, rel=1e-13)
    # marginal on the reference's 23-point grid (alpha=37, beta=5 -> two grid points)
    all_theta = np.arange(37, 236, 9).astype(float)
    rng = np.random.RandomState(0)
    for i in range(3):
        A = rng.rand(i + 3, i + 5)
        got = tc.loglik_marginal_lxr(37.0, 5.0, all_theta, A)
        g = -0.5 * ((all_theta[:2] - 37) / 5) ** 2 - math.log(5) - 0.5 * math.log(2 * math.pi)
        want = np.log(np.exp(A[:, :2] + g - math.log(np.exp(g).sum())).sum(axis=1))
        assert np.allclose(got, want, rtol=1e-14, atol=0)


def test_point_kernels_vs_oracle(oracle):
    from scape import taichi_core as tc
    rng = np.random.default_rng(1)
    n = 4000
    x = rng.integers(0, 2000, n).astype(float) + rng.random(n).round(2)
    l = rng.integers(31, 133, n).astype(float)
    pa = x + rng.integers(40, 600, n)
    r = rng.integers(1, 150, n).astype(float)
    s = np.arange(20, 150, 10).astype(float)
    pmf = rng.dirichlet(np.ones(13))
    for theta in (1200.0, 1195.0, 1205.0, 31.0, 2500.5):
        a, b = tc.loglik_xlr_t_pa(x, l, pa, theta, 50), oracle.loglik_xlr_t_pa(x, l, pa, theta, 50)
        assert _same_sent(a, b) and np.allclose(a, b, rtol=1e-14, atol=0)
        a = tc.loglik_xlr_t_r_unknown(x, l, r, s, pmf, theta, 300, 50)
        b = oracle.loglik_xlr_t_r_unknown(x, l, r, s, pmf, theta, 300, 50)
        assert _same_sent(a, b) and np.allclose(a, b, rtol=1e-13, atol=0)
        a = tc.loglik_xlr_t_r_known(x, l, r, s, pmf, theta, 300, 50)
        b = oracle.loglik_xlr_t_r_known(x, l, r, s, pmf, theta, 300, 50)
        assert np.allclose(a, b, rtol=1e-13, atol=0)
    assert len(tc.loglik_xlr_t_pa(x[:0], l[:0], pa[:0], 100.0, 50)) == 0      # empty input


def test_marginal_tensor_operator_vs_oracle(oracle):
    from scape import taichi_core as tc
    rng = np.random.default_rng(2)
    th = np.arange(31, 1200, 9).astype(float)
    betas = np.arange(5, 70, 5).astype(float)
    A = -rng.random((37, len(th))) * 700
    A[rng.random(A.shape) < 0.3] = SENT
    A[5] = SENT                                        # a bin that is impossible everywhere
    got, want = tc.get_loglik_marginal_tensor(th, betas, A), oracle.get_loglik_marginal_tensor(th, betas, A)
    assert got.shape == (len(th), 13, 37)
    assert _same_sent(got, want) and np.allclose(got, want, rtol=1e-13, atol=0)
    # non-uniform grid (fixed_run style, apa_core.py:895) and a single beta
    th2 = np.unique(np.concatenate([th[:20], th[40:70], th[100:]]))
    A2 = A[:, :len(th2)]
    got, want = tc.get_loglik_marginal_tensor(th2, betas[3:4], A2), oracle.get_loglik_marginal_tensor(th2, betas[3:4], A2)
    assert _same_sent(got, want) and np.allclose(got, want, rtol=1e-13, atol=0)


# ---------------------------------------------------------------- batched build vs the reference's own tensors
def _batch_for_trace(hip_ctx, name):
    from scape_amd.engine import HipBatch
    from scape_amd.host import prepare_utr
    f = load_npz(f"trace_{name}.npz")
    p = trace_params(f)
    preps = []
    for i in range(int(f["n_utr"])):
        gene, df = utr_df(f, i)
        preps.append(prepare_utr(df, gene_info_str=gene, pre_para=trace_pre_para(f), **p))
    batch = HipBatch(hip_ctx, preps)
    batch.build()
    return f, p, preps, batch


@pytest.mark.parametrize("name", TRACES)
def test_phase_a_b_vs_reference_trace(hip_ctx, name):
    f, p, preps, batch = _batch_for_trace(hip_ctx, name)
    for i, q in enumerate(preps):
        A = batch.fetch_loglik(i)
        ref = f[f"u{i}_A"]
        assert _same_sent(A, ref) and np.allclose(A, ref, rtol=1e-13, atol=0), (name, i)
        M = batch.fetch_tensor(i)
        if f"u{i}_M" in f.files:
            Mr, Mg = f[f"u{i}_M"], M
        else:
            Mr, Mg = f[f"u{i}_M_sel"], M[f[f"u{i}_M_rows"]]
        assert _same_sent(Mg, Mr), (name, i)
        # log-domain values; the linear-domain window sum differs from the reference's
        # max-shifted logsumexp only by rounding (|M| <= ~700 -> abs error ~1e-13)
        assert np.allclose(Mg, Mr, rtol=2e-13, atol=2e-12), (name, i, np.abs(Mg - Mr)[Mr > -1e30].max())
        fin = M > -1e30
        assert int((~fin).sum()) == int(f[f"u{i}_M_n_sent"])
        assert float(M[fin].sum()) == pytest.approx(float(f[f"u{i}_M_sum_finite"]), rel=1e-11)
    batch.free()


@pytest.mark.parametrize("name", TRACES)
def test_em_calls_vs_reference_trace(hip_ctx, name):
    """Every em_algo call the reference made (inits + k_arr from the trace) re-run on the GPU."""
    from scape_amd.engine import _Job
    f, p, preps, batch = _batch_for_trace(hip_ctx, name)
    jobs, meta = [], []
    for i, q in enumerate(preps):
        for c in range(len(f[f"u{i}_call_K"])):
            K = int(f[f"u{i}_call_K"][c])
            a0 = f[f"u{i}_call_a0"][c, :K]
            a_idx = np.searchsorted(q.theta, a0).astype(np.int32)
            assert np.array_equal(q.theta[a_idx], a0)
            b_idx = np.searchsorted(q.betas, f[f"u{i}_call_b0"][c, :K]).astype(np.int32)
            jobs.append(_Job(i, K, bool(f[f"u{i}_call_fixed"][c]), a_idx, b_idx, f[f"u{i}_call_w0"][c, :K + 1],
                             f[f"u{i}_call_k_arr"][c].astype(np.int8)))
            meta.append((i, c, K))
    fits = batch.em(jobs)
    n_exact = 0
    for (i, c, K), ft in zip(meta, fits):
        q = preps[i]
        tag = (name, i, c, K)
        same = (np.array_equal(q.theta[ft.a_idx], f[f"u{i}_call_a1"][c, :K])
                and np.array_equal(q.betas[ft.b_idx], f[f"u{i}_call_b1"][c, :K])
                and len(ft.lb) == int(f[f"u{i}_call_nlb"][c]))
        assert same, tag
        n_exact += same
        assert np.allclose(ft.ws, f[f"u{i}_call_w1"][c, :K + 1], rtol=1e-9, atol=1e-13), tag
        assert ft.bic == pytest.approx(float(f[f"u{i}_call_bic"][c]), rel=1e-10), tag
        assert np.allclose(ft.lb, f[f"u{i}_call_lb"][c, :len(ft.lb)], rtol=1e-10), tag
    assert n_exact == len(jobs)
    batch.free()


@pytest.mark.parametrize("name", TRACES)
def test_full_pipeline_reference_stream(name):
    """Engine in 'reference' RNG mode reproduces the reference's run of the same chunk."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    f = load_npz(f"trace_{name}.npz")
    p = trace_params(f)
    preps = [prepare_utr(utr_df(f, i)[1], gene_info_str=utr_df(f, i)[0], pre_para=trace_pre_para(f), **p)
             for i in range(int(f["n_utr"]))]
    eng = Engine(device=0)
    res = eng.run(preps, rng_mode="reference", seed=int(f["seed"]), re_run_mode=bool(p["re_run_mode"]), keep_trace=True)
    for i, r in enumerate(res):
        para = to_parameters(r)
        assert len(eng.traces[i]) == len(f[f"u{i}_call_K"]), (name, i)
        assert para.K == int(f[f"u{i}_res_K"])
        assert np.array_equal(para.alpha_arr, f[f"u{i}_res_alpha_arr"]) and para.alpha_arr.dtype == np.int64
        assert np.array_equal(para.beta_arr, f[f"u{i}_res_beta_arr"])
        assert np.allclose(para.ws, f[f"u{i}_res_ws"], rtol=1e-9, atol=1e-13)
        assert para.bic == pytest.approx(float(f[f"u{i}_res_bic"]), rel=1e-10)
        assert np.array_equal(para.label_arr, f[f"u{i}_res_label_arr"])
        assert np.allclose(para.lb_arr, f[f"u{i}_res_lb_arr"], rtol=1e-10)
        assert para.L == int(f[f"u{i}_L"]) and para.title == str(f[f"u{i}_res_title"])
        assert np.array_equal(para.cb_id_arr, f[f"u{i}_cb_id"]) and np.array_equal(para.readID_arr, f[f"u{i}_read_id"])


@pytest.mark.parametrize("name", TRACES)
def test_reference_stream_with_followers_vs_reference_trace(name):
    """The CLI's default path (no call trace kept): the UTRs of the chunk's one random stream ride several per EM call
    (Engine._drive_streams) - the reference's final Parameters and the generator state it leaves must come out."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    f = load_npz(f"trace_{name}.npz")
    p = trace_params(f)
    preps = [prepare_utr(utr_df(f, i)[1], gene_info_str=utr_df(f, i)[0], pre_para=trace_pre_para(f), **p)
             for i in range(int(f["n_utr"]))]
    eng = Engine(device=0)
    before = dict(Engine.spec_stats)
    res = eng.run(preps, rng_mode="reference", seed=int(f["seed"]), re_run_mode=bool(p["re_run_mode"]))
    assert Engine.spec_stats["utrs_kept"] - before["utrs_kept"] >= len(preps)
    for i, r in enumerate(res):
        para = to_parameters(r)
        assert para.K == int(f[f"u{i}_res_K"]), (name, i)
        assert np.array_equal(para.alpha_arr, f[f"u{i}_res_alpha_arr"])
        assert np.array_equal(para.beta_arr, f[f"u{i}_res_beta_arr"])
        assert np.allclose(para.ws, f[f"u{i}_res_ws"], rtol=1e-9, atol=1e-13)
        assert para.bic == pytest.approx(float(f[f"u{i}_res_bic"]), rel=1e-10)
        assert np.array_equal(para.label_arr, f[f"u{i}_res_label_arr"])
        assert np.allclose(para.lb_arr, f[f"u{i}_res_lb_arr"], rtol=1e-10)


def test_stream_followers_do_not_change_results():
    """Strictly serial (spec_depth 1) against 3 and 8 UTRs of a stream per EM call, on chunks whose UTRs prune, re-run
    (n_max_apa 3 with re_run_mode) and end clean: identical fits bit for bit, identical labels, job counts and final
    generator state; also for two streams in flight (run_streams)."""
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    chunks = []
    for kcap, reads, n_utr, base in ((3, 400, 14, 7700), (6, 600, 10, 7800)):
        kw = dict(n_max_apa=kcap, n_min_apa=1)
        chunks.append([prepare_utr(df, gene_info_str=g, **kw)
                       for g, df, _ in (synth_utr(i, reads, k_cap=5, base_seed=base) for i in range(n_utr))])
    eng = Engine(device=0)
    keep, keep_pp = Engine.spec_depth, Engine.pingpong_min_streams
    got = {}
    try:
        for depth in (1, 3, 8):
            Engine.spec_depth = depth
            before = dict(Engine.spec_stats)
            runs = []
            for preps in chunks:
                rs = np.random.RandomState(5)
                res = eng.run(preps, rng_mode="reference", rs=rs, re_run_mode=True)
                runs.append((res, rs.get_state()[1].copy(), rs.get_state()[2]))
            streams = eng.run_streams([(preps, 9) for preps in chunks], re_run_mode=True)
            got[depth] = (runs, streams, {k: Engine.spec_stats[k] - before[k] for k in before})
        # the two-halves schedule of many streams (a worker thread runs the EM call of one half while the other half's
        # results are taken in), forced onto these two streams
        Engine.spec_depth, Engine.pingpong_min_streams = 3, 2
        halves = eng.run_streams([(preps, 9) for preps in chunks], re_run_mode=True)
    finally:
        Engine.spec_depth, Engine.pingpong_min_streams = keep, keep_pp
    assert got[1][2]["utrs_discarded"] == 0 and got[1][2]["predicted"] == 0
    # some predictions held (fewer calls than UTRs + re-runs), some did not (followers drawn again)
    assert got[8][2]["utrs_discarded"] > 0 and got[8][2]["calls"] < got[1][2]["calls"]
    assert 0 < got[8][2]["predicted_right"] < got[8][2]["predicted"]

    def same(r1, r2):
        assert r1.n_jobs == r2.n_jobs
        a, b = r1.fit, r2.fit
        assert a.K == b.K and a.bic == b.bic
        for x, y in ((a.a_idx, b.a_idx), (a.b_idx, b.b_idx), (a.ws, b.ws), (a.lb, b.lb), (r1.labels_bin, r2.labels_bin)):
            assert np.array_equal(x, y)

    ks = set()
    for depth in (3, 8):
        for (res1, key1, pos1), (resd, keyd, posd) in zip(got[1][0], got[depth][0]):
            assert np.array_equal(key1, keyd) and pos1 == posd
            for r1, r2 in zip(res1, resd):
                same(r1, r2)
                ks.add((r1.fit.K, r1.n_jobs))
        for s1, sd in zip(got[1][1], got[depth][1]):
            for r1, r2 in zip(s1, sd):
                same(r1, r2)
    for s1, sd in zip(got[1][1], halves):
        for r1, r2 in zip(s1, sd):
            same(r1, r2)
    assert len({nj for _k, nj in ks}) >= 3      # clean ends, prunes (+1 job) and re-runs (another sweep) are all present


# ---------------------------------------------------------------- CLI end to end vs the committed example outputs
# (fixture, UTR) -> (ws abs tolerance, bic rel tolerance) where the reference at its current code reproduces the
# committed K / alpha / beta / labels exactly (measured: tests/golden/trace_*.npz res_* vs fixture_*.npz gold_*);
# ws / bic of the committed files come from an older init path and a loose stopping rule
GOLDEN_EXACT = {("chr17", 0): (5e-5, 1e-5), ("toy", 0): (2e-3, 1.2e-2)}


@pytest.mark.parametrize("name", FIXTURES)
def test_cli_on_example_chunks_vs_committed_goldens(tmp_path, name):
    from click.testing import CliRunner
    from scape.cli import cli
    from scape_amd import safe_pickle
    f = load_npz(f"fixture_{name}.npz")
    out = tmp_path / "out"
    (out / "pkl_input").mkdir(parents=True)
    chunk = out / "pkl_input" / f"{name}.100.1.1.input.pkl"
    with open(chunk, "wb") as fh:                       # prepare_input's append-mode layout
        for i in range(int(f["n_utr"])):
            gene, df = utr_df(f, i)
            for c in ("junction", "seg1_en", "seg2_en"):
                df[c] = 0
            pickle.dump((gene, df), fh)
    (out / "parameters.toml").write_text(
        "n_max_apa = 5\nn_min_apa = 1\nmin_LA = 20\nmax_LA = 150\nmu_f = 300\nsigma_f = 50\nmin_pa_gap = 100\n"
        "max_beta = 70\ntheta_step = 9\nbeta_step = 5\nmin_ws = 0.05\nmax_unif_ws = 0.15\nre_run_mode = true\n"
        "fixed_run_mode = false\nutr_file = \"x.csv\"\nchunksize = 100\noutput_dir = \"ignored\"\n")
    r = CliRunner().invoke(cli, ["infer_pa", "--pkl_input_file", str(chunk), "--output_dir", str(out)])
    assert r.exit_code == 0, r.output + repr(r.exception)
    res_file = out / "pkl_output" / f"{name}.100.1.1.res.pkl"
    with open(res_file, "rb") as fh:
        got = []
        while True:
            try:
                got.append(pickle.load(fh))
            except EOFError:
                break
    assert len(got) == int(f["n_utr"])
    assert [type(g).__module__ + "." + type(g).__name__ for g in got] == ["scape.apa_core.Parameters"] * len(got)
    assert len(safe_pickle.load_all(str(res_file))) == len(got)
    for i, para in enumerate(got):
        tag = (name, i)
        assert para.gene_info_str == str(f[f"u{i}_gene_info_str"])
        assert para.K == int(f[f"u{i}_gold_K"]) and para.L == int(f[f"u{i}_gold_L"]), tag
        if tag in GOLDEN_EXACT:
            # the current reference code reproduces its committed output here: K, alpha, beta and EVERY label
            ws_atol, bic_rel = GOLDEN_EXACT[tag]
            assert np.array_equal(para.alpha_arr, f[f"u{i}_gold_alpha_arr"]), tag
            assert np.array_equal(para.beta_arr, f[f"u{i}_gold_beta_arr"]), tag
            assert np.array_equal(para.label_arr, f[f"u{i}_gold_label_arr"]), tag
            assert np.allclose(para.ws, f[f"u{i}_gold_ws"], atol=ws_atol), tag
            assert para.bic == pytest.approx(float(f[f"u{i}_gold_bic"]), rel=bic_rel), tag
        else:
            # chr17 UTR 2 and chr19: the CURRENT reference itself no longer reproduces these committed files
            # (trace_chr17 u1: alpha 832 vs 841, beta [45,40,30] vs [10,45,30], 99.5 % of the labels; trace_chr19 u0:
            # beta [5,15] vs [5,10], 99.8 % of the labels) - they predate the code, so they
            # stay a coarse check; the exact check for these two UTRs is test_full_pipeline_reference_stream
            assert np.all(np.abs(para.alpha_arr - f[f"u{i}_gold_alpha_arr"]) <= 9), tag
            assert np.mean(para.label_arr == f[f"u{i}_gold_label_arr"]) >= 0.99, tag
            assert np.allclose(para.ws, f[f"u{i}_gold_ws"], atol=2e-2), tag
            assert para.bic == pytest.approx(float(f[f"u{i}_gold_bic"]), rel=2e-2), tag
        assert np.array_equal(para.cb_id_arr, f[f"u{i}_gold_cb_id_arr"])
        assert np.array_equal(para.readID_arr, f[f"u{i}_gold_readID_arr"])
        assert len(para.label_arr) == len(para.readID_arr)
        assert isinstance(para.bic, np.float64) and isinstance(para.lb_arr, list)
        assert para.label_arr.dtype == np.int64 and para.ws.dtype == np.float64


# ---------------------------------------------------------------- batched (per_utr RNG) mode vs oracle
def _oracle_per_utr(oracle, chunk, kw, seed):
    out = []
    for i, (gene, df) in enumerate(chunk):
        np.random.seed(seed + i)
        res, _m = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values, **kw)
        out.append(res)
    return out


def test_batched_mode_vs_oracle_config1(oracle):
    """BASELINE config #2 shape (500 reads, K <= 5), a 48-UTR slice; every UTR checked."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_chunk
    chunk = synth_chunk(48, 500, k_cap=5, base_seed=100)
    kw = dict(n_max_apa=5, n_min_apa=1)
    preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df in chunk]
    res = Engine(device=0).run(preps, rng_mode="per_utr", seed=7, re_run_mode=True)
    want = _oracle_per_utr(oracle, chunk, dict(kw, re_run_mode=True), 7)
    n_same = 0
    for r, w in zip(res, want):
        para = to_parameters(r)
        same = (para.K == w.K and np.array_equal(para.alpha_arr, w.alpha_arr)
                and np.array_equal(para.beta_arr, w.beta_arr) and np.array_equal(para.label_arr, w.label_arr))
        n_same += same
        if same:
            assert np.allclose(para.ws, w.ws, rtol=1e-4, atol=1e-9)          # north-star tolerance
            assert para.bic == pytest.approx(w.bic, rel=1e-8)
    assert n_same == len(chunk), f"{n_same}/{len(chunk)} UTRs identical to the oracle"


def test_edge_cases_vs_oracle(oracle, hip_ctx):
    """tiny UTRs, all-pa reads, r-known reads, K=0 after pruning everything, duplicated reads."""
    from scape_amd.engine import Engine, HipBatch, _Job
    from scape_amd.host import prepare_utr
    rng = np.random.default_rng(9)

    def mk(x, l, r=None, pa=None):
        n = len(x)
        return pd.DataFrame({"x": np.asarray(x, dtype=np.int64), "l": np.asarray(l, dtype=np.int64),
                             "r": np.full(n, np.nan) if r is None else r,
                             "pa": np.full(n, np.nan) if pa is None else pa,
                             "cb_id": np.arange(n), "read_id": np.arange(n)})
    cases = [
        ("one_bin", mk([400] * 120, [98] * 120)),
        ("two_bins", mk([400] * 60 + [900] * 60, [98] * 120)),
        ("all_pa", mk(rng.integers(300, 600, 150), rng.integers(31, 133, 150), pa=rng.integers(700, 760, 150).astype(float))),
        ("r_known", mk(rng.integers(300, 900, 200), rng.integers(31, 133, 200), r=rng.integers(5, 140, 200).astype(float))),
        ("long_utr", mk(rng.integers(0, 9000, 300), rng.integers(31, 133, 300))),
    ]
    kw = dict(n_max_apa=3, n_min_apa=1)
    preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df in cases]
    res = Engine(device=0).run(preps, rng_mode="per_utr", seed=3, re_run_mode=True)
    for i, ((g, df), r) in enumerate(zip(cases, res)):
        np.random.seed(3 + i)
        w, _m = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values, re_run_mode=True, **kw)
        q = r.prep
        assert r.fit.K == w.K, g
        assert np.array_equal(np.rint(q.theta[r.fit.a_idx]).astype(int), w.alpha_arr), g
        assert np.array_equal(q.betas[r.fit.b_idx], w.beta_arr), g
        assert np.allclose(r.fit.ws, w.ws, rtol=1e-8, atol=1e-12), g
        assert np.array_equal(r.labels_bin.astype(np.int64)[q.idx], w.label_arr), g
    # K = 0 fixed-inference job (rm_component can drop every component, apa_core.py:832-844)
    batch = HipBatch(hip_ctx, preps[:1])
    batch.build()
    ft = batch.em([_Job(0, 0, True, np.zeros(0, np.int32), np.zeros(0, np.int32), np.array([0.15]), np.zeros(50, np.int8))])[0]
    assert ft.K == 0 and np.allclose(ft.ws, [0.15]) and len(ft.lb) >= 2
    lab = batch.labels([(0, ft)])[0]
    assert np.all(lab == 0)
    batch.free()


def test_c_abi_error_paths(hip_ctx):
    from scape_amd import _lib
    from scape_amd.engine import HipBatch, _Job
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_chunk
    g, df = synth_chunk(1, 200, base_seed=5)[0]
    q = prepare_utr(df, gene_info_str=g)
    batch = HipBatch(hip_ctx, [q])
    good = _Job(0, 2, False, np.array([3, 9], np.int32), np.array([1, 2], np.int32), np.array([0.4, 0.5, 0.1]), np.zeros(50, np.int8))
    with pytest.raises(_lib.ScapeHipError, match="batch_build"):
        batch.em([good])                                   # EM before build
    batch.build()
    bad = [
        _Job(1, 2, False, good.a_idx, good.b_idx, good.ws, good.k_arr),                       # UTR index
        _Job(0, 2, False, np.array([3, q.T], np.int32), good.b_idx, good.ws, good.k_arr),     # alpha index
        _Job(0, 2, False, good.a_idx, np.array([1, 13], np.int32), good.ws, good.k_arr),      # beta index
        _Job(0, 2, False, np.array([9, 3], np.int32), good.b_idx, good.ws, good.k_arr),       # unsorted alpha
        _Job(0, 2, False, good.a_idx, good.b_idx, good.ws, np.full(50, 2, np.int8)),          # k_arr entry
    ]
    for j in bad:
        with pytest.raises(_lib.ScapeHipError):
            batch.em([j])
    # more non-fixed jobs on one UTR than the M-step's per-tile job table holds (SCAPE_MAX_JOBS_PER_UTR = 1024)
    with pytest.raises(_lib.ScapeHipError, match="more than 1024"):
        batch.em([good] * 1025)
    assert len(batch.em([good] * 1024)) == 1024             # the limit itself is fine
    assert batch.em([good])[0].K == 2                       # context still usable after errors
    with pytest.raises(_lib.ScapeHipError):
        _lib.Context(device=99)
    # lb_arr rows left on the device and fetched for chosen jobs (scape_hip_batch_em_fetch_lb, ABI 3)
    from scape_amd.engine import pack_jobs
    jobs = [good, _Job(0, 1, False, np.array([5], np.int32), np.array([2], np.int32), np.array([0.7, 0.3]), np.zeros(50, np.int8)), good]
    pj = pack_jobs(jobs)
    full = [a.copy() for a in batch.em_packed(pj)]
    lean = batch.em_packed(pj, want_lb=False)
    assert lean[5] is None
    for x, y in zip(full[:5], lean[:5]):
        assert np.array_equal(x, y)
    rows = batch.em_fetch_lb([2, 1])
    assert np.array_equal(rows[0], full[5][2]) and np.array_equal(rows[1], full[5][1])
    f0, f1 = HipBatch.fit_at(pj, full, 1), HipBatch.fit_at(pj, lean, 1, rows[1])
    assert np.array_equal(f0.lb, f1.lb) and len(f0.lb) == full[4][1] >= 2
    with pytest.raises(_lib.ScapeHipError, match="out of range"):
        batch.em_fetch_lb([3])
    batch.free()
    with pytest.raises(_lib.ScapeHipError):
        batch.em_fetch_lb([0])                              # nothing to fetch from after batch_free


def test_handle_rejects_concurrent_calls():
    """Calls on one handle are not re-entrant (scape_hip.h): a second host thread that enters while a
    call is in flight gets an error, the call in flight is unharmed; a private handle per thread works."""
    import ctypes
    import threading
    from scape_amd import _lib
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_chunk
    preps = [prepare_utr(df, gene_info_str=g, n_max_apa=6) for g, df in synth_chunk(24, 1500, k_cap=6, base_seed=77)]
    seeds = list(range(24))
    want = Engine().run(preps, rng_mode="per_utr", seeds=seeds, re_run_mode=False)
    eng = Engine(own_context=True)
    assert eng.ctx is not _lib.default_context(None)
    got, errs, done = [], [], threading.Event()

    def work():
        try:
            for _ in range(3):
                got.append(eng.run(preps, rng_mode="per_utr", seeds=seeds, re_run_mode=False))
        finally:
            done.set()
    th = threading.Thread(target=work)
    th.start()
    a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    while not done.is_set():
        if eng.ctx.lib.scape_hip_batch_bytes(eng.ctx.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)):
            errs.append(_lib.last_error())
    th.join()
    assert errs and all("in use by another host thread" in e for e in errs)
    assert len(got) == 3
    for res in got:
        for r, w in zip(res, want):
            assert r.fit.K == w.fit.K and np.array_equal(r.fit.a_idx, w.fit.a_idx) and np.array_equal(r.fit.ws, w.fit.ws)
            assert np.array_equal(r.labels_bin, w.labels_bin)
    eng.ctx.close()


# ---------------------------------------------------------------- properties at the headline shape
def test_properties_headline_shape():
    """2k reads x K<=10 (BASELINE config #3 shape), 24 UTRs: determinism and model invariants."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    kw = dict(n_max_apa=10, n_min_apa=1)
    items = [synth_utr(i, 2000, k_cap=10, base_seed=4242) for i in range(24)]
    preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df, _t in items]
    eng = Engine(device=0)
    r1 = eng.run(preps, rng_mode="per_utr", seed=11, re_run_mode=False)
    hits = 0
    for i, (a, (g, df, truth)) in enumerate(zip(r1, items)):
        pa = to_parameters(a)
        q = a.prep
        assert 1 <= pa.K <= 10 and len(pa.ws) == pa.K + 1 and len(pa.label_arr) == len(df)
        assert np.all(np.diff(pa.alpha_arr) >= 0) and np.all(np.isin(pa.alpha_arr, q.theta))
        assert np.all(np.isin(pa.beta_arr, q.betas))
        assert np.all(pa.ws >= 0) and pa.ws[-1] <= 0.15 + 1e-12 and abs(pa.ws.sum() - 1) < 1e-9
        assert np.isfinite(pa.bic) and np.all(np.isfinite(pa.lb_arr)) and 2 <= len(pa.lb_arr) <= 50
        assert pa.label_arr.min() >= 0 and pa.label_arr.max() <= pa.K
        # the dominant true site is recovered within 3 beta
        k = int(np.argmax(truth["ws"]))
        hits += bool(np.any(np.abs(pa.alpha_arr - truth["alphas"][k]) <= 3 * truth["betas"][k] + 9))
    assert hits >= 22
    # run-to-run determinism: identical bits
    r3 = eng.run(preps, rng_mode="per_utr", seed=11, re_run_mode=False)
    for a, b in zip(r1, r3):
        assert a.fit.K == b.fit.K and np.array_equal(a.fit.a_idx, b.fit.a_idx) and np.array_equal(a.fit.b_idx, b.fit.b_idx)
        assert np.array_equal(a.fit.ws, b.fit.ws) and a.fit.bic == b.fit.bic and np.array_equal(a.labels_bin, b.labels_bin)


def test_cli_fixed_run_mode_vs_oracle(tmp_path, oracle):
    """--pre_para_pkl_file (reference apa_core.py:94-99, :999-1017, :883-928): K and grids come from the first
    Parameters of a result file; same RNG stream as the reference -> compare with the oracle's fixed_run."""
    from types import SimpleNamespace
    from click.testing import CliRunner
    from scape.apa_core import Parameters
    from scape.cli import cli
    from scape_amd.synth import synth_chunk
    chunk = synth_chunk(3, 400, k_cap=3, base_seed=500, pa_rate=0.03)
    out = tmp_path / "out"
    (out / "pkl_input").mkdir(parents=True)
    inp = out / "pkl_input" / "fx.100.1.1.input.pkl"
    with open(inp, "wb") as fh:
        for g, df in chunk:
            pickle.dump((g, df), fh)
    (out / "parameters.toml").write_text("n_max_apa = 4\nre_run_mode = true\n")
    pre = Parameters(title="Final Result", alpha_arr=np.array([700, 1600]), beta_arr=np.array([20.0, 40.0]),
                     ws=np.array([0.5, 0.45, 0.05]), L=2500)
    pre_file = tmp_path / "pre.res.pkl"
    with open(pre_file, "wb") as fh:
        pickle.dump(pre, fh)
        pickle.dump(pre, fh)                       # only the first object is used (apa_core.py:1002-1003)
    r = CliRunner().invoke(cli, ["infer_pa", "--pkl_input_file", str(inp), "--output_dir", str(out),
                                 "--pre_para_pkl_file", str(pre_file)])
    assert r.exit_code == 0, r.output + repr(r.exception)
    toml = (out / "parameters.toml").read_text()
    assert "fixed_run_mode = true" in toml and "pre_para_pkl_file" in toml      # rewritten like tomli_w.dump (:98-99)
    got = []
    with open(out / "pkl_output" / "fx.100.1.1.res.pkl", "rb") as fh:
        while True:
            try:
                got.append(pickle.load(fh))
            except EOFError:
                break
    np.random.seed(1)
    pre_ns = SimpleNamespace(alpha_arr=pre.alpha_arr, beta_arr=pre.beta_arr, ws=pre.ws, L=2500, K=2)
    for (g, df), para in zip(chunk, got):
        want, _m = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values,
                                        pre_para=pre_ns, n_max_apa=4)
        assert para.title == "Final Result (subsample run)" and para.K == 2 and para.L == max(2500, want and para.L)
        assert np.array_equal(para.alpha_arr, want.alpha_arr) and np.array_equal(para.beta_arr, want.beta_arr)
        assert np.allclose(para.ws, want.ws, rtol=1e-9, atol=1e-13)
        assert para.bic == pytest.approx(want.bic, rel=1e-10)
        assert np.array_equal(para.label_arr, want.label_arr)


def test_infer_pa_all_equals_per_file_runs(tmp_path):
    """The chunk-directory driver keeps each file's own reference RNG stream while sharing GPU launches:
    every <stem>.res.pkl equals the one `scape infer_pa` writes for that file alone."""
    from click.testing import CliRunner
    from scape.cli import cli
    from scape_amd.synth import synth_chunk
    out = tmp_path / "out"
    (out / "pkl_input").mkdir(parents=True)
    (out / "parameters.toml").write_text("n_max_apa = 3\nre_run_mode = true\n")
    files = []
    for fi, n in enumerate((3, 1, 4)):
        f = out / "pkl_input" / f"s{fi}.100.3.{fi + 1}.input.pkl"
        with open(f, "wb") as fh:
            for g, df in synth_chunk(n, 250 + 50 * fi, k_cap=3, base_seed=900 + 10 * fi, pa_rate=0.04):
                pickle.dump((g, df), fh)
        files.append(f)
    (out / "pkl_input" / "s9.tmp.100.3.9.input.pkl").write_bytes(b"")        # incomplete chunk: skipped
    r = CliRunner().invoke(cli, ["infer_pa_all", "--output_dir", str(out)])
    assert r.exit_code == 0, r.output + repr(r.exception)

    def load(p):
        res = []
        with open(p, "rb") as fh:
            while True:
                try:
                    res.append(pickle.load(fh))
                except EOFError:
                    return res
    together = {f.name: load(out / "pkl_output" / (f.name[:-10] + ".res.pkl")) for f in files}
    assert not (out / "pkl_output" / "s9.tmp.100.3.9.res.pkl").exists()
    for f in files:
        r = CliRunner().invoke(cli, ["infer_pa", "--pkl_input_file", str(f), "--output_dir", str(out)])
        assert r.exit_code == 0, r.output + repr(r.exception)
        alone = load(out / "pkl_output" / (f.name[:-10] + ".res.pkl"))
        assert len(alone) == len(together[f.name])
        for a, b in zip(alone, together[f.name]):
            assert a.gene_info_str == b.gene_info_str and a.K == b.K
            assert np.array_equal(a.alpha_arr, b.alpha_arr) and np.array_equal(a.beta_arr, b.beta_arr)
            assert np.array_equal(a.ws, b.ws) and a.bic == b.bic and np.array_equal(a.label_arr, b.label_arr)
            assert np.array_equal(a.lb_arr, b.lb_arr)


def test_config2_full_size_every_utr_vs_cpu_port(oracle):
    """BASELINE config #2 at its full size: 1,000 UTRs x 500 reads, K = 1..5 x 10 restarts (50,000 EM jobs).
    One resident GPU batch; the CPU port (oracle) runs the same 50 jobs of EVERY UTR on the host threads and goes
    through the reference's selection and pruning rule; all 1,000 pA calls must agree (K, alpha, beta; ws to 1e-4
    where no re-fit is involved - the bench's own parity check)."""
    import bench
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    U, kw = 1000, dict(n_max_apa=5, n_min_apa=1)
    preps = []
    for i in range(U):
        g, df, _ = synth_utr(i, 500, k_cap=5, base_seed=777)
        preps.append(prepare_utr(df, gene_info_str=g, **kw))
    eng = Engine()
    plan = eng.plan(preps, [(777 + i) % 2 ** 32 for i in range(U)])
    assert len(plan["main"]) == 50 * U
    res = eng.process(eng.load(preps), preps, plan, re_run_mode=False)
    cores = min(16, os.cpu_count() or 1)
    cb, outs = bench.cpu_baseline(preps, plan, 1e9, cores)          # target time = infinity: every UTR
    assert len(outs) == U
    assert bench.parity_count(preps, plan, res, outs) == U
    for (fit, lab, nj), q in zip(res, preps):
        assert 1 <= fit.K <= 5 and len(lab) == q.N and abs(fit.ws.sum() - 1) < 1e-12


def _assert_parameters_equal_oracle(paras, outs, ws_rtol=1e-6):
    """Every field of the final Parameters the reference's consumers read, GPU vs oracle.pool:
    K, alpha, beta and ALL per-read labels exact; ws (also the re-fitted ws of a pruned UTR), bic and the lower-bound
    trace of the final fit (length exact = same number of EM rounds)."""
    for a, o in zip(paras, outs):
        g = o["gene"]
        assert a.gene_info_str == g
        assert a.K == o["K"], (g, a.K, o["K"])
        assert np.array_equal(a.alpha_arr, o["alpha"]), (g, a.alpha_arr, o["alpha"])
        assert np.array_equal(a.beta_arr, o["beta"]), (g, a.beta_arr, o["beta"])
        assert np.array_equal(a.label_arr, o["labels"]), (g, int((a.label_arr != o["labels"]).sum()))
        assert np.allclose(a.ws, o["ws"], rtol=ws_rtol, atol=1e-10), (g, a.ws, o["ws"])
        assert a.bic == pytest.approx(o["bic"], rel=1e-9), (g, a.bic, o["bic"])
        assert len(a.lb_arr) == len(o["lb"]), (g, len(a.lb_arr), len(o["lb"]))
        assert np.allclose(np.array(a.lb_arr), o["lb"], rtol=1e-9), g


def test_headline_shape_every_utr_vs_oracle(oracle):
    """BASELINE config #3 (the headline) shape: 64 UTRs x 2,000 reads of the bench's own stream, K = 10..1 x 10 restarts
    (6,400 EM jobs + prune re-fits + re-run sweeps) through the batched product path with re_run_mode = True (the
    reference's default, input_processor.py:97-112).  The oracle fits EVERY UTR in full (subsample_run: selection,
    rm_component re-fit, get_label, re-run loop) from the same per-UTR seed; all final fields must agree - K, alpha, beta and
    every per-read label exactly, ws / bic / lb_arr to rounding."""
    from oracle.pool import fit_many
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    U, kw, seed = 64, dict(n_max_apa=10, n_min_apa=1), 20250225
    preps = []
    for i in range(U):
        g, df, _ = synth_utr(i, 2000, k_cap=10, base_seed=seed)      # the bench's own stream
        preps.append(prepare_utr(df, gene_info_str=g, **kw))
    eng = Engine()
    res = eng.run(preps, rng_mode="per_utr", seed=seed, re_run_mode=True)
    outs = fit_many([(i, 2000, 10, seed, (seed + i) % 2 ** 32, True, kw) for i in range(U)])
    _assert_parameters_equal_oracle([to_parameters(r) for r in res], outs)
    pruned = sum(o["n_calls"] % 10 == 1 for o in outs)           # sweeps are multiples of 10 calls, a prune re-fit adds one
    assert pruned >= U // 4, pruned            # the re-fitted ws of pruned UTRs is part of what was compared
    for r, q in zip(res, preps):
        assert 1 <= r.fit.K and len(r.labels_bin) == q.N and abs(r.fit.ws.sum() - 1) < 1e-12


def test_small_call_shapes_give_identical_bits(hip_ctx):
    """Calls with few jobs / few live tensor tiles run low-latency kernel shapes: the E-step with 4 wavefronts per job
    that split the components of a bin (k2_estep_cs), the M-step with a tile's jobs cut into passes of 16 for separate
    workgroups and the tiles dealt over all XCDs; large calls run one wavefront per job, one workgroup per tile and 64
    jobs.  The same jobs through every combination must agree bit for bit (a UTR's result must not depend on what else
    shares the launch) - K <= 3 (the 4-column E-step kernel), K = 1..10 (12-column kernel, every exact variant),
    K = 12..14 (16-column kernel)."""
    from scape_amd.engine import Engine, HipBatch
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    for kcap, kmin, reads, n_utr in ((3, 1, 300, 3), (10, 1, 1500, 2), (14, 12, 700, 1)):
        kw = dict(n_max_apa=kcap, n_min_apa=kmin)
        preps = [prepare_utr(df, gene_info_str=g, **kw)
                 for g, df, _ in (synth_utr(i, reads, k_cap=5, base_seed=4100 + kcap) for i in range(n_utr))]
        batch = HipBatch(hip_ctx, preps)
        batch.build()
        plan = Engine.plan(preps, [11 + i for i in range(n_utr)])
        outs = {}
        for name, wide, split in (("throughput", "0", "0"), ("low_latency", "1000000", "1000000"),
                                  ("wide_estep_only", "1000000", "0"), ("split_mstep_only", "0", "1000000")):
            os.environ["SCAPE_HIP_WIDE_MAXJOBS"], os.environ["SCAPE_HIP_SPLIT_MAXTILES"] = wide, split
            try:
                outs[name] = [a.copy() for a in batch.em_packed(plan["main"])]
            finally:
                del os.environ["SCAPE_HIP_WIDE_MAXJOBS"], os.environ["SCAPE_HIP_SPLIT_MAXTILES"]
        ref = outs.pop("throughput")
        nlb = ref[4]
        assert nlb.min() >= 2
        for name, got in outs.items():
            for x, y, what in zip(ref, got, ("alpha", "beta", "ws", "bic", "n_lb", "lb")):
                if what == "lb":
                    mask = np.arange(x.shape[1])[None, :] < nlb[:, None]
                    assert np.array_equal(x[mask], y[mask]), (kcap, name, what)
                else:
                    assert np.array_equal(x, y), (kcap, name, what)
        batch.free()


def test_mstep_byte_tally_is_consistent(hip_ctx):
    """scape_hip_em_traffic: the M-step's own byte count of a sweep is positive, at most one pass over the
    tensor per launch, and the v bytes counted once never exceed the v bytes requested."""
    from scape_amd.engine import Engine, HipBatch
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    kw = dict(n_max_apa=5, n_min_apa=1)
    preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df, _ in (synth_utr(i, 800, k_cap=5, base_seed=31) for i in range(12))]
    batch = HipBatch(hip_ctx, preps)
    batch.build()
    plan = Engine.plan(preps, list(range(12)))
    batch.em_packed(plan["main"])
    tr = batch.em_traffic()
    tensor = 8 * sum(q.T * len(q.betas) * ((q.N + 15) // 16 * 16) for q in preps)
    assert 1 <= tr["launches"] <= 50
    assert 0 < tr["tensor_bytes"] <= tensor * tr["launches"]
    assert 0 < tr["v_bytes_unique"] <= tr["v_bytes_requested"]
    rounds, slab, _z = batch.em_counters()
    assert 8 * slab >= tr["tensor_bytes"]            # job-at-a-time reads at least what the shared stream reads
    batch.free()


def test_infer_files_reference_groups_equal_single_file_runs(tmp_path):
    """More chunk files than the first group holds (16): files of different groups, waves and stream positions
    still get exactly the result of running each file on its own."""
    from scape_amd.apa_core import infer, infer_files
    from scape_amd.synth import synth_chunk
    out = tmp_path / "o"
    (out / "pkl_input").mkdir(parents=True)
    kw = dict(n_max_apa=2, n_min_apa=1, re_run_mode=True)
    files = []
    for fi in range(21):
        f = out / "pkl_input" / f"q{fi:02d}.100.21.{fi + 1}.input.pkl"
        with open(f, "wb") as fh:
            for g, df in synth_chunk(1 + fi % 3, 180 + 10 * (fi % 4), k_cap=2, base_seed=2000 + fi, pa_rate=0.05):
                pickle.dump((g, df), fh)
        files.append(str(f))
    written = infer_files(files, str(out), files_in_flight=4, **kw)
    assert len(written) == 21

    def load(p):
        res = []
        with open(p, "rb") as fh:
            while True:
                try:
                    res.append(pickle.load(fh))
                except EOFError:
                    return res
    for f, w in zip(files, written):
        assert os.path.basename(w) == os.path.basename(f)[:-10] + ".res.pkl"
        alone = infer(f, str(tmp_path / "alone.res.pkl"), **kw)
        got = load(w)
        assert len(got) == len(alone) > 0
        for a, b in zip(got, alone):
            assert a.gene_info_str == b.gene_info_str and a.K == b.K and a.bic == b.bic
            assert np.array_equal(a.alpha_arr, b.alpha_arr) and np.array_equal(a.ws, b.ws)
            assert np.array_equal(a.label_arr, b.label_arr) and np.array_equal(a.lb_arr, b.lb_arr)


def test_infer_pa_all_two_worker_processes(tmp_path):
    """--gpus 2: chunk files sharded over two worker processes (each with its own prep pool and library handle;
    on a one-GPU box both use device 0).  Every file's result equals the single-process run."""
    from click.testing import CliRunner
    from scape.cli import cli
    from scape_amd.synth import synth_chunk
    outs = []
    for gpus in (1, 2):
        out = tmp_path / f"g{gpus}"
        (out / "pkl_input").mkdir(parents=True)
        (out / "parameters.toml").write_text('n_max_apa = 3\nre_run_mode = true\nrng_mode = "per_utr"\nseed = 11\n')
        for fi, n in enumerate((6, 2, 5, 3)):
            with open(out / "pkl_input" / f"m{fi}.100.4.{fi + 1}.input.pkl", "wb") as fh:
                for g, df in synth_chunk(n, 260 + 30 * fi, k_cap=3, base_seed=500 + 10 * fi, pa_rate=0.04):
                    pickle.dump((g, df), fh)
        r = CliRunner().invoke(cli, ["infer_pa_all", "--output_dir", str(out), "--gpus", str(gpus)])
        assert r.exit_code == 0, r.output + repr(r.exception)
        outs.append(out)

    def load(p):
        res = []
        with open(p, "rb") as fh:
            while True:
                try:
                    res.append(pickle.load(fh))
                except EOFError:
                    return res
    names = sorted(os.listdir(outs[0] / "pkl_output"))
    assert names == sorted(os.listdir(outs[1] / "pkl_output")) and len(names) == 4
    for n in names:
        a, b = load(outs[0] / "pkl_output" / n), load(outs[1] / "pkl_output" / n)
        assert len(a) == len(b) > 0
        for x, y in zip(a, b):
            assert x.gene_info_str == y.gene_info_str and x.K == y.K and x.bic == y.bic
            assert np.array_equal(x.alpha_arr, y.alpha_arr) and np.array_equal(x.ws, y.ws)
            assert np.array_equal(x.label_arr, y.label_arr)


def test_pipelined_per_utr_mode_equals_engine_run(tmp_path, oracle):
    """rng_mode='per_utr' through the host pipeline (prep processes -> native planner -> GPU thread ->
    writer, batches that straddle file boundaries) writes exactly what Engine.run gives for each file on
    its own with seeds seed + j, and those equal the oracle run from the same per-UTR RandomState."""
    from scape_amd.apa_core import infer_files, read_input_chunk, to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_chunk
    out = tmp_path / "out"
    (out / "pkl_input").mkdir(parents=True)
    kw = dict(n_max_apa=4, n_min_apa=1, re_run_mode=True)
    files = []
    for fi, n in enumerate((5, 0, 7, 3)):
        f = out / "pkl_input" / f"p{fi}.100.4.{fi + 1}.input.pkl"
        with open(f, "wb") as fh:
            for g, df in synth_chunk(n, 300 + 40 * fi, k_cap=4, base_seed=300 + 10 * fi, pa_rate=0.04):
                pickle.dump((g, df), fh)
        files.append(str(f))
    import scape_amd.pipeline as pl
    orig = pl.run_pipeline
    try:
        pl.run_pipeline = lambda *a, **k: orig(*a, **{**k, "batch_utrs": 4})      # force file-straddling batches
        stats = {}
        written = infer_files(files, str(out), rng_mode="per_utr", seed=7, stats=stats, **kw)
    finally:
        pl.run_pipeline = orig
    assert [os.path.basename(w) for w in written] == [os.path.basename(f)[:-10] + ".res.pkl" for f in files]
    assert stats["n_utr"] == 15 and stats["n_batch"] == 4
    eng = Engine()
    for f, w in zip(files, written):
        utrs = list(read_input_chunk(f))
        got = []
        with open(w, "rb") as fh:
            while True:
                try:
                    got.append(pickle.load(fh))
                except EOFError:
                    break
        assert len(got) == len(utrs)
        if not utrs:
            continue
        preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df in utrs]
        want = [to_parameters(r) for r in eng.run(preps, rng_mode="per_utr", seed=7, re_run_mode=True)]
        for j, (a, b, (g, df)) in enumerate(zip(got, want, utrs)):
            assert a.gene_info_str == b.gene_info_str == g and a.K == b.K
            assert np.array_equal(a.alpha_arr, b.alpha_arr) and np.array_equal(a.beta_arr, b.beta_arr)
            assert np.array_equal(a.ws, b.ws) and a.bic == b.bic and np.array_equal(a.label_arr, b.label_arr)
            np.random.seed(7 + j)                                     # the oracle draws from the global stream
            ref, _m = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values, **kw)
            assert a.K == ref.K and np.array_equal(a.alpha_arr, ref.alpha_arr) and np.array_equal(a.beta_arr, ref.beta_arr)
            assert np.allclose(a.ws, ref.ws, rtol=1e-9, atol=1e-13) and a.bic == pytest.approx(ref.bic, rel=1e-10)
            assert np.array_equal(a.label_arr, ref.label_arr)


@pytest.mark.parametrize("reads,kcap,n", [(10000, 10, 16), (5000, 12, 16)])
def test_deep_pileup_shapes_vs_oracle(oracle, reads, kcap, n):
    """BASELINE configs #4 (10k reads, K<=10) and #5 (5k reads, K=1..12) shapes - the 16-column E-step variant and
    2,154-bin tiles are the product path there.  16 UTRs per shape through the batched path (re_run_mode on) against
    the oracle's full fit of every UTR (all final fields, every label); for the first two UTRs also every single
    em_algo call against the oracle's call trace."""
    from oracle.pool import fit_many
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_chunk
    base = 31337 + reads
    chunk = synth_chunk(n, reads, k_cap=min(kcap, 5), base_seed=base)
    kw = dict(n_max_apa=kcap, n_min_apa=1)
    preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df in chunk]
    eng = Engine(device=0)
    full = eng.run(preps, rng_mode="per_utr", seed=5, re_run_mode=True)
    outs = fit_many([(i, reads, min(kcap, 5), base, 5 + i, True, kw) for i in range(n)])
    _assert_parameters_equal_oracle([to_parameters(r) for r in full], outs)
    chunk, preps = chunk[:2], preps[:2]
    res = eng.run(preps, rng_mode="per_utr", seed=5, re_run_mode=False, keep_trace=True)
    for i, ((g, df), r) in enumerate(zip(chunk, res)):
        np.random.seed(5 + i)
        want, model = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values,
                                           re_run_mode=False, **kw)
        assert len(eng.traces[i]) == len(model.calls)
        q = r.prep
        same = 0
        for ft, rc in zip(eng.traces[i], model.calls):
            same += (np.array_equal(q.theta[ft.a_idx], rc["a1"]) and np.array_equal(q.betas[ft.b_idx], rc["b1"])
                     and len(ft.lb) == len(rc["lb"]) and np.allclose(ft.ws, rc["w1"], rtol=1e-6, atol=1e-10))
        assert same == len(model.calls), f"{g}: {same}/{len(model.calls)} em_algo calls identical"
        para = to_parameters(r)
        assert para.K == want.K and np.array_equal(para.alpha_arr, want.alpha_arr)
        assert np.array_equal(para.beta_arr, want.beta_arr) and np.array_equal(para.label_arr, want.label_arr)
        assert np.allclose(para.ws, want.ws, rtol=1e-4, atol=1e-9)


def test_infer_pa_all_then_merge_pa_vs_reference(tmp_path):
    """SURVEY.md 8(f) rank 3 end to end on the GPU: `infer_pa_all` (this build, GPU) writes pkl_output/*.res.pkl for the
    synthetic chunk directory of tests/merge_chain_dir.py, `merge_pa` (this build) merges it; the expectation is the
    REFERENCE's own merge_pa (junction_handler.py:44-147) run on the same directory with the oracle's fits
    (tests/golden/fixture_merge_chain.npz).  Every per-UTR fit must equal the oracle's (K, alpha, beta, all labels, the
    read / cell ids) and both merged outputs must equal the reference's in every field."""
    import pickle
    import merge_chain_dir as mc
    from conftest import load_npz
    from test_merge_pa import _chain_check
    from scape_amd.apa_core import infer_all
    f = load_npz("fixture_merge_chain.npz")
    recs = mc.write_inputs(str(tmp_path))
    infer_all(str(tmp_path), gpus=1, rng_mode="per_utr", seed=mc.SEED, re_run_mode=True, **mc.KW)
    ri = 0
    for fi in range(mc.N_FILES):
        with open(tmp_path / "pkl_output" / (mc.stem(fi) + ".res.pkl"), "rb") as fh:
            for _ in range(mc.PER_FILE):
                p, pre = pickle.load(fh), f"fit{ri}_"
                assert p.gene_info_str == recs[ri][2] == str(f[pre + "gene_info_str"])
                assert p.K == int(f[pre + "K"]) and p.L == int(f[pre + "L"])
                for k in ("alpha_arr", "beta_arr", "label_arr", "cb_id_arr", "readID_arr"):
                    assert np.array_equal(getattr(p, k), f[pre + k]), (p.gene_info_str, k)
                assert np.allclose(p.ws, f[pre + "ws"], rtol=1e-9, atol=1e-13)
                ri += 1
    assert ri == len(recs)
    _chain_check(tmp_path, f)


def _many_sites_utr(n_sites=34, reads=3000, seed=5, gap=140, beta=8.0):
    """A UTR with more true pA sites than the old K <= 31 limit of the kernels (inputs only)."""
    rng = np.random.default_rng(seed)
    alphas = 400 + gap * np.arange(n_sites)
    comp = rng.integers(0, n_sites, reads)
    theta = rng.normal(alphas[comp], beta)
    s = rng.choice(np.arange(20, 150, 10), size=reads)
    x = np.clip(np.rint(rng.normal(theta + s - 300, 50)), 0, np.maximum(theta - 31, 0))
    room = np.maximum(theta - x, 31)
    l = np.clip(np.floor(31 + rng.random(reads) * (np.minimum(132, room) - 31 + 1)), 31, 132)
    pa = np.full(reads, np.nan)
    has = rng.random(reads) < 0.3
    pa[has] = np.rint(theta[has])
    return pd.DataFrame({"x": x.astype(np.int64), "l": l.astype(np.int64), "r": np.full(reads, np.nan), "pa": pa,
                         "cb_id": np.arange(reads), "read_id": np.arange(reads)})


def test_rerun_loop_beyond_31_components_vs_oracle(oracle):
    """The reference's re-run loop has no K cap (apa_core.py:1023-1030: n_max_apa += 2 while K == n_max_apa).  A UTR
    with 34 true sites, n_max_apa = n_min_apa = 31 and min_ws = 0 (nothing pruned) ends its first sweep at K = 31 = n_max
    and re-runs with K = 33, 32, 31 - jobs beyond the 32-column kernels (ABI 3 stopped there with a warning).  Every
    em_algo call of both sweeps and the final Parameters against the oracle."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    df = _many_sites_utr()
    kw = dict(n_max_apa=31, n_min_apa=31, min_ws=0.0)
    prep = prepare_utr(df, gene_info_str="syn:K33:1:1-5500:+", **kw)
    eng = Engine(device=0)
    res = eng.run([prep], rng_mode="per_utr", seed=3, re_run_mode=True, keep_trace=True)[0]
    np.random.seed(3)
    want, model = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values, re_run_mode=True, **kw)
    assert model.n_max_apa >= 33 and max(c["K"] for c in model.calls) >= 33          # the loop did go beyond 31
    assert len(eng.traces[0]) == len(model.calls)
    q = res.prep
    for ft, rc in zip(eng.traces[0], model.calls):
        assert ft.K == rc["K"]
        assert np.array_equal(q.theta[ft.a_idx], rc["a1"]) and np.array_equal(q.betas[ft.b_idx], rc["b1"]), rc["K"]
        assert len(ft.lb) == len(rc["lb"]) and np.allclose(ft.ws, rc["w1"], rtol=1e-6, atol=1e-10)
    para = to_parameters(res)
    assert para.K == want.K and np.array_equal(para.alpha_arr, want.alpha_arr) and np.array_equal(para.beta_arr, want.beta_arr)
    assert np.array_equal(para.label_arr, want.label_arr)
    assert np.allclose(para.ws, want.ws, rtol=1e-6, atol=1e-10) and para.bic == pytest.approx(want.bic, rel=1e-9)


def test_beta_step_one_vs_oracle(oracle):
    """beta_step = 1 gives 69 beta grid values (apa_core.py:942 builds arange(beta_step, max_beta, beta_step) of any
    length; ABI 3 refused more than 64): Phase B's generic path, 69-row alpha groups in the M-step tiles; whole fit vs
    the oracle."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_chunk
    kw = dict(n_max_apa=3, n_min_apa=1, beta_step=1)
    chunk = synth_chunk(2, 300, k_cap=3, base_seed=909)
    preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df in chunk]
    assert len(preps[0].betas) == 69
    res = Engine(device=0).run(preps, rng_mode="per_utr", seed=21, re_run_mode=True)
    for i, ((g, df), r) in enumerate(zip(chunk, res)):
        np.random.seed(21 + i)
        want, _m = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values, re_run_mode=True, **kw)
        para = to_parameters(r)
        assert para.K == want.K and np.array_equal(para.alpha_arr, want.alpha_arr) and np.array_equal(para.beta_arr, want.beta_arr)
        assert np.array_equal(para.label_arr, want.label_arr)
        assert np.allclose(para.ws, want.ws, rtol=1e-8, atol=1e-12) and para.bic == pytest.approx(want.bic, rel=1e-9)


def test_random_model_parameters_vs_oracle(oracle):
    """Every model key of parameters.toml the path reads (ApaModel.__init__, apa_core.py:333-363) drawn at random - theta
    / beta grids (uniform-grid Phase B with 11- and 12-step windows, the one-kernel form for > 16 beta values), fragment
    size model, poly(A) length grid, pA gap, pruning thresholds, K range - on small UTRs with pA-site and r-known reads;
    whole fits (re-run loop on) against the oracle, every field."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    rng = np.random.default_rng(int(os.environ.get("SCAPE_TEST_RANDOM_SEED", 2027)))      # (a longer hunt: more cases / another seed)
    for case in range(int(os.environ.get("SCAPE_TEST_RANDOM_CASES", 10))):
        kw = dict(theta_step=int(rng.choice([5, 7, 9, 12, 15])), beta_step=int(rng.choice([3, 5, 10])),
                  max_beta=int(rng.choice([40, 55, 70, 90])), n_max_apa=int(rng.integers(2, 7)), n_min_apa=1,
                  min_ws=float(rng.choice([0.01, 0.05, 0.1])), max_unif_ws=float(rng.choice([0.1, 0.15, 0.3])),
                  min_pa_gap=int(rng.choice([60, 100, 150])), mu_f=int(rng.choice([250, 300, 350])),
                  sigma_f=int(rng.choice([40, 50, 65])), min_LA=int(rng.choice([10, 20, 30])),
                  max_LA=int(rng.choice([150, 180, 200])))     # (>= the synthetic reads' largest r: beyond the s grid the reference's r-known likelihood is NaN)
        seed = 300 + 10 * case
        chunk = [synth_utr(i, int(rng.integers(150, 500)), k_cap=4, base_seed=seed, pa_rate=float(rng.choice([0.0, 0.02, 0.2])),
                           r_rate=float(rng.choice([0.0, 0.05])))[:2] for i in range(3)]
        preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df in chunk]
        res = Engine(device=0).run(preps, rng_mode="per_utr", seed=seed, re_run_mode=True)
        for i, ((g, df), r) in enumerate(zip(chunk, res)):
            np.random.seed(seed + i)
            want, _m = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values, re_run_mode=True, **kw)
            para = to_parameters(r)
            assert para.K == want.K and np.array_equal(para.alpha_arr, want.alpha_arr), (case, i, kw)
            assert np.array_equal(para.beta_arr, want.beta_arr) and np.array_equal(para.label_arr, want.label_arr), (case, i, kw)
            assert np.allclose(para.ws, want.ws, rtol=1e-6, atol=1e-10) and para.bic == pytest.approx(want.bic, rel=1e-9), (case, i, kw)
            assert len(para.lb_arr) == len(want.lb_arr), (case, i, kw)


def test_reference_stream_random_model_parameters_vs_oracle(oracle):
    """The CLI's default mode - one random stream per chunk (np.random.seed(seed), apa_core.py:125), the UTRs drawing
    from it one after the other; here several of them per EM call with predicted predecessors (Engine._drive_streams) -
    under random model parameters: every UTR of the chunk against the oracle run serially on the same stream, and the
    generator state both leave."""
    from scape_amd.apa_core import to_parameters
    from scape_amd.engine import Engine
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    rng = np.random.default_rng(int(os.environ.get("SCAPE_TEST_RANDOM_SEED", 4099)))
    eng = Engine(device=0)
    for case in range(int(os.environ.get("SCAPE_TEST_RANDOM_CASES", 10)) // 2):
        kw = dict(theta_step=int(rng.choice([6, 9, 12])), beta_step=int(rng.choice([5, 10])), max_beta=int(rng.choice([45, 70])),
                  n_max_apa=int(rng.integers(2, 6)), n_min_apa=1, min_ws=float(rng.choice([0.02, 0.05, 0.1])),
                  max_unif_ws=float(rng.choice([0.1, 0.15])), min_pa_gap=int(rng.choice([80, 100])))
        seed = 40 + case
        chunk = [synth_utr(i, int(rng.integers(150, 450)), k_cap=4, base_seed=900 + 20 * case,
                           pa_rate=float(rng.choice([0.0, 0.02])))[:2] for i in range(7)]
        preps = [prepare_utr(df, gene_info_str=g, **kw) for g, df in chunk]
        rs = np.random.RandomState(seed)
        res = eng.run(preps, rng_mode="reference", rs=rs, re_run_mode=True)
        np.random.seed(seed)
        for i, ((g, df), r) in enumerate(zip(chunk, res)):
            want, _m = oracle.subsample_run(df["x"].values, df["l"].values, df["r"].values, df["pa"].values, re_run_mode=True, **kw)
            para = to_parameters(r)
            assert para.K == want.K and np.array_equal(para.alpha_arr, want.alpha_arr), (case, i, kw)
            assert np.array_equal(para.beta_arr, want.beta_arr) and np.array_equal(para.label_arr, want.label_arr), (case, i, kw)
            assert np.allclose(para.ws, want.ws, rtol=1e-6, atol=1e-10) and para.bic == pytest.approx(want.bic, rel=1e-9), (case, i, kw)
        st, ot = rs.get_state(), np.random.get_state()
        assert np.array_equal(st[1], ot[1]) and st[2] == ot[2], (case, "generator state after the chunk")


def test_phase_b_variants_give_identical_tensors(hip_ctx, monkeypatch):
    """Phase B has three forms: k_phase_b (one kernel: the operator seam, non-uniform theta grids, > 16 beta values),
    tables + per-alpha matrix path (SCAPE_HIP_PHASE_B=split) and - the default on uniform grids - tables + log-bin
    columns + per-wavefront sliding windows.  They restate the same sums in the same order: the tensors must be equal
    bit for bit, and so must an EM call on them (which also reads the tile extents each variant raises) - on UTRs with
    few / many (> PB_LOGCAP = 64, written straight into the tensor) / only log-domain bins, r-known reads, one bin, a
    9 kb UTR."""
    from scape_amd.engine import HipBatch, _Job
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    rng = np.random.default_rng(17)

    def mk(x, l, r=None, pa=None):
        n = len(x)
        return pd.DataFrame({"x": np.asarray(x, dtype=np.int64), "l": np.asarray(l, dtype=np.int64),
                             "r": np.full(n, np.nan) if r is None else r, "pa": np.full(n, np.nan) if pa is None else pa,
                             "cb_id": np.arange(n), "read_id": np.arange(n)})
    dfs = [synth_utr(i, 700, k_cap=4, base_seed=515, pa_rate=rate, r_rate=rr)[1]
           for i, (rate, rr) in enumerate([(0.015, 0.0), (0.0, 0.0), (0.4, 0.0), (0.02, 0.1), (0.015, 0.0)])]
    dfs += [mk(rng.integers(300, 600, 150), rng.integers(31, 133, 150), pa=rng.integers(700, 760, 150).astype(float)),
            mk([400] * 120, [98] * 120), mk(rng.integers(0, 9000, 400), rng.integers(31, 133, 400))]
    preps = [prepare_utr(df, gene_info_str=f"syn:PB{i}:1:1-9999:+", n_max_apa=3) for i, df in enumerate(dfs)]
    n_log = [int((~np.isnan(q.pa) | ~np.isnan(q.r)).sum()) for q in preps]
    assert min(n_log) == 0 and max(n_log) > 64 and any(0 < v <= 64 for v in n_log), n_log
    _phase_b_variants_check(hip_ctx, monkeypatch, preps, rng)
    # other uniform grids: the interior block of four, the 11- and 12-step forms, windows much narrower than the ring
    for kw in ({"theta_step": 15}, {"theta_step": 6, "max_beta": 45}, {"theta_step": 8, "max_beta": 65}, {"theta_step": 21, "max_beta": 80, "beta_step": 10}):
        preps = [prepare_utr(df, gene_info_str=f"syn:PB{i}:1:1-9999:+", n_max_apa=3, **kw) for i, df in enumerate(dfs[:4] + dfs[-1:])]
        _phase_b_variants_check(hip_ctx, monkeypatch, preps, rng)


def _phase_b_variants_check(hip_ctx, monkeypatch, preps, rng):
    from scape_amd.engine import HipBatch, _Job
    jobs = []
    for u, q in enumerate(preps):
        for K in (1, 2, 3):
            a = np.sort(rng.choice(q.T, K, replace=False)).astype(np.int32)
            w = rng.dirichlet(np.ones(K + 1))
            ka = (np.arange(50) % K).astype(np.int8)
            jobs.append(_Job(u, K, False, a, rng.integers(0, len(q.betas), K).astype(np.int32), w, ka))
    out = {}
    for mode in ("v2", "split", None):
        if mode is None:
            monkeypatch.delenv("SCAPE_HIP_PHASE_B", raising=False)
        else:
            monkeypatch.setenv("SCAPE_HIP_PHASE_B", mode)
        batch = HipBatch(hip_ctx, preps)
        batch.build()
        assert batch.phase_b_form() == {"v2": 0, "split": 1, None: 2}[mode]
        tensors = [batch.fetch_tensor(u) for u in range(len(preps))]
        fits = batch.em(jobs)
        out[mode] = (tensors, [(f.K, f.a_idx.tobytes(), f.b_idx.tobytes(), f.ws.tobytes(), f.bic, f.lb.tobytes()) for f in fits])
        batch.free()
    for mode in ("split", None):
        for u, (a, b) in enumerate(zip(out["v2"][0], out[mode][0])):
            assert np.array_equal(a, b), (mode, u, int((a != b).sum()))
        assert out["v2"][1] == out[mode][1], mode


def test_mstep_kernels_give_identical_bits(hip_ctx, monkeypatch):
    """The M-step has three kernels - k2_mstep (register-staged, SCAPE_HIP_MSTEP=v2), k3_mstep (LDS-DMA rings, one tile
    per workgroup, v3) and k4_mstep (two tiles per workgroup, job slots ranked by support; the default of wave-sized
    calls).  Every score is the same chain of MFMAs in all of them: the same jobs must come back bit for bit - K = 1..10
    (one to four column groups per tile, tiles with more than 64 jobs: passes) on UTRs of different sizes."""
    from scape_amd.engine import Engine, HipBatch
    from scape_amd.host import prepare_utr
    from scape_amd.synth import synth_utr
    monkeypatch.setenv("SCAPE_HIP_SPLIT_MAXTILES", "0")        # never cut a tile's jobs into passes for several workgroups
    monkeypatch.setenv("SCAPE_HIP_WIDE_MAXJOBS", "0")
    kw = dict(n_max_apa=10, n_min_apa=1)
    preps = [prepare_utr(df, gene_info_str=g, **kw)
             for g, df, _ in (synth_utr(i, reads, k_cap=5, base_seed=6100) for i, reads in enumerate((1500, 400, 2500, 900, 1200, 700, 1800, 300, 1000)))]
    eng = Engine(device=0)
    plan = eng.plan(preps, [(6100 + i) % 2 ** 32 for i in range(len(preps))])
    pj = plan["main"]
    # more than 64 jobs on the tiles of UTR 0: its sweep three times over in one call
    from scape_amd.engine import concat_packed, PackedJobs
    lo, hi = int(plan["spans"][0]), int(plan["spans"][1])
    first = PackedJobs(pj.ju[lo:hi], pj.jk[lo:hi], pj.jf[lo:hi], pj.a[lo:hi], pj.b[lo:hi], pj.w[lo:hi], pj.ka[lo:hi])
    big = concat_packed([pj, first, first])
    batch = HipBatch(hip_ctx, preps)
    batch.build()
    got = {}
    for mode in ("v2", "v3", None):
        if mode is None:
            monkeypatch.delenv("SCAPE_HIP_MSTEP", raising=False)
        else:
            monkeypatch.setenv("SCAPE_HIP_MSTEP", mode)
        got[mode] = [np.array(x).copy() for x in batch.em_packed(big)]
    batch.free()
    for mode in ("v3", None):
        for x, y in zip(got["v2"], got[mode]):
            assert np.array_equal(x, y), mode
    n = len(pj)
    for x in got[None]:                                     # the repeated jobs came back like their originals
        assert np.array_equal(x[n:n + (hi - lo)], x[lo:hi]) and np.array_equal(x[n + (hi - lo):], x[lo:hi])
