"""Pins the CPU oracle (oracle/) to the reference:
(1) the scalar known-answer inputs of the reference's own kernel test script
    (src/scape/taichi_code_test.py:514-593, :741-744) against closed forms,
(2) traces of the reference's own Python run in the build container (tests/golden/trace_*.npz,
    made by tests/golden/make_golden.py): binning, Phase-A matrix, Phase-B tensor, every em_algo
    call (inits, k_arr, results), final Parameters - with the reference's RNG stream,
(3) the reference's committed example outputs (examples/*/pkl_output/*.res.pkl, transcribed to
    tests/golden/fixture_*.npz): K, alpha, labels (ws / bic only loosely).
"""
import math

import numpy as np
import pytest

from conftest import FIXTURES, TRACES, load_npz, trace_params, trace_pre_para

SENT = float(np.finfo("f").min)


# ---------------------------------------------------------------- (1) scalar known answers
def test_base_functions_known_answers(oracle):
    L = oracle.lib()
    assert L.so_my_log(3.4567) == pytest.approx(math.log(3.4567), rel=1e-15)
    assert L.so_my_log(-3.4567) == SENT
    z = (0.75 - 1.0) / 0.5
    assert L.so_logpdf_normal(0.75, 1.0, 0.5) == pytest.approx(-0.5 * z * z - math.log(0.5) - 0.5 * math.log(2 * math.pi), rel=1e-15)
    assert L.so_pdf_normal(0.75, 1.0, 0.5) == pytest.approx(math.exp(-0.5 * z * z) / math.sqrt(2 * math.pi) / 0.5, rel=1e-15)
    v = np.array([3.65, 7.89, 5., 6.12, 1.23])
    assert L.so_logsumexp(v.ctypes.data_as(oracle._D), 5) == pytest.approx(math.log(np.exp(v - 7.89).sum()) + 7.89, rel=1e-15)
    assert L.so_loglik_l_xt(30, 50, 70) == SENT                       # l=50 > theta-x=40
    assert L.so_loglik_l_xt(5, 50, 70) == pytest.approx(-math.log(65.0), rel=1e-15)
    assert L.so_lik_l_xt(30, 50, 70) == 0.0
    assert L.so_lik_l_xt(5, 50, 70) == pytest.approx(1 / 65.0, rel=1e-15)
    zz = (187 - 460) / 50
    assert L.so_loglik_x_st_pa(187, 460, 50) == pytest.approx(-0.5 * zz * zz - math.log(50) - 0.5 * math.log(2 * math.pi), rel=1e-15)
    z2 = (44 - (460 + 144 - 50)) / 50
    assert L.so_loglik_x_st(44, 144, 460, 50, 50) == pytest.approx(-0.5 * z2 * z2 - math.log(50) - 0.5 * math.log(2 * math.pi), rel=1e-15)
    assert L.so_lik_x_st(44, 144, 460, 50, 50) == pytest.approx(math.exp(-0.5 * z2 * z2) / math.sqrt(2 * math.pi) / 50, rel=1e-14)
    assert L.so_loglik_r_s(10, 20) == pytest.approx(-math.log(20), rel=1e-15)
    assert L.so_loglik_r_s(20, 10) == SENT
    assert L.so_lik_r_s(10, 20) == pytest.approx(1 / 20)
    assert L.so_lik_r_s(20, 10) == 0.0


def test_marginal_known_grid(oracle):
    """taichi_code_test.py:741-744: alpha=37, beta=5 on the 23-point grid -> window = first 2 points."""
    all_theta = np.arange(37, 236, 9).astype(np.float64)
    assert len(all_theta) == 23
    rng = np.random.RandomState(0)
    for i in range(3):
        A = rng.rand(i + 3, i + 5)
        th = all_theta[:A.shape[1]]
        got = oracle.get_loglik_marginal_tensor(th, np.array([5.0]), A)[0, 0]
        g = -0.5 * ((th[:2] - 37) / 5) ** 2 - math.log(5) - 0.5 * math.log(2 * math.pi)
        G = math.log(np.exp(g).sum())
        want = np.log(np.exp(A[:, :2] + g - G).sum(axis=1))
        assert np.allclose(got, want, rtol=1e-14, atol=0)


def test_np_pairwise_sum_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in [0, 1, 5, 7, 8, 9, 15, 16, 17, 100, 127, 128, 129, 255, 256, 1000, 1376, 4097, 10000]:
        a = rng.standard_normal(n) * 10 ** rng.uniform(-5, 5, n)
        assert oracle.lib().so_np_sum(a.ctypes.data_as(oracle._D), n) == (np.sum(a) if n else 0.0)


def test_numpy_twins_agree_with_c(oracle):
    f = load_npz("trace_synB.npz")
    p = trace_params(f)
    x, l, r, pa = (f[f"u1_bin_{c}"] for c in ("x", "l", "r", "pa"))
    s = np.arange(p["min_LA"], p["max_LA"], 10).astype(float)
    pmf = np.full(len(s), 1 / len(s))
    th = f["u1_all_theta"]
    A_c = oracle.phase_a(x, l, r, pa, th, s, pmf, p["mu_f"], p["sigma_f"])
    A_np = oracle.phase_a_np(x, l, r, pa, th, s, pmf, p["mu_f"], p["sigma_f"])
    assert np.allclose(A_c, A_np, rtol=1e-13, atol=0)
    rows = list(range(0, len(th), 17))
    M_c = oracle.get_loglik_marginal_tensor(th, f["u1_betas"], A_c)[rows]
    M_np = oracle.phase_b_np(th, f["u1_betas"], A_c, rows=rows)
    assert np.allclose(M_c, M_np, rtol=1e-13, atol=0)


# ---------------------------------------------------------------- (2) reference traces
def _run_oracle_on_trace(oracle, name):
    f = load_npz(f"trace_{name}.npz")
    p = trace_params(f)
    re_run = bool(p.pop("re_run_mode"))
    p.pop("fixed_run_mode", None)
    np.random.seed(int(f["seed"]))
    out = []
    for i in range(int(f["n_utr"])):
        st = np.random.get_state()
        assert np.array_equal(st[1], f[f"u{i}_rng_keys"]) and st[2] == int(f[f"u{i}_rng_pos"]), \
            f"{name} u{i}: RNG stream position differs from the reference before this UTR"
        res, model = oracle.subsample_run(f[f"u{i}_x"], f[f"u{i}_l"], f[f"u{i}_r"], f[f"u{i}_pa"],
                                          re_run_mode=re_run, pre_para=trace_pre_para(f), **p)
        out.append((res, model))
    return f, out


@pytest.mark.parametrize("name", TRACES)
def test_oracle_reproduces_reference_trace(oracle, name):
    f, out = _run_oracle_on_trace(oracle, name)
    for i, (res, model) in enumerate(out):
        tag = f"{name} u{i}"
        # binning (apa_core.py:285-327) - exact
        for k, v in (("x", model.x), ("l", model.l), ("r", model.r), ("pa", model.pa)):
            assert np.array_equal(v, f[f"u{i}_bin_{k}"], equal_nan=True), tag
        assert np.array_equal(model.cnt, f[f"u{i}_bin_cnt"]) and np.array_equal(model.idx, f[f"u{i}_bin_idx"]), tag
        assert np.array_equal(model.all_theta, f[f"u{i}_all_theta"]) and np.array_equal(model.betas, f[f"u{i}_betas"]), tag
        assert model.unif_ll == float(f[f"u{i}_unif_ll"]) and model.L == int(f[f"u{i}_L"]), tag
        assert np.array_equal(model.cov[1], f[f"u{i}_cov_y"]), tag
        # Phase A / Phase B (taichi_core.py) - to rounding
        A = f[f"u{i}_A"]
        assert np.array_equal(model.A == SENT, A == SENT), tag
        assert np.allclose(model.A, A, rtol=1e-13, atol=0), tag
        if f"u{i}_M" in f.files:
            M, Mo = f[f"u{i}_M"], model.M
        else:
            rows = f[f"u{i}_M_rows"]
            M, Mo = f[f"u{i}_M_sel"], model.M[rows]
        assert np.array_equal(Mo == SENT, M == SENT), tag
        assert np.allclose(Mo, M, rtol=1e-13, atol=0), tag
        fin = model.M > -1e30
        assert int((~fin).sum()) == int(f[f"u{i}_M_n_sent"]), tag
        assert float(model.M[fin].sum()) == pytest.approx(float(f[f"u{i}_M_sum_finite"]), rel=1e-12), tag
        # every em_algo call: same inits (RNG stream), same decisions, same numbers
        nc = len(f[f"u{i}_call_K"])
        assert len(model.calls) == nc, tag
        for c in range(nc):
            rc, K = model.calls[c], int(f[f"u{i}_call_K"][c])
            assert rc["K"] == K and rc["fixed"] == bool(f[f"u{i}_call_fixed"][c]), (tag, c)
            assert np.array_equal(rc["a0"], f[f"u{i}_call_a0"][c, :K]), (tag, c)
            assert np.array_equal(rc["b0"], f[f"u{i}_call_b0"][c, :K]), (tag, c)
            assert np.array_equal(rc["w0"], f[f"u{i}_call_w0"][c, :K + 1]), (tag, c)
            assert np.array_equal(rc["k_arr"], f[f"u{i}_call_k_arr"][c].astype(int)), (tag, c)
            assert np.array_equal(rc["a1"], f[f"u{i}_call_a1"][c, :K]), (tag, c)
            assert np.array_equal(rc["b1"], f[f"u{i}_call_b1"][c, :K]), (tag, c)
            assert len(rc["lb"]) == int(f[f"u{i}_call_nlb"][c]), (tag, c)
            assert np.allclose(rc["w1"], f[f"u{i}_call_w1"][c, :K + 1], rtol=1e-10, atol=1e-14), (tag, c)
            assert rc["bic"] == pytest.approx(float(f[f"u{i}_call_bic"][c]), rel=1e-11), (tag, c)
            assert np.allclose(rc["lb"], f[f"u{i}_call_lb"][c, :len(rc["lb"])], rtol=1e-11), (tag, c)
        # final Parameters
        assert res.K == int(f[f"u{i}_res_K"]), tag
        assert np.array_equal(res.alpha_arr, f[f"u{i}_res_alpha_arr"]), tag
        assert np.array_equal(res.beta_arr, f[f"u{i}_res_beta_arr"]), tag
        assert np.allclose(res.ws, f[f"u{i}_res_ws"], rtol=1e-10, atol=1e-14), tag
        assert res.bic == pytest.approx(float(f[f"u{i}_res_bic"]), rel=1e-11), tag
        assert np.array_equal(res.label_arr, f[f"u{i}_res_label_arr"]), tag
        assert res.title == str(f[f"u{i}_res_title"]), tag


# ---------------------------------------------------------------- (3) committed example outputs
@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_vs_committed_goldens(oracle, name):
    """The committed res.pkl files predate the current reference code, so they pin
    K, L, alpha to within one theta step and >= 99 % of the per-read labels; ws / bic to 2 %."""
    f = load_npz(f"fixture_{name}.npz")
    np.random.seed(1)                                   # apa_core.py:125
    for i in range(int(f["n_utr"])):
        res, model = oracle.subsample_run(f[f"u{i}_x"], f[f"u{i}_l"], f[f"u{i}_r"], f[f"u{i}_pa"],
                                          re_run_mode=True, n_max_apa=5, n_min_apa=1)
        tag = f"{name} u{i}"
        assert res.K == int(f[f"u{i}_gold_K"]), tag
        assert model.L == int(f[f"u{i}_gold_L"]), tag
        assert np.all(np.abs(res.alpha_arr - f[f"u{i}_gold_alpha_arr"]) <= 9), tag
        agree = np.mean(res.label_arr == f[f"u{i}_gold_label_arr"])
        assert agree >= 0.99, (tag, agree)
        assert np.allclose(res.ws, f[f"u{i}_gold_ws"], atol=2e-2), tag
        assert res.bic == pytest.approx(float(f[f"u{i}_gold_bic"]), rel=2e-2), tag
