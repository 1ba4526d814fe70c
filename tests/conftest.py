import os
import sys

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


def pytest_sessionstart(session):
    """GPU runs: bring the prep-worker pool up before any test initialises the GPU, so its fork server
    is born without a HIP context (scape_amd/pipeline.py)."""
    m = session.config.getoption("-m") or ""
    if "gpu" in m and "not gpu" not in m:
        from scape_amd.pipeline import shared_pool
        shared_pool(4)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def utr_df(f, i):
    """(gene_info_str, DataFrame) of UTR i of a fixture/trace file."""
    cols = {c: f[f"u{i}_{c}"] for c in ["x", "l", "r", "pa", "cb_id", "read_id"]}
    return str(f[f"u{i}_gene_info_str"]), pd.DataFrame(cols)


def trace_params(f):
    p = {k[6:]: f[k].item() for k in f.files if k.startswith("param_")}
    return p


def trace_pre_para(f):
    """pre-specified Parameters of a fixed_run_mode trace (None for ordinary traces)."""
    if "pre_alpha_arr" not in f.files:
        return None
    from types import SimpleNamespace
    return SimpleNamespace(alpha_arr=f["pre_alpha_arr"], beta_arr=f["pre_beta_arr"], ws=f["pre_ws"],
                           L=int(f["pre_L"]), K=len(f["pre_alpha_arr"]))


@pytest.fixture(scope="session")
def oracle():
    from oracle import scape_oracle
    scape_oracle.lib()
    return scape_oracle


@pytest.fixture(scope="session")
def hip_ctx():
    from scape_amd import _lib
    return _lib.default_context(0)


TRACES = ["synA", "synB", "synC", "synD", "synE", "synF", "chr17", "chr19", "toy"]
FIXTURES = ["chr17", "chr19", "toy"]
