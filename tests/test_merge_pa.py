"""merge_pa (scape_amd/junction_handler.py) against
(1) the reference's committed example outputs res.gene.pkl / res.utr.pkl (tests/golden/fixture_merge.npz),
(2) the reference's own proc_junction_{pos,neg}_pa run in the build container on random genes
    (tests/golden/trace_merge_syn.npz, generator tests/golden/make_golden.py merge),
(3) the CLI on a directory written by this build.  Everything is exact (integer bookkeeping; ws is a ratio of
integers computed the same way)."""
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest

from conftest import load_npz

FIELDS = ("alpha_arr", "beta_arr", "ws", "label_arr", "cb_id_arr", "readID_arr")


def _records(f, pre, n):
    ins, res = {}, {}
    for j in range(n):
        p = f"{pre}r{j}_"
        g = str(f[p + "gene_info_str"])
        res[g] = SimpleNamespace(gene_info_str=g, K=int(f[p + "K"]), alpha_arr=f[p + "alpha_arr"], beta_arr=f[p + "beta_arr"],
                                 label_arr=f[p + "label_arr"], cb_id_arr=f[p + "cb_id_arr"], readID_arr=f[p + "readID_arr"])
        ins[g] = pd.DataFrame({c: f[p + "in_" + c] for c in ("read_id", "junction", "seg1_en", "seg2_en")})
    return ins, res


def _same(para, f, pre):
    assert para.gene_info_str == str(f[pre + "gene_info_str"])
    assert para.K == int(f[pre + "K"]) and para.L == int(f[pre + "L"]) and para.title == str(f[pre + "title"])
    for k in FIELDS:
        got, want = getattr(para, k), f[pre + k]
        assert got.dtype == want.dtype and np.array_equal(got, want), (pre, k, got, want)
    assert not hasattr(para, "bic")


@pytest.mark.parametrize("mode", ["gene", "utr"])
def test_committed_example_outputs(mode):
    from scape_amd.junction_handler import merge_gene
    f = load_npz("fixture_merge.npz")
    for name in f["names"]:
        ins, res = _records(f, f"{name}_", int(f[f"{name}_n_rec"]))
        groups = {}
        for g in res:
            key = g.split(":")[1] if mode == "gene" else ":".join(g.split(":")[1:3])
            groups.setdefault(key, {})[g] = res[g]
        gold = {str(f[f"{name}_{mode}_g{j}_gene_info_str"]): f"{name}_{mode}_g{j}_" for j in range(int(f[f"{name}_{mode}_n_gold"]))}
        assert len(gold) == len(groups)
        for key, rs in groups.items():
            para, _j, _c = merge_gene({g: ins[g] for g in rs}, rs, key)
            _same(para, f, gold[para.gene_info_str])


def test_reference_run_on_random_genes():
    from scape_amd.junction_handler import merge_gene
    f = load_npz("trace_merge_syn.npz")
    n, merged, dropped_sites, multi = int(f["n_case"]), 0, 0, 0
    for c in range(n):
        pre = f"c{c}_"
        ins, res = _records(f, pre, int(f[pre + "n_rec"]))
        para, junc, change = merge_gene(ins, res, str(f[pre + "key"]))
        _same(para, f, pre + "out_")
        assert junc == int(f[pre + "junc"]) and change == int(f[pre + "change"])
        merged += change
        multi += len(res) > 1
        dropped_sites += sum(r.K for r in res.values()) - para.K
    assert n >= 300 and merged >= 40 and multi >= 60 and dropped_sites >= 100     # the trace exercises the branches


def test_gene_without_any_site_raises_like_the_reference():
    from scape_amd.junction_handler import merge_gene
    g = "chr1:G:1:100-2100:+"
    res = {g: SimpleNamespace(gene_info_str=g, K=0, alpha_arr=np.zeros(0, int), beta_arr=np.zeros(0), label_arr=np.zeros(5, int),
                              cb_id_arr=np.arange(5), readID_arr=np.arange(5))}
    ins = {g: pd.DataFrame({"read_id": np.arange(5), "junction": np.zeros(5, int), "seg1_en": np.nan, "seg2_en": np.nan})}
    with pytest.raises(IndexError):
        merge_gene(ins, res, "G")


def test_merge_pa_cli_on_own_directory(tmp_path):
    """Directory layout and error behaviour of _merge_pa (:44-71): the results are this build's own pickles of
    scape.apa_core.Parameters, the inputs prepare_input-style tuples."""
    from click.testing import CliRunner
    from scape.cli import cli
    from scape_amd.apa_core import Parameters
    f = load_npz("trace_merge_syn.npz")
    out = tmp_path / "o"
    (out / "pkl_input").mkdir(parents=True)
    r = CliRunner().invoke(cli, ["merge_pa", "--output_dir", str(out)])
    assert r.exit_code != 0 and "infer_pa" in str(r.exception)
    (out / "pkl_output").mkdir()
    cases = [c for c in range(int(f["n_case"])) if ":" not in str(f[f"c{c}_key"])
             and all(int(f[f"c{c}_r{j}_K"]) > 0 for j in range(int(f[f"c{c}_n_rec"])))][:12]   # K = 0 records raise (above)
    want = {}
    for fi, chunk in enumerate((cases[:7], cases[7:])):
        with open(out / "pkl_input" / f"s.100.2.{fi + 1}.input.pkl", "wb") as fin, \
                open(out / "pkl_output" / f"s.100.2.{fi + 1}.res.pkl", "wb") as fres:
            for c in chunk:
                pre = f"c{c}_"
                ins, res = _records(f, pre, int(f[pre + "n_rec"]))
                for g, r_ in res.items():
                    n = len(ins[g])
                    df = pd.DataFrame({"x": np.zeros(n, int), "l": 50, "r": np.nan, "pa": np.nan, "cb_id": 0,
                                       "read_id": ins[g]["read_id"], "junction": ins[g]["junction"],
                                       "seg1_en": ins[g]["seg1_en"], "seg2_en": ins[g]["seg2_en"]})
                    pickle.dump((g, df), fin)
                    p = Parameters(title="Final Result", alpha_arr=r_.alpha_arr, beta_arr=r_.beta_arr,
                                   ws=np.full(r_.K + 1, 1 / (r_.K + 1)), L=2000, cb_id_arr=r_.cb_id_arr, readID_arr=r_.readID_arr)
                    p.label_arr, p.gene_info_str = r_.label_arr, g
                    pickle.dump(p, fres)
                want[str(f[pre + "out_gene_info_str"])] = pre + "out_"
    r = CliRunner().invoke(cli, ["merge_pa", "--output_dir", str(out)])
    assert r.exit_code == 0, r.output + repr(r.exception)
    got = []
    with open(out / "res.gene.pkl", "rb") as fh:
        while True:
            try:
                got.append(pickle.load(fh))
            except EOFError:
                break
    assert len(got) == len(want) and type(got[0]).__module__ == "scape.apa_core"
    for para in got:
        _same(para, f, want[para.gene_info_str])
    r = CliRunner().invoke(cli, ["prebin", "--output_dir", str(out), "--workers", "2"])     # columnar copies of the chunks
    assert r.exit_code == 0, r.output + repr(r.exception)
    assert sorted(os.listdir(out / "pkl_input")) == ["s.100.2.1.binned.npz", "s.100.2.1.input.pkl",
                                                     "s.100.2.2.binned.npz", "s.100.2.2.input.pkl"]
    first = open(out / "res.gene.pkl", "rb").read()
    r = CliRunner().invoke(cli, ["merge_pa", "--output_dir", str(out)])
    assert r.exit_code == 0 and open(out / "res.gene.pkl", "rb").read() == first           # same bytes from the binned inputs
    r = CliRunner().invoke(cli, ["merge_pa", "--output_dir", str(out), "--utr_merge", "False"])
    assert r.exit_code == 0 and os.path.exists(out / "res.utr.pkl")
    os.remove(out / "pkl_output" / "s.100.2.2.res.pkl")
    r = CliRunner().invoke(cli, ["merge_pa", "--output_dir", str(out)])
    assert r.exit_code != 0 and "Number of *.res.pkl" in str(r.exception)


def test_parallel_merge_over_chunk_files_equals_serial(tmp_path):
    """Six chunk files, the records of multi-record genes spread over different files (a gene belongs to the chunk
    it first appears in, later chunks are loaded for its other records): the worker-pool path writes the same
    bytes as the in-process path."""
    from scape_amd.apa_core import Parameters
    from scape_amd.junction_handler import _merge_pa
    f = load_npz("trace_merge_syn.npz")
    out = tmp_path / "o"
    (out / "pkl_input").mkdir(parents=True)
    (out / "pkl_output").mkdir()
    cases = [c for c in range(int(f["n_case"])) if ":" not in str(f[f"c{c}_key"])
             and all(int(f[f"c{c}_r{j}_K"]) > 0 for j in range(int(f[f"c{c}_n_rec"])))][:40]
    F = 6
    fin = [open(out / "pkl_input" / f"s.100.{F}.{i + 1}.input.pkl", "wb") for i in range(F)]
    fres = [open(out / "pkl_output" / f"s.100.{F}.{i + 1}.res.pkl", "wb") for i in range(F)]
    n_multi = 0
    for ci, c in enumerate(cases):
        pre = f"c{c}_"
        ins, res = _records(f, pre, int(f[pre + "n_rec"]))
        n_multi += len(res) > 1
        for j, (g, r_) in enumerate(res.items()):
            k = (ci + 2 * j) % F                      # records of one gene land in different chunk files
            n = len(ins[g])
            df = pd.DataFrame({"x": np.zeros(n, int), "l": 50, "r": np.nan, "pa": np.nan, "cb_id": 0,
                               "read_id": ins[g]["read_id"], "junction": ins[g]["junction"],
                               "seg1_en": ins[g]["seg1_en"], "seg2_en": ins[g]["seg2_en"]})
            pickle.dump((g, df), fin[k])
            p = Parameters(title="Final Result", alpha_arr=r_.alpha_arr, beta_arr=r_.beta_arr,
                           ws=np.full(r_.K + 1, 1 / (r_.K + 1)), L=2000, cb_id_arr=r_.cb_id_arr, readID_arr=r_.readID_arr)
            p.label_arr, p.gene_info_str = r_.label_arr, g
            pickle.dump(p, fres[k])
    for fh in fin + fres:
        fh.close()
    assert n_multi >= 5
    for mode, name in ((True, "res.gene.pkl"), (False, "res.utr.pkl")):
        n0 = _merge_pa(str(out), mode, workers=0)
        serial = open(out / name, "rb").read()
        n1 = _merge_pa(str(out), mode, workers=2)
        assert n0 == n1 > 0 and open(out / name, "rb").read() == serial
        assert not [p for p in os.listdir(out) if ".part" in p]


def _chain_check(root, f):
    """res.gene.pkl / res.utr.pkl of `root` (written by this build's merge_pa) against the reference's own merge_pa run
    on the same directory content (tests/golden/fixture_merge_chain.npz, generator make_golden.py mergechain)."""
    from scape_amd.junction_handler import _merge_pa
    for mode, fn, tag in ((True, "res.gene.pkl", "gene"), (False, "res.utr.pkl", "utr")):
        _merge_pa(str(root), mode, workers=0)
        got = []
        with open(os.path.join(root, fn), "rb") as fh:
            while True:
                try:
                    got.append(pickle.load(fh))
                except EOFError:
                    break
        want = {str(f[f"{tag}{k}_gene_info_str"]): f"{tag}{k}_" for k in range(int(f[f"{tag}_n"]))}
        assert len(got) == len(want) and {p.gene_info_str for p in got} == set(want)
        for para in got:
            _same(para, f, want[para.gene_info_str])


def test_merge_chain_directory_vs_reference_merge(tmp_path):
    """The infer_pa -> merge_pa chain directory (tests/merge_chain_dir.py) with the ORACLE's fits as pkl_output: this
    build's merge_pa must write what the reference's merge_pa wrote for it (both modes, every field).  The -m gpu
    twin (test_gpu_parity.py::test_infer_pa_all_then_merge_pa_vs_reference) replaces the oracle's fits by the GPU's."""
    import merge_chain_dir as mc
    from scape_amd.apa_core import Parameters
    f = load_npz("fixture_merge_chain.npz")
    recs = mc.write_inputs(str(tmp_path))
    outs = [open(tmp_path / "pkl_output" / (mc.stem(fi) + ".res.pkl"), "wb") for fi in range(mc.N_FILES)]
    for ri, (fi, _j, g, df) in enumerate(recs):
        pre = f"fit{ri}_"
        assert str(f[pre + "gene_info_str"]) == g
        p = Parameters(title="Final Result", alpha_arr=f[pre + "alpha_arr"], beta_arr=f[pre + "beta_arr"], ws=f[pre + "ws"],
                       L=int(f[pre + "L"]), cb_id_arr=f[pre + "cb_id_arr"], readID_arr=f[pre + "readID_arr"])
        p.label_arr, p.gene_info_str = f[pre + "label_arr"], g
        pickle.dump(p, outs[fi])
    for fh in outs:
        fh.close()
    _chain_check(tmp_path, f)
