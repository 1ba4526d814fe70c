"""Pre-binned columnar chunks: ``<stem>.binned.npz`` next to ``<stem>.input.pkl``.

``prepare_input`` writes one pickled (gene_info_str, DataFrame) per UTR (input_processor.py:223-259); at
MI355X rates the host spends more time decoding those pickles (1.3 ms per UTR with the non-executing reader)
and binning the reads (``bin_data``, apa_core.py:285-327, 0.9 ms) than the GPU spends on the UTR (0.4 ms).
The bin widths are constants, so the binned form is parameter independent: it is computed once per chunk and
stored as plain concatenated arrays with offsets - no pickle, nothing executable, loadable with
``numpy.load(allow_pickle=False)``:

    gene_info     [U]   str
    read_off      [U+1] int64   reads of UTR u are rows read_off[u]:read_off[u+1] of the per-read arrays
    bin_off       [U+1] int64   same for the per-bin arrays
    x, l, r, pa   [bins] f64    per-bin means (NaN where the reference has NaN)
    cnt           [bins] int64  reads per bin
    idx           [reads] int32 bin of each read (within its UTR)
    cb_id, read_id[reads] int64
    junc_off      [U+1] int64   junction reads only (what merge_pa reads from the input chunk):
    junc_pos      [junction reads] int32 position of the read inside its UTR; junc_seg1, junc_seg2 f64
    x_max, l_max  [U]   int64   over the raw reads (utr_length)
    source_size, source_mtime_ns, version         staleness check against the pickle it was made from

``infer_pa`` / ``infer_pa_all`` / the pipeline use the binned file when it is present and matches its pickle;
results are identical by construction (tests/test_host.py::test_binned_chunk_*).
"""
from __future__ import annotations

import os

import numpy as np

from .host import BinnedUtr, bin_utr

VERSION = 1
# note: seg1_en / seg2_en of non-junction reads (NaN in prepare_input's output) are not stored


def binned_path(pkl_input_file):
    return pkl_input_file[:-len(".input.pkl")] + ".binned.npz" if pkl_input_file.endswith(".input.pkl") \
        else pkl_input_file + ".binned.npz"


def write_binned(path, utrs, source=None):
    """utrs: iterable of (gene_info_str, DataFrame).  Returns the number of UTRs written."""
    genes, read_off, bin_off = [], [0], [0]
    cols = {k: [] for k in ("x", "l", "r", "pa", "cnt", "idx", "cb_id", "read_id", "junc_pos", "junc_seg1", "junc_seg2")}
    x_max, l_max, junc_off = [], [], [0]
    for gene, df in utrs:
        b = bin_utr(df)
        genes.append(str(gene))
        n = len(b.idx)
        read_off.append(read_off[-1] + n)
        bin_off.append(bin_off[-1] + len(b.cnt))
        for k in ("x", "l", "r", "pa", "cnt"):
            cols[k].append(getattr(b, k))
        cols["idx"].append(b.idx.astype(np.int32))
        cols["cb_id"].append(np.asarray(b.cb_id, dtype=np.int64))
        cols["read_id"].append(np.asarray(b.read_id, dtype=np.int64))
        if hasattr(df, "columns") and "junction" in df.columns:
            jp = np.nonzero(np.asarray(df["junction"]) == 1)[0]
            cols["junc_pos"].append(jp.astype(np.int32))
            cols["junc_seg1"].append(np.asarray(df["seg1_en"], dtype=np.float64)[jp])
            cols["junc_seg2"].append(np.asarray(df["seg2_en"], dtype=np.float64)[jp])
            junc_off.append(junc_off[-1] + len(jp))
        else:
            junc_off.append(junc_off[-1])
        x_max.append(b.x_max)
        l_max.append(b.l_max)
    dt = dict(x=np.float64, l=np.float64, r=np.float64, pa=np.float64, cnt=np.int64, idx=np.int32, cb_id=np.int64,
              read_id=np.int64, junc_pos=np.int32, junc_seg1=np.float64, junc_seg2=np.float64)
    out = {k: (np.concatenate(v).astype(dt[k]) if v else np.zeros(0, dt[k])) for k, v in cols.items()}
    st = os.stat(source) if source else None
    tmp = path + ".tmp.npz"
    np.savez(tmp, version=np.array(VERSION), gene_info=np.array(genes, dtype=str),
             read_off=np.array(read_off, dtype=np.int64), bin_off=np.array(bin_off, dtype=np.int64),
             x_max=np.array(x_max, dtype=np.int64), l_max=np.array(l_max, dtype=np.int64),
             junc_off=np.array(junc_off, dtype=np.int64),
             source_size=np.array(st.st_size if st else -1), source_mtime_ns=np.array(st.st_mtime_ns if st else -1), **out)
    os.replace(tmp, path)
    return len(genes)


def is_current(path, source):
    """True if `path` exists and was made from `source` as it is now."""
    if not os.path.exists(path):
        return False
    try:
        with np.load(path, allow_pickle=False) as f:
            st = os.stat(source)
            return (int(f["version"]) == VERSION and int(f["source_size"]) == st.st_size
                    and int(f["source_mtime_ns"]) == st.st_mtime_ns)
    except Exception:                                   # noqa: BLE001 - unreadable file = not current
        return False


def read_binned(path):
    """Yields (gene_info_str, BinnedUtr, junction columns dict) per UTR, in file order."""
    with np.load(path, allow_pickle=False) as f:
        if int(f["version"]) != VERSION:
            raise ValueError(f"{path}: binned chunk version {int(f['version'])}, expected {VERSION}")
        a = {k: f[k] for k in f.files}
    ro, bo, jo = a["read_off"], a["bin_off"], a["junc_off"]
    for u, gene in enumerate(a["gene_info"]):
        r0, r1, b0, b1 = int(ro[u]), int(ro[u + 1]), int(bo[u]), int(bo[u + 1])
        j0, j1 = int(jo[u]), int(jo[u + 1])
        junction = np.zeros(r1 - r0, dtype=np.int64)
        seg1, seg2 = np.full(r1 - r0, np.nan), np.full(r1 - r0, np.nan)
        jp = a["junc_pos"][j0:j1]
        junction[jp], seg1[jp], seg2[jp] = 1, a["junc_seg1"][j0:j1], a["junc_seg2"][j0:j1]
        b = BinnedUtr(a["x"][b0:b1], a["l"][b0:b1], a["r"][b0:b1], a["pa"][b0:b1], a["cnt"][b0:b1],
                      a["idx"][r0:r1].astype(np.int64), a["cb_id"][r0:r1], a["read_id"][r0:r1],
                      int(a["x_max"][u]), int(a["l_max"][u]))
        yield str(gene), b, dict(read_id=a["read_id"][r0:r1], junction=junction, seg1_en=seg1, seg2_en=seg2)


def prebin_chunk(pkl_input_file, force=False):
    """Write <stem>.binned.npz for one prepare_input chunk (skipped if it is current).  Returns its path."""
    from .apa_core import read_input_chunk
    path = binned_path(pkl_input_file)
    if force or not is_current(path, pkl_input_file):
        write_binned(path, read_input_chunk(pkl_input_file), source=pkl_input_file)
    return path
