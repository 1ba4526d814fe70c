"""``scape merge_pa``: the consumer of ``infer_pa``'s ``pkl_output/*.res.pkl``.

Per gene (or per gene:utr with ``--utr_merge False``) the pA sites of all its UTR records are put on
one genomic axis, sites that are artefacts of one splice junction are merged into the neighbouring
site, sites carrying <= 5 % of the reads are dropped, and one ``Parameters`` per gene is written to
``res.gene.pkl`` / ``res.utr.pkl``.  Behaviour follows the reference's
``src/scape/junction_handler.py`` (``_merge_pa`` :44-147, ``proc_junction_{neg,pos}_pa`` :261-495,
helpers :503-707) decision for decision; it is host-side integer bookkeeping (no GPU work), written
here as array operations per gene instead of per-read pandas look-ups.

Reference quirks kept on purpose (each one changes outputs):
  * a site with zero assigned reads counts as one read (:549);
  * the "weight" of a site is the live read total, so it grows as sites are merged into it (:586 aliases
    the two arrays);
  * the closest-site search keeps seeing sites that were already merged away (:606-621);
  * the surviving site ids are enumerated through a Python ``set`` (:659), whose iteration order decides
    the order of ``alpha_arr``; the same construct is used here;
  * reads of dropped sites and of the uniform component are removed from cb_id_arr / readID_arr /
    label_arr (:683-686); the output has no uniform weight, ``L = 0``, no ``bic``.
"""
from __future__ import annotations

import os
import pickle
from timeit import default_timer as timer

import click
import numpy as np

from .apa_core import Parameters, read_input_chunk

JUNCTION_PCT_THRES = 0.4          # _merge_pa :126
TOTAL_READ_PCT_THRES = 0.05       # _merge_pa :127


# ---------------------------------------------------------------- per-gene tables (ex_attr_Parameters_to_arr :503-588)
class _GeneTables:
    __slots__ = ("beta", "loc", "label", "cb", "read", "seg1", "seg2", "seg_label", "junc", "total",
                 "utr_st", "utr_en", "chrom", "gene", "strand")


def _column(df, name):
    v = df[name]
    return v.to_numpy() if hasattr(v, "to_numpy") else np.asarray(v)


def _rows_of(read_ids, wanted):
    """positions of `wanted` read ids in `read_ids` (DataFrame.loc on the read_id index, :553)"""
    order = np.argsort(read_ids, kind="stable")
    srt = read_ids[order]
    if len(srt) > 1 and np.any(srt[1:] == srt[:-1]):
        raise ValueError("merge_pa: duplicate read_id in an input chunk")
    pos = np.searchsorted(srt, wanted)
    bad = (pos >= len(srt)) | (srt[np.minimum(pos, len(srt) - 1)] != wanted) if len(srt) else np.ones(len(wanted), bool)
    if np.any(bad):
        raise KeyError(f"merge_pa: read ids {wanted[bad][:5].tolist()} of the result are not in the input chunk")
    return order[pos]


def _gene_tables(inputs, results):
    t = _GeneTables()
    beta, loc, label, cb, read, seg1, seg2, seg_label, junc, total, st_l, en_l = ([] for _ in range(12))
    k_off = 0
    for key in np.sort(list(results.keys())):          # UTR records in name order (:517)
        res = results[str(key)]
        info = res.gene_info_str.split(":")            # chrom:gene:utr:start-end:strand
        t.chrom, t.gene, t.strand = info[0], info[1], info[4]
        utr_st, utr_en = (int(v) for v in info[3].split("-"))
        st_l.append(utr_st)
        en_l.append(utr_en)
        K = int(res.K)
        alpha = np.asarray(res.alpha_arr)
        beta.append(np.asarray(res.beta_arr, dtype=np.float64))
        loc.append(utr_st + alpha if t.strand == "+" else utr_en - alpha + 1)
        lab_all = np.asarray(res.label_arr)
        on_site = lab_all < K                          # reads of the uniform component are left out
        lab = lab_all[on_site]
        ids = np.asarray(res.readID_arr)[on_site]
        label.append(k_off + lab)
        cb.append(np.asarray(res.cb_id_arr)[on_site])
        read.append(ids)
        n_read = np.bincount(lab, minlength=K)[:K].astype(np.float64) if K else np.zeros(0)
        n_read[n_read == 0] = 1                        # :549
        total.append(n_read)
        df = inputs[str(key)]
        rows = _rows_of(_column(df, "read_id"), ids)
        is_junc = _column(df, "junction")[rows] == 1
        seg1.append(_column(df, "seg1_en")[rows][is_junc].astype(np.float64))
        seg2.append(_column(df, "seg2_en")[rows][is_junc].astype(np.float64))
        seg_label.append(k_off + lab[is_junc])
        junc.append(np.bincount(lab[is_junc], minlength=K)[:K].astype(np.float64) if K else np.zeros(0))
        k_off += K
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)   # noqa: E731
    loc_all = cat(loc, np.int64)
    # site ids by genomic position, 5' -> 3' on the gene's strand (:572-577); same sort call as the reference
    order = np.argsort(loc_all) if t.strand == "+" else np.argsort(-loc_all)
    new_id = np.empty(len(order), dtype=np.int64)
    new_id[order] = np.arange(len(order))
    t.beta, t.loc = cat(beta, np.float64)[order], loc_all[order]
    t.junc, t.total = cat(junc, np.float64)[order], cat(total, np.float64)[order]
    t.label = new_id[cat(label, np.int64)]
    t.seg_label = new_id[cat(seg_label, np.int64)]
    t.cb, t.read = cat(cb, np.int64), cat(read, np.int64)
    t.seg1, t.seg2 = cat(seg1, np.float64), cat(seg2, np.float64)
    t.utr_st, t.utr_en = np.array(st_l, dtype=np.int64), np.array(en_l, dtype=np.int64)
    return t


# ---------------------------------------------------------------- junction merge (proc_junction_*_pa :261-495)
def _closest_site(position, loc, strand):
    """first site at or beyond `position` towards the 3' end (:606-621); None if there is none (or NaN)"""
    # ids run along the sorted positions (ascending on "+", descending on "-"), as the reference re-sorts here
    hit = np.nonzero(np.sort(loc) >= position)[0] if strand == "+" else np.nonzero(-np.sort(-loc) <= position)[0]
    return int(hit[0]) if len(hit) else None


def _resolve(site, merged_into):
    while site in merged_into:                         # update_pa_dict :625-627
        site = merged_into[site]
    return site


def _merge_junction_sites(t, junction_pct_thres):
    """Returns {site: site it was merged into} and whether any site had > threshold junction reads."""
    strand, loc, total, junc = t.strand, t.loc, t.total, t.junc
    pct = junc / total
    queue = [int(i) for i in np.nonzero(pct > junction_pct_thres)[0]]
    had_junction = 1 if queue else 0
    merged_into = {}
    while queue:
        first = queue.pop(0)
        if pct[first] <= junction_pct_thres:
            continue                                   # lost its junction share through an earlier merge
        with np.errstate(all="ignore"):
            sel = t.seg_label == first
            m1 = np.median(t.seg1[sel]) if sel.any() else np.nan
            m2 = np.median(t.seg2[sel]) if sel.any() else np.nan
        c1, c2 = _closest_site(m1, loc, strand), _closest_site(m2, loc, strand)
        if c1 == c2:
            continue                                   # both segment ends point at the same site
        # on "+" the site behind segment 2's end is the read's own, on "-" the one behind segment 1's end
        if first == c1:
            other = c2
        elif first == c2:
            other = c1
        else:
            continue
        if other is None:
            continue
        keep, drop = (first, other) if total[first] > total[other] else (other, first)
        drop = _resolve(drop, merged_into)
        if drop == keep:
            continue
        merged_into[drop] = keep
        total[keep] = total[keep] + total[drop]        # recal_pa_junc_pct :637-649
        junc[keep] = junc[keep] + junc[drop]
        total[drop] = 0
        junc[drop] = 0
        pct[keep] = junc[keep] / total[keep]
        pct[drop] = 0
    return merged_into, had_junction


def _apply_merges(ids, merged_into):
    """Series.replace(pa_dict) until no id is a key any more (:655-657): one simultaneous mapping per pass"""
    if not merged_into:
        return ids
    n = int(max(max(merged_into), max(merged_into.values()), int(ids.max()) if len(ids) else 0)) + 1
    lut = np.arange(n)
    for k, v in merged_into.items():
        lut[k] = v
    keys = np.fromiter(merged_into.keys(), dtype=np.int64)
    for _ in range(n + 1):
        if not np.isin(ids, keys).any():
            return ids
        ids = lut[ids]
    raise RuntimeError("merge_pa: cyclic site merges")


def merge_gene(inputs, results, gene_key, junction_pct_thres=JUNCTION_PCT_THRES,
               total_read_pct_thres=TOTAL_READ_PCT_THRES):
    """One gene (or gene:utr): {gene_info_str: input DataFrame}, {gene_info_str: Parameters} ->
    (merged Parameters, had_junction_site, changed)."""
    t = _gene_tables(inputs, results)
    first_key = list(results.keys())[0]
    # the handler (closest-site direction, alpha formula, output strand) follows the first record's strand
    # (_merge_pa :135); the genomic ordering above followed the last record's (:520, :572) - the same thing
    # unless a gene mixes strands
    strand = "+" if first_key[-1:] == "+" else "-"
    t.strand = strand
    n_site = len(t.loc)
    merged_into, had_junction = _merge_junction_sites(t, junction_pct_thres)
    site_ids = _apply_merges(np.arange(n_site, dtype=np.int64), merged_into)
    read_site = _apply_merges(t.label, merged_into)
    remain = np.array(list(set(site_ids)))             # :659 - the set's iteration order is the output order
    if remain.dtype.kind != "i":
        raise IndexError("merge_pa: gene " + str(gene_key) + " has no pA site (K = 0 in every record)")
    share = t.total[remain] / np.sum(t.total[remain])
    dropped, remain = remain[share <= total_read_pct_thres], remain[share > total_read_pct_thres]
    n_keep = len(remain)
    lut = np.arange(max(n_site, 1), dtype=np.int64)
    lut[remain] = np.arange(n_keep)
    lut[dropped] = n_keep
    new_label = lut[read_site] if len(read_site) else read_site
    on_site = new_label < n_keep
    loc = t.loc[remain]
    utr_st, utr_en = int(t.utr_st.min()), int(t.utr_en.max())
    para = Parameters(title="Final Result",
                      alpha_arr=(utr_en - loc + 1) if strand == "-" else (loc - utr_st),
                      beta_arr=t.beta[remain], ws=t.total[remain] / np.sum(t.total[remain]), L=0,
                      cb_id_arr=t.cb[on_site], readID_arr=t.read[on_site])
    para.label_arr = new_label[on_site]
    if len(str(gene_key).split(":")) == 1:
        para.gene_info_str = f"{t.chrom}:{t.gene}:1:{utr_st}-{utr_en}:{strand}"
    else:
        para.gene_info_str = f"{t.chrom}:{gene_key}:{utr_st}-{utr_en}:{strand}"
    return para, had_junction, 1 if merged_into else 0


# ---------------------------------------------------------------- driver (_merge_pa :44-147)
def _load_results(path):
    out = []
    for obj in read_input_chunk(path):
        out.append(obj)
    return out


def _key_of(gene_info_str, utr_merge):
    parts = gene_info_str.split(":")
    return parts[1] if utr_merge else ":".join(parts[1:3])


def _input_stream(path):
    """(gene_info_str, columns) of one input chunk: from its columnar copy when that is current, else the pickle"""
    from .binned import binned_path, is_current, read_binned
    if is_current(binned_path(path), path):
        return ((g, cols) for g, _b, cols in read_binned(binned_path(path)))
    return read_input_chunk(path)


def _merge_keys_task(args):
    """Worker: the gene keys of one result file, in order of first appearance."""
    path, utr_merge = args
    seen = {}
    for para in _load_results(path):
        seen.setdefault(_key_of(para.gene_info_str, utr_merge), None)
    return list(seen)


def _merge_file_task(args):
    """Worker: merge the genes that first appear in one chunk (their records may continue in other chunks, which
    are loaded too) and write them, in order, to a part file.  Returns the number of genes written."""
    output_dir, stems, genes, utr_merge, part_path = args
    wanted = set(genes)
    res_by_gene, in_by_gene = {g: {} for g in genes}, {g: {} for g in genes}
    for stem in stems:                                  # sorted file order = the serial path's insertion order
        for para in _load_results(os.path.join(output_dir, "pkl_output", stem + ".res.pkl")):
            k = _key_of(para.gene_info_str, utr_merge)
            if k in wanted:
                res_by_gene[k][para.gene_info_str] = para
        for gene_info_str, df in _input_stream(os.path.join(output_dir, "pkl_input", stem + ".input.pkl")):
            k = _key_of(gene_info_str, utr_merge)
            if k in wanted:
                in_by_gene[k][gene_info_str] = df
    with open(part_path, "wb") as fh:
        for g in genes:
            para, _junc, _change = merge_gene(in_by_gene[g], res_by_gene[g], g)
            pickle.dump(para, fh)
    return len(genes)


def _merge_pa(output_dir: str, utr_merge=True, workers=None):
    """Returns the number of merged records written.  workers: None = the shared prep pool (parallel over chunk
    files when there are at least four), 0 = in this process."""
    if not os.path.exists(os.path.join(output_dir, "pkl_output")):
        raise Exception("Please use the same directory that stores res pickle files by infer_pa")
    if not os.path.exists(os.path.join(output_dir, "pkl_input")):
        raise Exception("Please use the same directory that stores res pickle files by prepare_input")
    in_files = sorted(f for f in os.listdir(os.path.join(output_dir, "pkl_input")) if ".input.pkl" in f)
    out_files = sorted(f for f in os.listdir(os.path.join(output_dir, "pkl_output"))
                       if f.endswith(".res.pkl") and f[:-8] + ".input.pkl" in in_files)
    if len(in_files) != len(out_files):
        raise Exception("Number of *.res.pkl is different from number of *.input.pkl. Please make sure that all "
                        "input files are successfully used for infering PAS.")
    outfile = os.path.join(output_dir, "res.gene.pkl" if utr_merge else "res.utr.pkl")
    st = timer()
    try:
        os.remove(outfile)
    except OSError:
        pass
    stems = [f[:-8] for f in out_files]
    if workers == 0 or len(stems) < 4:
        n = _merge_file_task((output_dir, stems, _all_keys(output_dir, stems, utr_merge), utr_merge, outfile))
        print("Done read model output_dir")
        print("Done read model input")
        print(timer() - st)
        return n
    # parallel over chunk files: every gene belongs to the chunk it first appears in (file order, record order - the
    # order the serial path writes them in); the part files are concatenated in chunk order
    from .pipeline import shared_pool
    pool = shared_pool(workers)
    keys = list(pool.ex.map(_merge_keys_task, [(os.path.join(output_dir, "pkl_output", s + ".res.pkl"), utr_merge)
                                               for s in stems]))
    print("Done read model output_dir")
    first, files_of = {}, {}
    for i, ks in enumerate(keys):
        for k in ks:
            first.setdefault(k, i)
            files_of.setdefault(k, []).append(i)
    tasks, parts = [], []
    for i, ks in enumerate(keys):
        genes = [k for k in ks if first[k] == i]
        if not genes:
            continue
        need = sorted({j for k in genes for j in files_of[k]} | {i})
        parts.append(outfile + f".part{i}")
        tasks.append((output_dir, [stems[j] for j in need], genes, utr_merge, parts[-1]))
    print("Done read model input")
    counts = list(pool.ex.map(_merge_file_task, tasks))
    with open(outfile + ".part", "wb") as fh:
        for part in parts:
            with open(part, "rb") as src:
                while True:
                    blk = src.read(1 << 24)
                    if not blk:
                        break
                    fh.write(blk)
            os.remove(part)
    os.replace(outfile + ".part", outfile)
    print(timer() - st)
    return int(sum(counts))


def _all_keys(output_dir, stems, utr_merge):
    seen = {}
    for stem in stems:
        for k in _merge_keys_task((os.path.join(output_dir, "pkl_output", stem + ".res.pkl"), utr_merge)):
            seen.setdefault(k, None)
    return list(seen)


@click.command(name="merge_pa")
@click.option('--output_dir', type=str, required=True,
              help='Directory which was used in previous steps to save output by `prepare_input` and `infer_pa`')
@click.option('--utr_merge', type=bool, default=True,
              help='By default, True if want to process all pa site of one gene at together. False if want to process '
                   'each utr_file separately.')
@click.option('--workers', type=int, default=None,
              help='processes to spread the chunk files over (default: one per host core, at most 14; 0 = none)')
def merge_pa(output_dir: str, utr_merge=True, workers=None):
    """Merge junction-artefact pA sites per gene and write res.gene.pkl / res.utr.pkl
    (reference junction_handler.py:28-42)."""
    _merge_pa(output_dir, utr_merge, workers=workers)
