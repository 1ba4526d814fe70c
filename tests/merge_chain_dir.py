"""The synthetic `output_dir` of the infer_pa -> merge_pa chain test (inputs only; shared by
tests/golden/make_golden.py, which runs the REFERENCE's merge_pa on the oracle's fits of it, and by
tests/test_gpu_parity.py, which runs this build's `infer_pa_all` (GPU) + `merge_pa` on it).

Two chunk files, five UTR records each, 400 reads per record with junction reads; records (2g, 2g + 1) belong to one
gene (UTR ids 1 and 2), every third gene lies on the minus strand.  `infer_pa_all` in rng_mode per_utr draws UTR j of a
file from RandomState(SEED + j)."""
import os
import pickle

N_FILES, PER_FILE, READS, K_CAP, BASE_SEED, SEED = 2, 5, 400, 3, 4242, 11
KW = dict(n_max_apa=3, n_min_apa=1)


def records():
    """[(file index, j in file, gene_info_str, DataFrame)] in file order."""
    from scape_amd.synth import synth_utr
    out = []
    for i in range(N_FILES * PER_FILE):
        _g, df, truth = synth_utr(i, READS, k_cap=K_CAP, base_seed=BASE_SEED, pa_rate=0.03)
        gene, utr = i // 2, i % 2 + 1
        strand = "-" if gene % 3 == 2 else "+"
        start = 1000 * gene + 1
        g = f"syn:M{gene:05d}:{utr}:{start}-{start + truth['L_true'] - 1}:{strand}"
        df = df.copy()
        for c in ("seg1_en", "seg2_en"):                      # genomic coordinates of the segment ends
            df[c] = df[c] + (start - 1)
        out.append((i // PER_FILE, i % PER_FILE, g, df))
    return out


def stem(fi):
    return f"chain.{PER_FILE}.{N_FILES}.{fi + 1}"


def write_inputs(root):
    os.makedirs(os.path.join(root, "pkl_input"), exist_ok=True)
    os.makedirs(os.path.join(root, "pkl_output"), exist_ok=True)
    fhs = [open(os.path.join(root, "pkl_input", stem(fi) + ".input.pkl"), "wb") for fi in range(N_FILES)]
    recs = records()
    for fi, _j, g, df in recs:
        pickle.dump((g, df), fhs[fi])
    for fh in fhs:
        fh.close()
    return recs
