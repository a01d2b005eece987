"""Loader for tests/golden/*.npz (data only: inputs + outputs of the reference's own code; see make_golden.py)."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = ["kat_micro", "gt_v4_a2_pair", "gp_v8_a2_minsnp", "pl_v32_a3", "gt_v64_a2", "gt_v3_alpha_quirk", "gt_v5_dense",
         "gp_v70_a5", "pl_v12_a6_pair", "gt_v24_a2_deep", "gp_v32_a2_dense", "pl_v100_a2"]


class Golden:
    def __init__(self, name):
        z = np.load(GOLDEN / f"{name}.npz")
        self.name = name
        self.z = z
        self.sample_ids = [str(s) for s in z["sample_ids"]]
        self.g = z["g"]
        self.alphas = tuple(float(a) for a in z["alphas"])
        self.doublet_prior = float(z["doublet_prior"])
        self.min_total, self.min_uniq, self.min_snp = int(z["min_total"]), int(z["min_uniq"]), int(z["min_snp"])
        self.write_pair = bool(int(z["write_pair"]))
        self.ref_barcodes = [str(s) for s in z["ref_barcodes"]]
        self.files = {k[5:]: z[k].tobytes() for k in z.files if k.startswith("file_")}

    def problem(self, O):
        z = self.z
        ev = O.Events([str(s) for s in z["ev_barcode"]], z["ev_snp"], [str(s) for s in z["ev_umi"]], z["ev_allele"],
                      z["ev_bq"], z["ev_newread"])
        return O.Problem(self.sample_ids, self.g, ev,
                         O.Params(self.alphas, self.doublet_prior, self.min_total, self.min_uniq, self.min_snp, self.write_pair))


def summary_from_grid(grid, l00, alphas, prior, n_pairs, summary_dtype):
    """The per-cell record K3 produces, computed with the reference's own scans (cmd_cram_demuxlet.cpp:713-734,
    :746-758, :799-828) in numpy/python — test-side restatement for index-exact comparisons."""
    import numpy as np
    V, _, A = grid.shape
    s = np.zeros((), dtype=summary_dtype)
    mx = -1e300
    for v in grid.ravel():
        if mx < v: mx = v
    ss = 0.0; sd = 0.0
    for j in range(V):
        ss += (np.exp(grid[j, 0, 0] - mx) * (1. - prior) / V)
        for k in range(V):
            if j == k: continue
            for n in range(1, A):
                sd += (np.exp(grid[j, k, n] - mx) * prior / V / (V - 1) / (A - 1) / (2.0 if alphas[n] == 0.5 else 1.0))
    i1 = i2 = -1; m1 = m2 = -1e300
    for j in range(V):
        v = grid[j, 0, 0]
        if m1 < v: m2, i2, i1, m1 = m1, i1, j, v
        elif m2 < v: i2, m2 = j, v
    jb = kb = nb = -1; mab = -1e300
    for j in range(V):
        for k in range(V):
            if j == k: continue
            for n in range(1, A):
                if mab < grid[j, k, n]: jb, kb, nb, mab = j, k, n, grid[j, k, n]
    s["max_llk"], s["sum_single"], s["sum_double"] = mx, ss, sd
    s["i_sing1"], s["i_sing2"], s["j_best"], s["k_best"], s["n_best"] = i1, i2, jb, kb, nb
    s["sing_llk1"], s["sing_llk2"] = grid[i1, 0, 0], grid[i2, 0, 0]
    s["llk12"], s["llk1"], s["llk2"] = grid[jb, kb, nb], grid[jb, 0, 0], grid[kb, 0, 0]
    s["llk10"], s["llk20"] = grid[jb, 0, nb], grid[kb, 0, nb]
    s["llk00_0"], s["llk00_best"] = l00[0], l00[nb]
    s["n_pairs"] = n_pairs
    return s


def printed_mask(V, A):
    """Grid entries demuxlet prints or decides on (cmd_cram_demuxlet.cpp:726-733,:746-797,:799-828): the singlet column
    llksAB[j][0][0] and every entry of the doublet alphas n >= 1.  llksAB[j][k][0] for k != 0 is read by the maxLLK scan only
    (:713-721) — DMX_MODE_FAST does not compute those (it stores [j][0][0] there)."""
    m = np.zeros((V, V, A), dtype=bool)
    m[:, 0, 0] = True
    m[:, :, 1:] = True
    return m
