"""Loader for tests/golden/*.npz (data only: inputs + outputs of the reference's own code; see make_golden.py)."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = ["kat_micro", "gt_v4_a2_pair", "gp_v8_a2_minsnp", "pl_v32_a3", "gt_v64_a2", "gt_v3_alpha_quirk", "gt_v5_dense"]


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
