"""One random problem of tools/fuzz_parity.py's sweep (its generator, kept apart so that a case the sweep stopped at can be replayed from
its seed and committed as a fixture: tools/extract_fuzz_case.py)."""
import os
import numpy as np
from demuxlet_amd import engine, synth


def gen_case(rng, scale=1):
    V = int(rng.choice([2, 3, 5, 8, 9, 16, 17, 20, 31, 32, 33, 48, 64, 65, 90, 128, 129, 160, 257]))
    A = int(rng.choice([2, 2, 2, 3, 4, 5, 6, 8, 9]))
    alphas = tuple([0.0] + sorted(rng.choice(np.arange(1, 50), size=A - 2, replace=False) / 100.0) + [0.5]) if A > 2 else (0.0, 0.5)
    if rng.random() < 0.15:
        alphas = tuple(sorted(rng.choice(np.arange(0, 51), size=A, replace=False) / 100.0))       # alpha[0] != 0, no 0.5
    field = str(rng.choice(["GT", "GT", "GP", "PL"]))
    if os.environ.get("DMX_FUZZ_CLSP"):          # the producer / consumer class kernel only: GT panels of 33..64 samples on the default grid
        V, A, alphas, field = int(rng.integers(33, 65)), 2, (0.0, 0.5), "GT"
    dense = bool(rng.random() < 0.3)
    S = int(rng.integers(5, 150 if V > 64 else 400 * (scale if V <= 32 else 1)))
    B = int(rng.integers(1, 6 if V > 64 else 40 * (scale if V <= 32 else 1)))
    delta = 1.0 if dense else float(rng.uniform(0.02, 0.6))
    rbar = float(rng.choice([1.0, 1.25, 2.0, 4.0, 9.0]))
    if os.environ.get("DMX_FUZZ_DEEP") and V <= 16:            # hundreds of reads per pair: 16-bit counts, the plain-division path
        rbar = float(rng.choice([40.0, 300.0])); S = min(S, 40); B = min(B, 6)
    missing = float(rng.choice([0.0, 0.0, 0.1]))
    raw = synth.make_raw_genotypes(rng, S, V, missing_rate=missing if field == "GT" else 0.0)
    al = np.where(raw.alleles < 0, 0, raw.alleles)
    if field == "GT":
        g = np.stack([engine.geno_from_gt(raw.alleles[s], 0.01) for s in range(S)])
    elif field == "GP":
        gp = synth.raw_gp_from_alleles(rng, al)
        g = np.stack([engine.geno_from_gp(gp[s], 0.01) for s in range(S)])
    else:
        plv = synth.raw_pl_from_alleles(rng, al)
        g = np.stack([engine.geno_from_pl(plv[s]) for s in range(S)])
    sp = synth.make_pileup(rng, al, B, delta, rbar, dense_layout=dense, doublet_rate=0.3)
    if rng.random() < 0.3:                       # a wider range of base qualities than the generator's 13..40
        sp.reads[:] = (sp.reads & 0x80) | rng.integers(0, 94, size=len(sp.reads)).astype(np.uint8)
    return dict(V=V, A=A, alphas=alphas, field=field, dense=dense, S=S, B=B, rbar=rbar, g=g, sp=sp)
