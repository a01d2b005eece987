"""Randomised END-TO-END sweep on a GPU box: BAM-ordered events -> dmx_store -> dmx_demuxlet_run (random engine count, random
range budget, tie arbiter on) against the oracle's four files.  BEST calls and every string field must be identical; printed
numbers may differ in their last printed digit only.      python tools/fuzz_e2e.py [n_cases] [seed]"""
import os, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
from demuxlet_amd import build, engine, synth
from oracle import oracle_py as O

build.build(); O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
tot_diff = 0
for case in range(n_cases):
    V = int(rng.choice([2, 3, 4, 8, 12, 16, 24, 32, 40, 64, 70]))
    A = int(rng.choice([2, 2, 2, 3, 5]))
    alphas = tuple([0.0] + sorted(rng.choice(np.arange(1, 50), size=A - 2, replace=False) / 100.0) + [0.5]) if A > 2 else (0.0, 0.5)
    field = str(rng.choice(["GT", "GP", "PL"]))
    if os.environ.get("DMX_FUZZ_CLSP"):          # the producer / consumer class kernel only: GT panels of 33..64 samples on the default grid
        V, A, alphas, field = int(rng.integers(33, 65)), 2, (0.0, 0.5), "GT"
    S = int(rng.integers(20, 300)); B = int(rng.integers(2, 30 if V <= 32 else 8))
    raw = synth.make_raw_genotypes(rng, S, V, missing_rate=0.1 if (field == "GT" and rng.random() < 0.5) else 0.0)
    al = np.where(raw.alleles < 0, 0, raw.alleles)
    if field == "GT":
        g = np.stack([engine.geno_from_gt(raw.alleles[s], 0.01) for s in range(S)])
    elif field == "GP":
        g = np.stack([engine.geno_from_gp(x, 0.01) for x in synth.raw_gp_from_alleles(rng, al)])
    else:
        g = np.stack([engine.geno_from_pl(x) for x in synth.raw_pl_from_alleles(rng, al)])
    sp = synth.make_pileup(rng, al, B, float(rng.uniform(0.05, 0.6)), float(rng.choice([1.0, 1.5, 3.0])), dense_layout=False, doublet_rate=0.4)
    bc, snp, umi, allele, bq, newread = synth.pileup_to_events(rng, sp)
    ev = O.Events(bc, snp, umi, allele, bq, newread)
    params = O.Params(alphas, float(rng.choice([0.5, 0.1])), int(rng.choice([0, 0, 30])), int(rng.choice([0, 0, 10])), int(rng.choice([0, 0, 5])),
                      bool(rng.random() < 0.5))
    sm = [f"SM{j:03d}" for j in range(V)]
    with tempfile.TemporaryDirectory() as td:
        O.run_problem(O.Problem(sm, g, ev, params), os.path.join(td, "ref"))
        st = engine.Store()
        for _ in range(S):
            st.add_snp()
        for e in range(len(bc)):
            c = st.add_cell(bc[e])
            if newread[e]:
                st.count_read(c)
            if snp[e] >= 0:
                st.add_read(int(snp[e]), c, umi[e], int(allele[e]), int(bq[e]))
        n_gpus = int(rng.choice([1, 1, 2, 3]))
        rb = int(rng.choice([0, 0, 2000, 50000]))
        if rb: os.environ["DMX_RANGE_BYTES"] = str(rb)
        else: os.environ.pop("DMX_RANGE_BYTES", None)
        os.environ["DMX_THREADS"] = str(int(rng.choice([1, 3, 8])))
        engine.demuxlet_run(st, g, sm, alphas, os.path.join(td, "got"), params.doublet_prior, params.min_total, params.min_uniq, params.min_snp,
                            params.write_pair, arbiter=True, n_gpus=n_gpus,
                            mode=engine.capi.DMX_MODE_FAST if os.environ.get("DMX_FUZZ_FAST") else engine.capi.DMX_MODE_STRICT)
        ndiff = 0
        for suf in ("single", "sing2", "best") + (("pair",) if params.write_pair else ()):
            got = open(os.path.join(td, f"got.{suf}")).read().splitlines()
            want = open(os.path.join(td, f"ref.{suf}")).read().splitlines()
            assert len(got) == len(want), (case, suf, len(got), len(want))
            for a, b in zip(got, want):
                if a == b:
                    continue
                fa, fb = a.split("\t"), b.split("\t")
                assert len(fa) == len(fb), (case, suf, a, b)
                for x, y in zip(fa, fb):
                    if x == y:
                        continue
                    fx, fy = float(x), float(y)          # strings (barcodes, ids, BEST) must have been equal: float() raises otherwise
                    assert abs(fx - fy) <= 1e-3 * max(1e-3, abs(fy)) + 1.01e-4, (case, suf, a, b)
                    ndiff += 1
        tot_diff += ndiff
    print(f"case {case:3d}: V={V:2d} A={A} {field} B={B:2d} S={S:3d} engines={n_gpus} range_bytes={rb} write_pair={int(params.write_pair)}: "
          f"files equal up to {ndiff} last-digit differences", flush=True)
print(f"{n_cases} cases ok; {tot_diff} printed numbers differed in the last digit")
