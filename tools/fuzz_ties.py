"""Randomised sweep of the TIE-HEAVY regime on a GPU box: panels with random groups of genotype-identical samples, barcodes that
cover a handful of SNPs (down to one), random alpha grids — through dmx_demuxlet_run (random engine count / range budget / write_pair, host or
device-resident pileup) AND through the gathered-records path (K3 records + flagged barcodes' device grids, and records + host re-evaluation)
against the oracle's files.  STRICT: byte-identical.  FAST (DMX_FUZZ_FAST=1): every string field identical, numbers within the last printed digit.
    python tools/fuzz_ties.py [n_cases] [seed]"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
from demuxlet_amd import build, capi, engine, synth
from oracle import oracle_py as O

build.build(); O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
fast = bool(os.environ.get("DMX_FUZZ_FAST"))
md = capi.DMX_MODE_FAST if fast else capi.DMX_MODE_STRICT


def same(got, want, what):
    a, b = Path(got).read_bytes(), Path(want).read_bytes()
    if a == b:
        return 0
    la, lb = a.decode().splitlines(), b.decode().splitlines()
    assert len(la) == len(lb), (what, len(la), len(lb))
    nd = 0
    for x, y in zip(la, lb):
        if x == y:
            continue
        assert fast, (what, x, y)                                  # STRICT: the oracle's bytes
        fx, fy = x.split("\t"), y.split("\t")
        assert len(fx) == len(fy), (what, x, y)
        for p, q in zip(fx, fy):
            if p != q:
                assert abs(float(p) - float(q)) <= 1e-3 * max(1e-3, abs(float(q))) + 1.01e-4, (what, x, y)
                nd += 1
    return nd


tot_flag = tot_cov = tot_nd = 0
for case in range(n_cases):
    V = int(rng.choice([2, 3, 4, 5, 8, 9, 16, 17, 24, 32, 33, 40, 64, 70]))
    A = int(rng.choice([2, 2, 2, 3, 5]))
    alphas = tuple([0.0] + sorted(rng.choice(np.arange(1, 50), size=A - 2, replace=False) / 100.0) + [0.5]) if A > 2 else (0.0, 0.5)
    if rng.random() < 0.1:
        alphas = tuple(sorted(rng.choice(np.arange(1, 51), size=A, replace=False) / 100.0))       # alpha[0] != 0
    field = str(rng.choice(["GT", "GT", "GP", "PL"]))
    if os.environ.get("DMX_FUZZ_CLSP"):          # the producer / consumer class kernel only: GT panels of 33..64 samples on the default grid
        V, A, alphas, field = int(rng.integers(33, 65)), 2, (0.0, 0.5), "GT"
    S = int(rng.integers(4, 90)); B = int(rng.integers(3, 50 if V <= 33 else 12))
    raw = synth.make_raw_genotypes(rng, S, V)
    al = raw.alleles.copy()
    # random groups of identical samples: every sample copies an earlier one with probability p_dup
    p_dup = float(rng.choice([0.0, 0.2, 0.5, 1.0]))
    src = list(range(V))
    for j in range(1, V):
        if rng.random() < p_dup:
            src[j] = src[int(rng.integers(0, j))]
            al[:, j] = al[:, src[j]]
    if field == "GT":
        g = np.stack([engine.geno_from_gt(al[s], 0.01) for s in range(S)])
    elif field == "GP":
        gp = synth.raw_gp_from_alleles(rng, al)
        for j in range(V): gp[:, j] = gp[:, src[j]]
        g = np.stack([engine.geno_from_gp(gp[s], 0.01) for s in range(S)])
    else:
        pl_ = synth.raw_pl_from_alleles(rng, al)
        for j in range(V): pl_[:, j] = pl_[:, src[j]]
        g = np.stack([engine.geno_from_pl(pl_[s]) for s in range(S)])
    sp = synth.make_pileup(rng, al, B, float(rng.uniform(0.02, 0.4)), float(rng.choice([1.0, 1.3, 3.0])), dense_layout=False, doublet_rate=0.4)
    pl = engine.HostPileup(sp.n_cells, sp.n_snps, sp.cell_pair_off, sp.cell_read_off, sp.pair_snp, sp.pair_nrd, sp.reads, sp.rd_totl, sp.rd_pass, sp.rd_uniq)
    bcs = [f"BC{(i * 7919) % 100003:06d}-1" for i in range(B)] if rng.random() < 0.7 else [f"BC{i:05d}-1" for i in range(B)]
    sms = [f"SM{j:02d}" for j in range(V)]
    wp = bool(rng.random() < 0.4)
    prior = float(rng.choice([0.5, 0.1]))
    words = ((pl.reads >> 7).astype(np.uint32) << 24) | ((pl.reads & 0x7F).astype(np.uint32) << 16) | 1
    csr = O.Csr(bcs, pl.cell_pair_off, pl.pair_snp, np.concatenate([[0], np.cumsum(pl.pair_nrd.astype(np.int64))]), words.astype(np.uint32),
                pl.rd_totl, pl.rd_pass, pl.rd_uniq)
    with tempfile.TemporaryDirectory() as td:
        O.run_csr(csr, sms, g, O.Params(alphas, prior, 0, 0, 0, wp), os.path.join(td, "ref"))
        n_gpus = int(rng.choice([1, 1, 2, 3]))
        rb = int(rng.choice([0, 0, 3000, 60000]))
        if rb: os.environ["DMX_RANGE_BYTES"] = str(rb)
        else: os.environ.pop("DMX_RANGE_BYTES", None)
        dev_pl = n_gpus == 1 and rng.random() < 0.4
        store = pl
        keep = None
        if dev_pl:                                                   # the same job from a device-resident pileup
            hs = pl.as_struct()
            dvc = torch.device("cuda", 0)
            keep = {k: torch.from_numpy(np.ascontiguousarray(getattr(pl, k))).to(dvc) for k in ("cell_pair_off", "cell_read_off", "pair_snp", "pair_nrd", "reads")}
            store = capi.Pileup(B, S, hs.n_pairs, hs.n_reads, keep["cell_pair_off"].data_ptr(), keep["cell_read_off"].data_ptr(), keep["pair_snp"].data_ptr(),
                                keep["pair_nrd"].data_ptr(), pl.pair_nrd.dtype.itemsize, capi.DMX_MEM_DEVICE, keep["reads"].data_ptr(),
                                pl.rd_totl.ctypes.data, pl.rd_pass.ctypes.data, pl.rd_uniq.ctypes.data)
        engine.demuxlet_run(store, g, sms, alphas, os.path.join(td, "run"), prior, write_pair=wp, arbiter=True, n_gpus=n_gpus, mode=md, barcodes=bcs)
        nd = 0
        for suf in ("single", "sing2", "best") + (("pair",) if wp else ()):
            nd += same(os.path.join(td, f"run.{suf}"), os.path.join(td, f"ref.{suf}"), (case, "dmx_demuxlet_run", suf, V, A, field, n_gpus, rb, dev_pl))
        e = engine.Engine(V, alphas, prior, mode=md)
        e.set_genotypes(g); e.set_pileup(pl)
        e.run()
        llks, llk0s = e.get_singlet()
        _, l00, summ = e.get_doublet(want_grid=False)
        sing = e.get_sing()
        near = engine.near_tie_cells(summ)
        grids = e.get_cell_grids(near)
        e.close()
        fa = engine.FinalArgs(bcs, sms, alphas, prior, pl.rd_totl, pl.rd_pass, pl.rd_uniq, pl.n_snp_per_cell)
        engine.write_doublet_summary(fa, sing, l00, summ, os.path.join(td, "rec"), tie_pileup=pl, tie_g=g, cell_grids={int(c): gr for c, gr in zip(near, grids)})
        engine.write_doublet_summary(fa, sing, l00, summ, os.path.join(td, "rech"), tie_pileup=pl, tie_g=g)
        for pre in ("rec", "rech"):
            for suf in ("sing2", "best"):
                nd += same(os.path.join(td, f"{pre}.{suf}"), os.path.join(td, f"ref.{suf}"), (case, pre, suf, V, A, field))
    cov = int((summ["n_pairs"] > 0).sum())
    tot_flag += len(near); tot_cov += cov; tot_nd += nd
    print(f"case {case:3d}: V={V:2d} A={A} {field} B={B:2d} S={S:2d} p_dup={p_dup:.1f} alpha0={alphas[0]:.2f} engines={n_gpus} range_bytes={rb} device_pileup={int(dev_pl)} "
          f"write_pair={int(wp)}: {len(near)} of {cov} barcodes flagged; files identical" + (f" up to {nd} last-digit differences" if fast else ""), flush=True)
print(f"{n_cases} cases ok ({'FAST' if fast else 'STRICT'}); {tot_flag} of {tot_cov} covered barcodes carried a near-tie flag; {tot_nd} printed numbers differed in the last digit")
