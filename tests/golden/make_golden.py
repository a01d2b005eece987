#!/usr/bin/env python3
"""Generates tests/golden/*.npz — inputs plus the outputs of the REFERENCE's own code run in the dev container.

Run here (needs /root/reference):   python tests/golden/make_golden.py
  1. builds oracle/_ref/ref_slice_harness (cmd_cram_demuxlet.cpp:390-881 + sc_drop_seq.cpp + PhredHelper.cpp + Error.cpp,
     compiled from /root/reference, recipe oracle/Makefile) and oracle/_ref/libref_units.so,
  2. draws seeded synthetic problems (demuxlet_amd/synth.py) and feeds them to the harness as a text spec,
  3. stores, per case, ONLY data: the inputs (events, float32 genotype matrix, parameters) and the reference's outputs
     (the four text files as bytes + raw binary64 arrays llks, llk0s, llksAB, llks00 + per-cell counters).
No reference source text is stored. The fixtures travel to the GPU box; /root/reference does not.
"""
import ctypes as C
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle_py as O          # noqa: E402
from demuxlet_amd import synth             # noqa: E402

OUT = Path(__file__).resolve().parent


def geno_matrix(rng, S, V, field, gt_error, missing_rate=0.0):
    raw = synth.make_raw_genotypes(rng, S, V, missing_rate)
    if field == "GT":
        g = np.stack([O.geno_from_gt(raw.alleles[s], gt_error) for s in range(S)])
    elif field == "GP":
        gp = synth.raw_gp_from_alleles(rng, raw.alleles)
        g = np.stack([O.geno_from_gp(gp[s], gt_error) for s in range(S)])
    elif field == "PL":
        pl = synth.raw_pl_from_alleles(rng, raw.alleles)
        g = np.stack([O.geno_from_pl(pl[s]) for s in range(S)])
    else:
        raise ValueError(field)
    return raw, g.astype(np.float32)


def make_case(name, seed, B, S, V, alphas, field, delta, rbar, gt_error=0.01, write_pair=False, min_snp=0, min_total=0,
              min_uniq=0, doublet_prior=0.5, missing_rate=0.0, empty_cells=0):
    rng = np.random.default_rng(seed)
    raw, g = geno_matrix(rng, S, V, field, gt_error, missing_rate)
    sp = synth.make_pileup(rng, np.where(raw.alleles < 0, 0, raw.alleles), B, delta, rbar, doublet_rate=0.3)
    bc, snp, umi, al, bq, new = synth.pileup_to_events(rng, sp)
    for k in range(empty_cells):         # cells whose reads overlap no SNP: .single rows only (cmd_cram_demuxlet.cpp:592)
        bc += [f"ZZEMPTY{k}-1"] * 3
        snp = np.concatenate([snp, [-1, -1, -1]]).astype(np.int32)
        umi += [".", ".", "."]
        al = np.concatenate([al, [0, 0, 0]]).astype(np.uint8)
        bq = np.concatenate([bq, [0, 0, 0]]).astype(np.uint8)
        new = np.concatenate([new, [1, 1, 1]]).astype(np.uint8)
    ev = O.Events(bc, snp, umi, al, bq, new)
    params = O.Params(tuple(alphas), doublet_prior, min_total, min_uniq, min_snp, write_pair)
    pb = O.Problem([f"SM{j:02d}" for j in range(V)], g, ev, params)
    return name, pb


def kat_case():
    G = np.stack([O.geno_from_gt(np.array(a), 0.01) for a in ([[0, 0], [1, 1]], [[0, 1], [0, 1]], [[1, 1], [0, 0]])])
    ev = O.Events(["AAA", "AAA", "AAA", "AAA", "CCC", "CCC", "CCC"], np.array([0, 1, 1, 2, 0, 2, 2], dtype=np.int32),
                  ["u1", "u1", "u2", "u3", "u1", "u1", "u2"], np.array([0, 1, 0, 1, 1, 0, 2], dtype=np.uint8),
                  np.array([30, 20, 40, 13, 30, 30, 30], dtype=np.uint8), np.ones(7, dtype=np.uint8))
    return "kat_micro", O.Problem(["S0", "S1"], G.astype(np.float32), ev, O.Params(write_pair=True))


def save_case(name, pb, tmp):
    ref = O.run_ref(pb, os.path.join(tmp, name))
    V = pb.n_samples
    A = len(pb.params.alphas)
    B = len(ref.barcodes)
    grid = np.zeros((B, V, V, A))
    l00 = np.zeros((B, A))
    proc = np.zeros(B, dtype=np.uint8)
    grid[ref.cell_ids] = ref.llksAB
    l00[ref.cell_ids] = ref.llks00
    proc[ref.cell_ids] = 1
    ev = pb.events
    p = pb.params
    np.savez_compressed(
        OUT / f"{name}.npz",
        sample_ids=np.array(pb.sample_ids), g=pb.g,
        ev_barcode=np.array(ev.barcode), ev_snp=ev.snp, ev_umi=np.array(ev.umi), ev_allele=ev.allele, ev_bq=ev.bq,
        ev_newread=ev.newread,
        alphas=np.array(p.alphas, dtype=np.float64), doublet_prior=np.float64(p.doublet_prior),
        min_total=np.int32(p.min_total), min_uniq=np.int32(p.min_uniq), min_snp=np.int32(p.min_snp),
        write_pair=np.int32(p.write_pair),
        ref_barcodes=np.array(ref.barcodes), ref_counters=ref.counters,
        ref_llks=ref.llks, ref_llk0s=ref.llk0s, ref_llksAB=grid, ref_llks00=l00, ref_processed=proc,
        **{f"file_{k}": np.frombuffer(v, dtype=np.uint8) for k, v in ref.files.items()},
    )
    print(f"{name}: B={B} V={V} A={A} events={len(ev.barcode)} best_rows={len(ref.cell_ids)} "
          f"-> {(OUT / (name + '.npz')).stat().st_size / 1024:.0f} KiB")


def phred_and_store_fixture():
    """Rows a2 and a1 straight from the reference translation units (no shim): oracle/_ref/libref_units.so."""
    L = C.CDLL(str(O.REF_UNITS))
    mat = np.zeros(256)
    err = np.zeros(256)
    L.ref_phred_tables(mat.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p))
    L.ref_phred_prob.restype = C.c_double
    prob = np.array([L.ref_phred_prob(q) for q in range(0, 300)])
    # a1: a scripted event stream with duplicates, conflicting duplicates, allele 2, several cells/SNPs
    rng = np.random.default_rng(77)
    n = 600
    cells = [f"BC{int(x):02d}" for x in rng.integers(0, 9, size=n)]
    snps = rng.integers(0, 12, size=n).astype(np.int32)
    umis = [f"{'ACGT'[int(a)]}{'ACGT'[int(b)]}{int(c)}" for a, b, c in zip(rng.integers(0, 4, n), rng.integers(0, 4, n), rng.integers(0, 3, n))]
    als = rng.integers(0, 3, size=n).astype(np.uint8)
    bqs = rng.integers(13, 41, size=n).astype(np.uint8)
    L.ref_scl_new.restype = C.c_void_p
    for f in (L.ref_scl_free, L.ref_scl_add_snp, L.ref_scl_ncells):
        f.argtypes = [C.c_void_p]
    L.ref_scl_add_cell.argtypes = [C.c_void_p, C.c_char_p]
    L.ref_scl_add_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
    L.ref_scl_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_scl_flatten_cell.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long]
    L.ref_scl_flatten_cell.restype = C.c_long
    st = L.ref_scl_new()
    for _ in range(12):
        L.ref_scl_add_snp(st)
    rets = np.zeros(n, dtype=np.uint8)
    ids = np.zeros(n, dtype=np.int32)
    for e in range(n):
        ids[e] = L.ref_scl_add_cell(st, cells[e].encode())
        rets[e] = L.ref_scl_add_read(st, int(snps[e]), int(ids[e]), umis[e].encode(), int(als[e]), int(bqs[e]))
    B = L.ref_scl_ncells(st)
    counters = np.zeros((B, 3), dtype=np.int32)
    flat_snp, flat_n, flat_w, cell_np = [], [], [], []
    for c in range(B):
        a, b, d = C.c_int(), C.c_int(), C.c_int()
        L.ref_scl_counters(st, c, C.byref(a), C.byref(b), C.byref(d))
        counters[c] = (a.value, b.value, d.value)
        sn = np.zeros(64, dtype=np.int32)
        nn = np.zeros(64, dtype=np.int32)
        ww = np.zeros(4096, dtype=np.uint32)
        k = L.ref_scl_flatten_cell(st, c, sn.ctypes.data, nn.ctypes.data, ww.ctypes.data, 64, 4096)
        assert k >= 0
        cell_np.append(k)
        flat_snp.append(sn[:k].copy())
        flat_n.append(nn[:k].copy())
        flat_w.append(ww[:int(nn[:k].sum())].copy())
    L.ref_scl_free(st)
    np.savez_compressed(OUT / "ref_units.npz", phred_mat=mat, phred_err=err, phred_prob=prob,
                        ev_cell=np.array(cells), ev_snp=snps, ev_umi=np.array(umis), ev_allele=als, ev_bq=bqs,
                        ret_new=rets, ret_cellid=ids, counters=counters, cell_npairs=np.array(cell_np, dtype=np.int32),
                        flat_snp=np.concatenate(flat_snp), flat_nper=np.concatenate(flat_n), flat_words=np.concatenate(flat_w))
    print("ref_units: phred tables + UMI-store trace ->", (OUT / "ref_units.npz").stat().st_size // 1024, "KiB")


def main():
    O.build()
    if not O.have_ref():
        sys.exit("needs /root/reference (dev container)")
    cases = [
        kat_case(),
        make_case("gt_v4_a2_pair", 101, B=50, S=400, V=4, alphas=(0.0, 0.5), field="GT", delta=0.3, rbar=1.6, write_pair=True,
                  missing_rate=0.1, empty_cells=2),
        make_case("gp_v8_a2_minsnp", 102, B=64, S=2000, V=8, alphas=(0.0, 0.5), field="GP", delta=0.05, rbar=1.3, min_snp=90,
                  min_uniq=100),
        make_case("pl_v32_a3", 103, B=10, S=1000, V=32, alphas=(0.0, 0.25, 0.5), field="PL", delta=0.2, rbar=2.0),
        make_case("gt_v64_a2", 104, B=6, S=500, V=64, alphas=(0.0, 0.5), field="GT", delta=0.4, rbar=1.25),
        make_case("gt_v3_alpha_quirk", 105, B=12, S=150, V=3, alphas=(0.25, 0.5, 0.75, 0.1), field="GT", delta=0.5, rbar=3.0,
                  write_pair=True, doublet_prior=0.3, gt_error=0.0),
        make_case("gt_v5_dense", 106, B=8, S=300, V=5, alphas=(0.0, 0.5), field="GT", delta=1.0, rbar=1.25, min_total=1),
        # wide panel x long alpha grid (j-slabs, alphas padded to 8 per pair), a 6-entry grid with --write-pair on a narrow
        # panel, and deep pairs (6 reads on average: GL seed tables + the read loop) with missing genotypes
        make_case("gp_v70_a5", 107, B=4, S=250, V=70, alphas=(0.0, 0.1, 0.25, 0.4, 0.5), field="GP", delta=0.3, rbar=1.5),
        make_case("pl_v12_a6_pair", 108, B=14, S=300, V=12, alphas=(0.0, 0.1, 0.2, 0.3, 0.4, 0.5), field="PL", delta=0.3, rbar=3.0,
                  write_pair=True, doublet_prior=0.2),
        make_case("gt_v24_a2_deep", 109, B=10, S=200, V=24, alphas=(0.0, 0.5), field="GT", delta=0.5, rbar=6.0, write_pair=True,
                  missing_rate=0.1),
        # the headline shape in small: soft field, 32 samples, default grid, dense layout (the A = 2 kernels of 17..32 samples and,
        # in FAST mode, the printed-entries kernel with its LDS-direct row loads); and a 100-sample soft-field panel (entry slabs)
        make_case("gp_v32_a2_dense", 110, B=12, S=160, V=32, alphas=(0.0, 0.5), field="GP", delta=1.0, rbar=1.25, write_pair=True),
        make_case("pl_v100_a2", 111, B=5, S=220, V=100, alphas=(0.0, 0.5), field="PL", delta=0.35, rbar=1.5),
    ]
    only = set(sys.argv[1:])                 # python make_golden.py [case ...]: regenerate just these
    if only:
        cases = [c for c in cases if c[0] in only]
    with tempfile.TemporaryDirectory() as tmp:
        for name, pb in cases:
            save_case(name, pb, tmp)
    if not only:
        phred_and_store_fixture()


if __name__ == "__main__":
    main()
