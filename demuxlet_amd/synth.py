"""Synthetic pileups for the BASELINE.json configs.

Generative model (ours; the reference ships no generator): per SNP AF ~ U(0.05,0.95), per-sample genotype ~
Binomial(2, AF); a cell is a singlet of sample (c mod V) w.p. 0.9, else a 50/50 doublet with a second, different
sample; each (cell, SNP) is covered w.p. delta; a covered pair carries 1+Poisson(rbar-1) unique UMIs; a read shows ALT
w.p. g/2 of its source sample, bq ~ U{13..40}, and a base error w.p. 10^(-bq/10) moves it to one of the three other
bases (so allele 2 = "other" occurs).

Two back-ends:
  * numpy  (host; tests, fixtures, small problems, events with strings for the UMI-store path)
  * torch  (device; bench-scale problems are generated directly in HBM so nothing crosses PCIe)
The packed read byte is the C-ABI's: (allele<<7)|bq with allele in {0,1}; allele-2 reads are dropped from the byte
stream (they are skipped by both likelihood loops, cmd_cram_demuxlet.cpp:435,:604) but still create the pair.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class RawGeno:
    alleles: np.ndarray            # int32 [S][V][2]   (-1 = missing)
    af: np.ndarray                 # float64 [S]


def make_raw_genotypes(rng: np.random.Generator, S: int, V: int, missing_rate: float = 0.0) -> RawGeno:
    af = rng.uniform(0.05, 0.95, size=S)
    a = (rng.random((S, V, 2)) < af[:, None, None]).astype(np.int32)
    a.sort(axis=2)
    if missing_rate > 0:
        miss = rng.random((S, V)) < missing_rate
        a[miss] = -1
    return RawGeno(a, af)


def raw_gp_from_alleles(rng: np.random.Generator, alleles: np.ndarray, soft: float = 0.05) -> np.ndarray:
    """'softened one-hot' raw GP field (SURVEY §8d cfg 3): un-normalised floats, mostly on the true genotype."""
    S, V, _ = alleles.shape
    gt = np.clip(alleles, 0, 1).sum(axis=2)
    gp = rng.uniform(0.0, soft, size=(S, V, 3)).astype(np.float32)
    np.put_along_axis(gp, gt[..., None], (1.0 - rng.uniform(0, soft, size=(S, V, 1))).astype(np.float32), axis=2)
    return gp


def raw_pl_from_alleles(rng: np.random.Generator, alleles: np.ndarray) -> np.ndarray:
    """PL field (SURVEY §8d cfg 5): PL[true genotype]=0, others U{10..255}."""
    S, V, _ = alleles.shape
    gt = np.clip(alleles, 0, 1).sum(axis=2)
    pl = rng.integers(10, 256, size=(S, V, 3)).astype(np.int32)
    np.put_along_axis(pl, gt[..., None], 0, axis=2)
    return pl


@dataclass
class SynthPileup:
    """CSR pileup in the C-ABI's layout (cells in id order)."""
    n_cells: int
    n_snps: int
    cell_pair_off: np.ndarray      # int64 [B+1]
    cell_read_off: np.ndarray      # int64 [B+1]
    pair_snp: Optional[np.ndarray]  # int32 [P]  (None: dense, pair index within a cell == snp id)
    pair_nrd: np.ndarray           # uint8/uint16 [P]  reads kept per pair (allele 2 dropped)
    reads: np.ndarray              # uint8 [R]   (allele<<7)|bq
    rd_totl: np.ndarray            # int32 [B]
    rd_pass: np.ndarray
    rd_uniq: np.ndarray
    truth: np.ndarray              # int32 [B][2]  true (sample1, sample2 or -1)


def make_pileup(rng: np.random.Generator, alleles: np.ndarray, B: int, delta: float, rbar: float,
                dense_layout: bool = False, doublet_rate: float = 0.1, chunk_cells: int = 256) -> SynthPileup:
    S, V, _ = alleles.shape
    dosage = np.clip(alleles, 0, 1).sum(axis=2).astype(np.float64)      # [S][V]
    err_of_bq = np.power(10.0, -np.arange(64) / 10.0)
    cell_pair_off = np.zeros(B + 1, dtype=np.int64)
    cell_read_off = np.zeros(B + 1, dtype=np.int64)
    snp_chunks, nrd_chunks, rd_chunks = [], [], []
    totl = np.zeros(B, dtype=np.int32)
    uniq = np.zeros(B, dtype=np.int32)
    truth = np.full((B, 2), -1, dtype=np.int32)
    for c0 in range(0, B, chunk_cells):
        c1 = min(B, c0 + chunk_cells)
        nc = c1 - c0
        s1 = (np.arange(c0, c1) % V).astype(np.int32)
        is_dbl = (rng.random(nc) < doublet_rate) & (V > 1)
        s2 = (s1 + 1 + rng.integers(0, max(V - 1, 1), size=nc)) % V
        truth[c0:c1, 0] = s1
        truth[c0:c1, 1] = np.where(is_dbl, s2, -1)
        if delta >= 1.0:
            cov = np.ones((nc, S), dtype=bool)
        else:
            cov = rng.random((nc, S)) < delta
        cc, ss = np.nonzero(cov)                                        # row-major: cell-major, snp ascending
        npairs = len(cc)
        nreads = 1 + rng.poisson(max(rbar - 1.0, 0.0), size=npairs)
        pair_of_read = np.repeat(np.arange(npairs), nreads)
        rc, rs = cc[pair_of_read], ss[pair_of_read]
        src = np.where(is_dbl[rc] & (rng.random(len(rc)) < 0.5), s2[rc], s1[rc])
        alt = rng.random(len(rc)) < dosage[rs, src] / 2.0
        bq = rng.integers(13, 41, size=len(rc)).astype(np.uint8)
        e = rng.random(len(rc)) < err_of_bq[bq]
        u = rng.integers(0, 3, size=len(rc))
        allele = np.where(e, np.where(u == 0, 1 - alt.astype(np.int32), 2), alt.astype(np.int32)).astype(np.uint8)
        keep = allele != 2
        kept_per_pair = np.bincount(pair_of_read[keep], minlength=npairs)
        per_cell_pairs = np.bincount(cc, minlength=nc)
        per_cell_reads_all = np.bincount(rc, minlength=nc)
        per_cell_reads_kept = np.bincount(rc[keep], minlength=nc)
        cell_pair_off[c0 + 1:c1 + 1] = per_cell_pairs
        cell_read_off[c0 + 1:c1 + 1] = per_cell_reads_kept
        totl[c0:c1] = per_cell_reads_all
        uniq[c0:c1] = per_cell_reads_all
        snp_chunks.append(ss.astype(np.int32))
        nrd_chunks.append(kept_per_pair)
        rd_chunks.append(((allele[keep] << 7) | bq[keep]).astype(np.uint8))
    np.cumsum(cell_pair_off, out=cell_pair_off)
    np.cumsum(cell_read_off, out=cell_read_off)
    nrd = np.concatenate(nrd_chunks) if nrd_chunks else np.zeros(0, dtype=np.int64)
    nrd = nrd.astype(np.uint8 if (len(nrd) == 0 or nrd.max() <= 255) else np.uint16)
    pair_snp = np.concatenate(snp_chunks) if snp_chunks else np.zeros(0, dtype=np.int32)
    reads = np.concatenate(rd_chunks) if rd_chunks else np.zeros(0, dtype=np.uint8)
    use_dense = dense_layout and delta >= 1.0
    return SynthPileup(B, S, cell_pair_off, cell_read_off, None if use_dense else pair_snp, nrd, reads,
                       totl, totl.copy(), uniq, truth)


def barcode_name(i: int) -> str:
    """Deterministic 16-mer barcode whose byte-wise sort order is NOT the id order (exercises the sorted-output rule)."""
    x = (i * 2654435761 + 12345) & 0xFFFFFFFF
    s = []
    for _ in range(16):
        s.append("ACGT"[x & 3])
        x = (x >> 2) | ((x & 3) << 30)
    return "".join(s) + "-1"


def pileup_to_events(rng: np.random.Generator, sp: SynthPileup, shuffle: bool = True, dup_rate: float = 0.1,
                     other_rate: float = 0.03, extra_read_rate: float = 0.2):
    """Turn a CSR pileup into a BAM-ordered event list for the UMI-store path (row a1): UMIs are strings whose sort
    order differs from arrival order, a fraction of reads are PCR duplicates (same UMI, possibly a different base —
    the first one must win), some reads carry allele 2, and some reads overlap no SNP (RD.TOTL only).
    Returns (barcode list, snp, umi list, allele, bq, newread)."""
    B = sp.n_cells
    bcs, snps, umis, als, bqs = [], [], [], [], []
    for c in range(B):
        p0, p1 = int(sp.cell_pair_off[c]), int(sp.cell_pair_off[c + 1])
        r = int(sp.cell_read_off[c])
        for p in range(p0, p1):
            snp = int(sp.pair_snp[p]) if sp.pair_snp is not None else p - p0
            n = int(sp.pair_nrd[p])
            ids = rng.permutation(n + 2)[:n] if n else []
            for t in range(n):
                byte = int(sp.reads[r + t])
                umi = f"U{int(ids[t]) * 7919 % 1000:03d}{'ACGT'[int(ids[t]) % 4]}"
                bcs.append(c); snps.append(snp); umis.append(umi); als.append(byte >> 7); bqs.append(byte & 0x7F)
                if rng.random() < dup_rate:        # PCR duplicate arriving later with a different observation
                    bcs.append(c); snps.append(snp); umis.append(umi); als.append(int(rng.integers(0, 3))); bqs.append(int(rng.integers(13, 41)))
            if n == 0 or rng.random() < other_rate:  # an "other base" read: keeps the pair alive, never enters a likelihood
                bcs.append(c); snps.append(snp); umis.append(f"X{len(umis) % 97:02d}"); als.append(2); bqs.append(int(rng.integers(13, 41)))
            r += n
    n_ev = len(bcs)
    order = rng.permutation(n_ev) if shuffle else np.arange(n_ev)
    # duplicates must stay AFTER their original: keep relative order of equal (cell,snp,umi) keys
    seq = np.empty(n_ev, dtype=np.int64)
    for pos, e in enumerate(order):
        seq[e] = pos
    groups = {}
    for e in range(n_ev):
        groups.setdefault((bcs[e], snps[e], umis[e]), []).append(e)
    for k, es in groups.items():
        if len(es) > 1:
            ps = sorted(seq[e] for e in es)
            for e, p in zip(es, ps):
                seq[e] = p
    order = np.argsort(seq, kind="stable")
    names = [barcode_name(c) for c in range(B)]
    out_bc, out_snp, out_umi, out_al, out_bq, out_new = [], [], [], [], [], []
    for e in order:
        out_bc.append(names[bcs[e]]); out_snp.append(snps[e]); out_umi.append(umis[e]); out_al.append(als[e]); out_bq.append(bqs[e])
        out_new.append(0 if rng.random() < 0.1 else 1)   # 0: same read as the previous event (a read spanning two SNPs)
        if rng.random() < extra_read_rate:   # a read of the same cell that overlaps no SNP
            out_bc.append(names[bcs[e]]); out_snp.append(-1); out_umi.append("."); out_al.append(0); out_bq.append(0); out_new.append(1)
    return (out_bc, np.array(out_snp, dtype=np.int32), out_umi, np.array(out_al, dtype=np.uint8),
            np.array(out_bq, dtype=np.uint8), np.array(out_new, dtype=np.uint8))
