"""Barcode sharding over the GPUs of one node and the single gather of per-cell records.

Barcodes are independent (every accumulator of cmd_cram_demuxlet.cpp:576-734 belongs to one barcode), so the path shards
with NO data-path collective: rank r owns a contiguous range of the byte-wise sorted barcodes (the reference's output
order, :472,:576), balanced on the work the range carries, stages only that slice of the pileup, and runs the engine on
its own GPU.  The one collective is the gather of the fixed-size per-cell records to rank 0 at the end — over RCCL
(backend "nccl") on GPUs, over gloo in the CPU tests.  `--write-pair` grids are never gathered: every rank formats its
own barcode range and the text shards concatenate in rank order.

Barcodes whose record carries a near-tie flag (DMX_CELL_NEAR_DOUBLET / _NEAR_SINGLET: another sample pair, another alpha or a third
singlet within 1e-7 of a decision — duplicate samples in the panel, a handful of covered SNPs) cannot be decided from the record
alone: the reference's strict-< scans (cmd_cram_demuxlet.cpp:746-758,:799-814) need the contenders' exact values.  Their grids
(V*V*A doubles each) ride along in the SAME gather as extra rows behind the records (`near_grids`): rank 0 hands them to the
writer, whose tie arbiter re-evaluates the contenders with the host libm exactly as the single-process path does.

torch.distributed is plumbing here; nothing numerical happens in this module."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .engine import HostPileup


def sorted_barcode_order(barcodes: Sequence[str]) -> np.ndarray:
    """Cell ids in ascending byte-wise barcode order (std::map<std::string,int32_t>)."""
    enc = [b.encode() for b in barcodes]
    return np.array(sorted(range(len(enc)), key=lambda i: enc[i]), dtype=np.int64)


def balanced_ranges(cost: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Cut [0, len(cost)) into `world` contiguous ranges whose cost sums are as even as a prefix cut allows."""
    n = len(cost)
    csum = np.concatenate([[0.0], np.cumsum(np.asarray(cost, dtype=np.float64))])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(csum, target, side="left"))
        if i > 0 and abs(csum[i - 1] - target) <= abs(csum[min(i, n)] - target):
            i -= 1
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def cell_cost(n_pairs: np.ndarray, n_samples: int, n_alpha: int, doublet: bool = True) -> np.ndarray:
    """log() evaluations a cell costs: P_cell * (V+1) for the singlet stage, + P_cell * (V*V*A + A) for the grid."""
    per_pair = (n_samples + 1) + ((n_samples * n_samples * n_alpha + n_alpha) if doublet else 0)
    return np.asarray(n_pairs, dtype=np.float64) * per_pair + 1.0


def slice_pileup(pl: HostPileup, cells: np.ndarray) -> HostPileup:
    """The pileup restricted to `cells` (ids into pl), re-indexed 0..len(cells)-1 in that order."""
    cells = np.asarray(cells, dtype=np.int64)
    p0, p1 = pl.cell_pair_off[cells], pl.cell_pair_off[cells + 1]
    r0, r1 = pl.cell_read_off[cells], pl.cell_read_off[cells + 1]
    npair, nread = (p1 - p0), (r1 - r0)
    pidx = np.concatenate([np.arange(a, b) for a, b in zip(p0, p1)]) if len(cells) else np.zeros(0, dtype=np.int64)
    ridx = np.concatenate([np.arange(a, b) for a, b in zip(r0, r1)]) if len(cells) else np.zeros(0, dtype=np.int64)
    return HostPileup(len(cells), pl.n_snps, np.concatenate([[0], np.cumsum(npair)]).astype(np.int64),
                      np.concatenate([[0], np.cumsum(nread)]).astype(np.int64),
                      None if pl.pair_snp is None else pl.pair_snp[pidx], pl.pair_nrd[pidx], pl.reads[ridx],
                      pl.rd_totl[cells], pl.rd_pass[cells], pl.rd_uniq[cells])


@dataclass
class CellRecords:
    """What one rank contributes to the gather: fixed-size rows, one per cell of its shard (shard order)."""
    llks: np.ndarray        # [n][V]
    llk0s: np.ndarray       # [n]
    sing: np.ndarray        # [n][V]      llksAB[j][0][0]
    llks00: np.ndarray      # [n][A]
    summary: np.ndarray     # [n] capi.SUMMARY_DTYPE
    near_cells: Optional[np.ndarray] = None   # [m] shard-local ids of the near-tie-flagged cells (engine.near_tie_cells)
    near_grids: Optional[np.ndarray] = None   # [m][V][V][A] their grids (Engine.get_cell_grids)

    def as_matrix(self) -> np.ndarray:
        """One float64 row per cell (the summary struct is reinterpreted as 8-byte words)."""
        n = len(self.llk0s)
        sm = np.ascontiguousarray(self.summary).view(np.float64).reshape(n, -1)
        return np.ascontiguousarray(np.concatenate([self.llks, self.llk0s[:, None], self.sing, self.llks00, sm], axis=1))

    @staticmethod
    def from_matrix(m: np.ndarray, V: int, A: int, summary_dtype) -> "CellRecords":
        n = m.shape[0]
        o = 0
        llks = m[:, o:o + V]; o += V
        llk0s = m[:, o]; o += 1
        sing = m[:, o:o + V]; o += V
        l00 = m[:, o:o + A]; o += A
        sm = np.ascontiguousarray(m[:, o:]).view(summary_dtype).reshape(n)
        return CellRecords(np.ascontiguousarray(llks), np.ascontiguousarray(llk0s), np.ascontiguousarray(sing),
                           np.ascontiguousarray(l00), sm)


def gather_records(local: np.ndarray, counts: Sequence[int], device=None, dst: int = 0) -> Optional[np.ndarray]:
    """THE collective: every rank's [n_r][W] float64 record matrix -> rank `dst`, concatenated in rank order.
    Shards may differ in length; rows are padded to the longest shard for the fixed-size gather."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    nmax, W = max(counts), local.shape[1]
    buf = torch.zeros((nmax, W), dtype=torch.float64, device=device)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(buf.device)
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, outs, dst=dst)
    if rank != dst:
        return None
    return np.concatenate([outs[r][:counts[r]].cpu().numpy() for r in range(world)], axis=0)


def engine_compute(g: np.ndarray, alphas: Sequence[float], doublet_prior: float = 0.5, device: int = 0, mode: int = 0):
    """The `compute` of run_sharded on a GPU: one engine on `device`, the shard staged, K1 beside K2 -> K3 -> K3b, and what leaves the device
    is the fixed-size record per barcode plus the grids of the barcodes K3 flagged as near-ties (cmd_cram_demuxlet.cpp:412-461, :576-734)."""
    from . import engine as eng

    def compute(shard: HostPileup) -> CellRecords:
        e = eng.Engine(g.shape[1], alphas, doublet_prior, device=device, mode=mode)
        try:
            e.set_genotypes(g)
            e.set_pileup(shard)
            e.run()
            llks, llk0s = e.get_singlet()
            _, l00, summ = e.get_doublet(want_grid=False)
            sing = e.get_sing()
            near = eng.near_tie_cells(summ)
            grids = e.get_cell_grids(near)
        finally:
            e.close()
        return CellRecords(llks, llk0s, sing, l00, summ, near, grids)
    return compute


def write_from_records(order: np.ndarray, rec: CellRecords, pl: HostPileup, barcodes: Sequence[str], sample_ids: Sequence[str], alphas: Sequence[float],
                       out_prefix: str, g: Optional[np.ndarray] = None, doublet_prior: float = 0.5, min_total: int = 0, min_uniq: int = 0, min_snp: int = 0):
    """Rank 0 after run_sharded: .single / .sing2 / .best from the gathered records (sorted-barcode order -> cell ids), the flagged barcodes'
    grids handed to the writer; with `g` the tie arbiter walks the host pileup for the orders the device left open (:799-814)."""
    from . import engine as eng
    inv = np.empty_like(order)
    inv[order] = np.arange(len(order))
    fa = eng.FinalArgs(barcodes, sample_ids, alphas, doublet_prior, pl.rd_totl, pl.rd_pass, pl.rd_uniq, pl.n_snp_per_cell, min_total, min_uniq, min_snp, False)
    eng.write_single(fa, rec.llks[inv], rec.llk0s[inv], out_prefix + ".single")
    grids = {int(order[i]): gr for i, gr in zip(rec.near_cells, rec.near_grids)} if rec.near_cells is not None else None
    eng.write_doublet_summary(fa, rec.sing[inv], rec.llks00[inv], rec.summary[inv], out_prefix, tie_pileup=pl if g is not None else None, tie_g=g,
                              cell_grids=grids)


def run_sharded(pl: HostPileup, barcodes: Sequence[str], n_samples: int, n_alpha: int,
                compute: Callable[[HostPileup], CellRecords], summary_dtype, device=None):
    """Shard -> compute (one engine per rank) -> gather.  Returns on rank 0 (order, CellRecords over ALL cells in
    sorted-barcode order); None elsewhere.  `compute` maps a shard's pileup to its records (the GPU engine in the product;
    the tests inject the CPU oracle to exercise this plumbing under gloo)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    order = sorted_barcode_order(barcodes)
    ranges = balanced_ranges(cell_cost(pl.n_snp_per_cell[order], n_samples, n_alpha), world)
    lo, hi = ranges[rank]
    rec = compute(slice_pileup(pl, order[lo:hi]))
    m = rec.as_matrix()
    # the flagged cells' grids travel as extra rows of the same matrix: [shard-local id, grid entries ..., zero padding] cut into rows of
    # the record width; every rank announces how many such rows it appends (a tiny all_gather of one integer, no data-path collective)
    W = m.shape[1]
    nAB = n_samples * n_samples * n_alpha
    rows_per_grid = -(-(1 + nAB) // W)
    n_near = 0 if rec.near_cells is None else len(rec.near_cells)
    if n_near:
        extra = np.zeros((n_near, rows_per_grid * W), dtype=np.float64)
        extra[:, 0] = rec.near_cells
        extra[:, 1:1 + nAB] = np.asarray(rec.near_grids, dtype=np.float64).reshape(n_near, nAB)
        m = np.concatenate([m, extra.reshape(n_near * rows_per_grid, W)], axis=0)
    import torch
    mine = torch.tensor([n_near], dtype=torch.int64, device=device)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    near_counts = [int(x.item()) for x in allc]
    base = [b - a for a, b in ranges]
    full = gather_records(m, [base[r] + near_counts[r] * rows_per_grid for r in range(world)], device=device)
    if rank != 0:
        return None
    recs, cells, grids, off = [], [], [], 0
    for r in range(world):
        recs.append(full[off:off + base[r]])
        ex = full[off + base[r]:off + base[r] + near_counts[r] * rows_per_grid].reshape(near_counts[r], rows_per_grid * W)
        cells.append(ex[:, 0].astype(np.int64) + ranges[r][0])            # -> index into the sorted-barcode order
        grids.append(ex[:, 1:1 + nAB].reshape(near_counts[r], n_samples, n_samples, n_alpha))
        off += base[r] + near_counts[r] * rows_per_grid
    out = CellRecords.from_matrix(np.concatenate(recs, axis=0), n_samples, n_alpha, summary_dtype)
    out.near_cells = np.concatenate(cells) if cells else np.zeros(0, np.int64)
    out.near_grids = np.concatenate(grids, axis=0) if grids else np.zeros((0, n_samples, n_samples, n_alpha))
    return order, out
