"""Bench-scale synthetic pileups generated directly in HBM with torch (plumbing only: RNG + device memory).
Same generative model as demuxlet_amd/synth.py; the arrays come out in the C-ABI's dmx_pileup layout
(DMX_MEM_DEVICE) so nothing crosses PCIe before the timed region."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import capi


@dataclass
class DevicePileup:
    n_cells: int
    n_snps: int
    cell_pair_off: torch.Tensor      # int64 [B+1]
    cell_read_off: torch.Tensor      # int64 [B+1]
    pair_snp: Optional[torch.Tensor]  # int32 [P] or None (dense)
    pair_nrd: torch.Tensor           # uint8 [P]
    reads: torch.Tensor              # uint8 [R]
    truth: torch.Tensor              # int32 [B][2]

    @property
    def n_pairs(self) -> int: return int(self.pair_nrd.numel())
    @property
    def n_reads(self) -> int: return int(self.reads.numel())

    def as_struct(self) -> capi.Pileup:
        return capi.Pileup(self.n_cells, self.n_snps, self.n_pairs, self.n_reads, self.cell_pair_off.data_ptr(),
                           self.cell_read_off.data_ptr(), self.pair_snp.data_ptr() if self.pair_snp is not None else None,
                           self.pair_nrd.data_ptr(), 1, capi.DMX_MEM_DEVICE, self.reads.data_ptr(), None, None, None)

    def host_slice(self, first: int, n_cells: int):
        """Cells [first, first + n_cells) as numpy arrays, offsets rebased to 0 (for the bounded CPU baseline)."""
        po = self.cell_pair_off[first:first + n_cells + 1]
        ro = self.cell_read_off[first:first + n_cells + 1]
        p0, p1, r0, r1 = int(po[0].item()), int(po[-1].item()), int(ro[0].item()), int(ro[-1].item())
        return dict(n_cells=n_cells, n_snps=self.n_snps, cell_pair_off=(po - p0).cpu().numpy(),
                    cell_read_off=(ro - r0).cpu().numpy(),
                    pair_snp=None if self.pair_snp is None else self.pair_snp[p0:p1].cpu().numpy(),
                    pair_nrd=self.pair_nrd[p0:p1].cpu().numpy(), reads=self.reads[r0:r1].cpu().numpy())


def make_device_pileup(dosage: torch.Tensor, B: int, delta: float, rbar: float, seed: int, device: torch.device,
                       doublet_rate: float = 0.1, chunk_cells: int = 512, dense_layout: bool = True,
                       cell_id_base: int = 0) -> DevicePileup:
    """dosage: float32 [S][V] ALT-allele count of each sample (on `device`)."""
    S, V = dosage.shape
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    err_of_bq = torch.pow(10.0, -torch.arange(64, device=device, dtype=torch.float32) / 10.0)
    dense = delta >= 1.0
    pair_cnt, read_cnt, snp_chunks, nrd_chunks, rd_chunks, truth_chunks = [], [], [], [], [], []
    for c0 in range(0, B, chunk_cells):
        nc = min(chunk_cells, B - c0)
        s1 = ((torch.arange(c0, c0 + nc, device=device) + cell_id_base) % V).to(torch.int64)
        is_dbl = (torch.rand(nc, device=device, generator=gen) < doublet_rate) & (V > 1)
        s2 = (s1 + 1 + torch.randint(0, max(V - 1, 1), (nc,), device=device, generator=gen)) % V
        truth_chunks.append(torch.stack([s1, torch.where(is_dbl, s2, torch.full_like(s2, -1))], dim=1).to(torch.int32))
        if dense:
            cc = torch.arange(nc, device=device).repeat_interleave(S)
            ss = torch.arange(S, device=device).repeat(nc)
        else:
            cov = torch.rand((nc, S), device=device, generator=gen) < delta
            cc, ss = torch.nonzero(cov, as_tuple=True)
        npairs = cc.numel()
        lam = torch.full((npairs,), max(rbar - 1.0, 0.0), device=device, dtype=torch.float32)
        nreads = 1 + torch.poisson(lam, generator=gen).to(torch.int64)
        pair_of_read = torch.arange(npairs, device=device).repeat_interleave(nreads)
        rc, rs = cc[pair_of_read], ss[pair_of_read]
        nr = rc.numel()
        from_s2 = is_dbl[rc] & (torch.rand(nr, device=device, generator=gen) < 0.5)
        src = torch.where(from_s2, s2[rc], s1[rc])
        alt = torch.rand(nr, device=device, generator=gen) < dosage[rs, src] * 0.5
        bq = torch.randint(13, 41, (nr,), device=device, generator=gen)
        e = torch.rand(nr, device=device, generator=gen) < err_of_bq[bq]
        u = torch.randint(0, 3, (nr,), device=device, generator=gen)
        alt_i = alt.to(torch.int64)
        allele = torch.where(e, torch.where(u == 0, 1 - alt_i, torch.full_like(alt_i, 2)), alt_i)
        keep = allele != 2
        kept_per_pair = torch.bincount(pair_of_read[keep], minlength=npairs)
        assert int(kept_per_pair.max().item()) <= 255
        pair_cnt.append(torch.bincount(cc, minlength=nc))
        read_cnt.append(torch.bincount(rc[keep], minlength=nc))
        if not dense or not dense_layout:
            snp_chunks.append(ss.to(torch.int32))
        nrd_chunks.append(kept_per_pair.to(torch.uint8))
        rd_chunks.append(((allele[keep] << 7) | bq[keep]).to(torch.uint8))
    zero = torch.zeros(1, dtype=torch.int64, device=device)
    cell_pair_off = torch.cat([zero, torch.cumsum(torch.cat(pair_cnt), 0)])
    cell_read_off = torch.cat([zero, torch.cumsum(torch.cat(read_cnt), 0)])
    return DevicePileup(B, S, cell_pair_off.contiguous(), cell_read_off.contiguous(),
                        torch.cat(snp_chunks).contiguous() if snp_chunks else None, torch.cat(nrd_chunks).contiguous(),
                        torch.cat(rd_chunks).contiguous(), torch.cat(truth_chunks))


class _RawDevArray:
    """__cuda_array_interface__ shim so torch can view a raw device pointer owned by libdmx without a copy."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def tensor_from_ptr(ptr: int, shape, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_RawDevArray(ptr, shape, typestr), device=device)
