"""Host-side mirror of the reference's interface for the likelihood path, over the C-ABI (demuxlet_amd/capi.py).

Names follow the reference: `Store` is sc_dropseq_lib_t (add_snp/add_cell/add_read, sc_drop_seq.h:34-58); `Engine` runs what
cmd_cram_demuxlet.cpp:390-734 computes; `write_single` / `write_doublet` are the writers of :465-527 and :713-875.
Everything numerical happens in libdmx.so (HIP); this module only owns buffers and marshals pointers."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .capi import check


def device_warm_up(device: int = 0, n_gpus: int = 1) -> None:
    """dmx_device_warm_up: HIP context(s) and this library's code object, ahead of the first job of the process."""
    check(capi.load().dmx_device_warm_up(int(device), int(n_gpus)))


def phred_tables():
    mat, err = np.zeros(256), np.zeros(256)
    check(capi.load().dmx_phred_tables(mat.ctypes.data, err.ctypes.data))
    return mat, err


def geno_from_gt(alleles, gt_error: float) -> np.ndarray:
    a = np.ascontiguousarray(alleles, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((a.shape[0], 3), dtype=np.float32)
    check(capi.load().dmx_geno_from_gt(a.ctypes.data, a.shape[0], gt_error, out.ctypes.data))
    return out


def geno_from_pl(pl) -> np.ndarray:
    a = np.ascontiguousarray(pl, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), dtype=np.float32)
    check(capi.load().dmx_geno_from_pl(a.ctypes.data, a.shape[0], out.ctypes.data))
    return out


def geno_from_gp(gp, gt_error: float) -> np.ndarray:
    a = np.ascontiguousarray(gp, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), dtype=np.float32)
    check(capi.load().dmx_geno_from_gp(a.ctypes.data, a.shape[0], gt_error, out.ctypes.data))
    return out


@dataclass
class HostPileup:
    """numpy view of a dmx_pileup in host memory (arrays are kept alive by this object)."""
    n_cells: int
    n_snps: int
    cell_pair_off: np.ndarray
    cell_read_off: np.ndarray
    pair_snp: Optional[np.ndarray]
    pair_nrd: np.ndarray
    reads: np.ndarray
    rd_totl: np.ndarray
    rd_pass: np.ndarray
    rd_uniq: np.ndarray

    def as_struct(self) -> capi.Pileup:
        for name, dt in (("cell_pair_off", np.int64), ("cell_read_off", np.int64), ("reads", np.uint8),
                         ("rd_totl", np.int32), ("rd_pass", np.int32), ("rd_uniq", np.int32)):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))
        if self.pair_snp is not None:
            self.pair_snp = np.ascontiguousarray(self.pair_snp, dtype=np.int32)
        if self.pair_nrd.dtype not in (np.uint8, np.uint16, np.uint32):
            self.pair_nrd = self.pair_nrd.astype(np.uint32)
        self.pair_nrd = np.ascontiguousarray(self.pair_nrd)
        return capi.Pileup(self.n_cells, self.n_snps, len(self.pair_nrd), len(self.reads),
                           self.cell_pair_off.ctypes.data, self.cell_read_off.ctypes.data,
                           self.pair_snp.ctypes.data if self.pair_snp is not None else None,
                           self.pair_nrd.ctypes.data, self.pair_nrd.dtype.itemsize, capi.DMX_MEM_HOST,
                           self.reads.ctypes.data, self.rd_totl.ctypes.data, self.rd_pass.ctypes.data,
                           self.rd_uniq.ctypes.data)

    @property
    def n_snp_per_cell(self) -> np.ndarray:
        return np.diff(self.cell_pair_off).astype(np.int32)


class Store:
    """sc_dropseq_lib_t (sc_drop_seq.h:34-58): the UMI-deduplicated pileup, built read by read."""

    def __init__(self):
        self._L = capi.load()
        self._h = self._L.dmx_store_new()
        if not self._h:
            check(-6)

    def close(self):
        if self._h:
            self._L.dmx_store_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._h

    def add_snp(self) -> int: return check(self._L.dmx_store_add_snp(self._h))
    def add_cell(self, barcode: str) -> int: return check(self._L.dmx_store_add_cell(self._h, barcode.encode()))
    def count_read(self, cell: int) -> None: check(self._L.dmx_store_count_read(self._h, cell))

    def add_read(self, snp: int, cell: int, umi: str, allele: int, bq: int) -> bool:
        return bool(check(self._L.dmx_store_add_read(self._h, snp, cell, umi.encode(), allele, bq)))

    def add_batch(self, snp, cell, umis: Sequence[str], allele, bq, n_threads: int = 0) -> np.ndarray:
        """len(umis) add_read calls in order, inserted on several host threads (dmx_store_add_batch); returns their return values."""
        n = len(umis)
        snp = np.ascontiguousarray(snp, dtype=np.int32); cell = np.ascontiguousarray(cell, dtype=np.int32)
        allele = np.ascontiguousarray(allele, dtype=np.uint8); bq = np.ascontiguousarray(bq, dtype=np.uint8)
        enc = [u.encode() for u in umis]
        lens = np.array([len(b) for b in enc], dtype=np.uint32)
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64) if n else np.zeros(0, np.uint64)
        pool = b"".join(enc) + b"\0"
        new = np.zeros(n, dtype=np.uint8)
        check(self._L.dmx_store_add_batch(self._h, n, snp.ctypes.data, cell.ctypes.data, pool, offs.ctypes.data, lens.ctypes.data,
                                          allele.ctypes.data, bq.ctypes.data, new.ctypes.data, n_threads))
        return new

    @property
    def n_cells(self) -> int: return self._L.dmx_store_n_cells(self._h)
    @property
    def n_snps(self) -> int: return self._L.dmx_store_n_snps(self._h)

    def barcodes(self) -> List[str]:
        return [self._L.dmx_store_barcode(self._h, i).decode() for i in range(self.n_cells)]

    def freeze(self) -> HostPileup:
        pl = capi.Pileup()
        check(self._L.dmx_store_freeze(self._h, C.byref(pl)))

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()

        nrd_dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[pl.nrd_width]
        B = pl.n_cells
        return HostPileup(B, pl.n_snps, arr(pl.cell_pair_off, B + 1, np.int64), arr(pl.cell_read_off, B + 1, np.int64),
                          arr(pl.pair_snp, pl.n_pairs, np.int32), arr(pl.pair_nrd, pl.n_pairs, nrd_dt),
                          arr(pl.reads, pl.n_reads, np.uint8), arr(pl.rd_totl, B, np.int32), arr(pl.rd_pass, B, np.int32),
                          arr(pl.rd_uniq, B, np.int32))


class Engine:
    """The likelihood engine on one MI355X."""

    def __init__(self, n_samples: int, alphas: Sequence[float] = (0.0, 0.5), doublet_prior: float = 0.5, device: int = 0,
                 mode: int = capi.DMX_MODE_STRICT, flags: int = 0):
        self._L = capi.load()
        self.V = int(n_samples)
        self.alphas = np.ascontiguousarray(alphas, dtype=np.float64)
        self.A = len(self.alphas)
        self.prior = float(doublet_prior)
        cfg = capi.EngineConfig(self.V, self.A, self.alphas.ctypes.data, self.prior, device, mode, flags)
        h = C.c_void_p()
        check(self._L.dmx_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._keep = []
        self.B = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.dmx_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_stream(self, hip_stream: int) -> None:
        check(self._L.dmx_engine_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_phred_tables(self, mat: np.ndarray, err: np.ndarray) -> None:
        """dmx_engine_set_phred_tables: the engine's own match / error tables (256 doubles each) instead of PhredHelper's."""
        m = np.ascontiguousarray(mat, dtype=np.float64); e = np.ascontiguousarray(err, dtype=np.float64)
        assert m.shape == (256,) and e.shape == (256,)
        check(self._L.dmx_engine_set_phred_tables(self._h, m.ctypes.data, e.ctypes.data))

    def set_genotypes(self, g: np.ndarray) -> None:
        g = np.ascontiguousarray(g, dtype=np.float32)
        if g.ndim != 3 or g.shape[1] != self.V or g.shape[2] != 3:
            raise ValueError(f"g must be [S][{self.V}][3]")
        check(self._L.dmx_engine_set_genotypes(self._h, g.ctypes.data, g.shape[0], capi.DMX_MEM_HOST))

    def set_genotypes_device(self, ptr: int, n_snps: int) -> None:
        check(self._L.dmx_engine_set_genotypes(self._h, C.c_void_p(ptr), n_snps, capi.DMX_MEM_DEVICE))

    def set_pileup(self, pl: HostPileup) -> None:
        st = pl.as_struct()
        self._keep = [pl]
        check(self._L.dmx_engine_set_pileup(self._h, C.byref(st)))
        self.B = pl.n_cells

    def set_pileup_struct(self, st: capi.Pileup, keep=None) -> None:
        self._keep = [keep]
        check(self._L.dmx_engine_set_pileup(self._h, C.byref(st)))
        self.B = st.n_cells

    def run_singlet(self) -> None: check(self._L.dmx_engine_run_singlet(self._h))
    def run_doublet(self) -> None: check(self._L.dmx_engine_run_doublet(self._h))
    def run(self) -> None: check(self._L.dmx_engine_run(self._h))   # K1 beside K2 -> K3 -> K3b (dmx_engine_run)
    def sync(self) -> None: check(self._L.dmx_engine_sync(self._h))

    def get_singlet(self):
        llks = np.zeros((self.B, self.V))
        llk0s = np.zeros(self.B)
        check(self._L.dmx_engine_get_singlet(self._h, llks.ctypes.data, llk0s.ctypes.data))
        return llks, llk0s

    def get_doublet(self, want_grid: bool = True):
        grid = np.zeros((self.B, self.V, self.V, self.A)) if want_grid else None
        l00 = np.zeros((self.B, self.A))
        summ = np.zeros(self.B, dtype=capi.SUMMARY_DTYPE)
        check(self._L.dmx_engine_get_doublet(self._h, grid.ctypes.data if want_grid else None, l00.ctypes.data,
                                             summ.ctypes.data))
        return grid, l00, summ

    def get_sing(self) -> np.ndarray:
        sing = np.zeros((self.B, self.V))
        check(self._L.dmx_engine_get_sing(self._h, sing.ctypes.data))
        return sing

    def get_cell_grids(self, cells) -> np.ndarray:
        """llksAB[V][V][A] of the given cells (dmx_engine_get_cell_grids): the grids of the barcodes K3 flagged as near-ties are what a
        records-only consumer (write_doublet_summary, the multi-GPU gather) needs besides the records."""
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        out = np.zeros((len(cells), self.V, self.V, self.A), dtype=np.float64)
        check(self._L.dmx_engine_get_cell_grids(self._h, cells.ctypes.data, len(cells), out.ctypes.data))
        return out

    def format_pair(self, cells, barcodes: Sequence[str], sample_ids: Sequence[str], host_rows=None, ovr=None):
        """dmx_engine_format_pair: the `.pair` rows (cmd_cram_demuxlet.cpp:772-797) of the barcodes `cells` (ids of the staged pileup, output order) formatted on
        the device.  Returns (text bytes, cell_off[n+1], cell_flag[n], patches as a list of (offset, value, out_cell, singlet), format_ms): the POSTPRB fields the
        device leaves to the host's libm are EMPTY in the text and listed in `patches`."""
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        n = len(cells)
        bc, k1 = _cstrs(barcodes)
        sm, k2 = _cstrs(sample_ids)
        hr = np.ascontiguousarray(host_rows, dtype=np.uint8) if host_rows is not None else None
        ov = None
        if ovr is not None:
            ov = (capi.PairOverride * max(n, 1))()
            for i, o in enumerate(ovr):
                ov[i] = capi.PairOverride(*o) if o is not None else capi.PairOverride(-1, -1, -1, 0, 0.0, 0.0)
        rq = capi.PairRequest(n, cells.ctypes.data if n else None, C.cast(bc, C.c_void_p), C.cast(sm, C.c_void_p),
                              hr.ctypes.data if hr is not None else None, C.cast(ov, C.c_void_p) if ov is not None else None)
        h = C.c_void_p()
        check(self._L.dmx_engine_format_pair(self._h, C.byref(rq), C.byref(h)))
        try:
            info = capi.PairTextInfo()
            check(self._L.dmx_pair_text_get_info(h, C.byref(info)))
            buf = C.create_string_buffer(max(int(info.n_bytes), 1))
            check(self._L.dmx_pair_text_read(h, 0, info.n_bytes, buf))
            off = np.array([info.cell_off[i] for i in range(n + 1)], dtype=np.int64)
            flag = np.array([info.cell_flag[i] for i in range(n)], dtype=np.uint8)
            patches = [(int(info.patches[i].offset), float(info.patches[i].value), int(info.patches[i].out_cell), int(info.patches[i].singlet))
                       for i in range(info.n_patches)]
            return buf.raw[:info.n_bytes], off, flag, patches, float(info.format_ms)
        finally:
            self._L.dmx_pair_text_free(h)

    def device_view(self) -> capi.DeviceView:
        v = capi.DeviceView()
        check(self._L.dmx_engine_device_view(self._h, C.byref(v)))
        return v

    def kernel_times(self) -> capi.KernelTimes:
        t = capi.KernelTimes()
        check(self._L.dmx_engine_last_kernel_times(self._h, C.byref(t)))
        return t

    def mean_kernel_times(self, reset: bool = False) -> capi.KernelTimeMeans:
        """Mean HIP-event time of K1, K2, K3 and K3b over the launches since the last reset (at most the last 16)."""
        t = capi.KernelTimeMeans()
        check(self._L.dmx_engine_mean_kernel_times(self._h, int(reset), C.byref(t)))
        return t

    def reset_kernel_times(self) -> None:
        check(self._L.dmx_engine_mean_kernel_times(self._h, 1, None))

    def kernel_names(self) -> dict:
        """Which kernels the last run launched (dmx_engine_kernel_names): {'singlet', 'doublet', 'certify': rocprofv3-style names, 'k1_placement'}."""
        n = capi.KernelNames()
        check(self._L.dmx_engine_kernel_names(self._h, C.byref(n)))
        return {"singlet": n.singlet.decode(), "doublet": n.doublet.decode(), "certify": n.certify.decode(), "k1_placement": int(n.k1_placement)}

    def algorithmic_bytes(self) -> capi.KernelBytes:
        b = capi.KernelBytes()
        check(self._L.dmx_engine_algorithmic_bytes(self._h, C.byref(b)))
        return b


def _cstrs(strs: Sequence[str]):
    keep = [s.encode() for s in strs]
    arr = (C.c_char_p * max(1, len(keep)))(*keep) if keep else (C.c_char_p * 1)()
    return arr, keep


@dataclass
class FinalArgs:
    barcodes: Sequence[str]
    sample_ids: Sequence[str]
    alphas: Sequence[float]
    doublet_prior: float
    rd_totl: np.ndarray
    rd_pass: np.ndarray
    rd_uniq: np.ndarray
    n_snp: np.ndarray
    min_total: int = 0
    min_uniq: int = 0
    min_snp: int = 0
    write_pair: bool = False


def _final_struct(fa: FinalArgs, llks=None, llk0s=None, grid=None, l00=None, tie_pileup: Optional[HostPileup] = None,
                  tie_g: Optional[np.ndarray] = None, cell_grids=None):
    keep = []
    alphas = np.ascontiguousarray(fa.alphas, dtype=np.float64)
    bc, k1 = _cstrs(fa.barcodes)
    sm, k2 = _cstrs(fa.sample_ids)
    arrs = [np.ascontiguousarray(x, dtype=np.int32) for x in (fa.rd_totl, fa.rd_pass, fa.rd_uniq, fa.n_snp)]
    keep += [alphas, bc, k1, sm, k2, arrs]

    def p(x):
        if x is None:
            return None
        x = np.ascontiguousarray(x, dtype=np.float64)
        keep.append(x)
        return x.ctypes.data

    tp = None
    if tie_pileup is not None:
        st = tie_pileup.as_struct()
        keep += [st, tie_pileup]
        tp = C.addressof(st)
        tie_g = np.ascontiguousarray(tie_g, dtype=np.float32)
        keep.append(tie_g)
    fin = capi.FinalInput(len(fa.barcodes), len(fa.sample_ids), len(alphas), alphas.ctypes.data, fa.doublet_prior,
                          fa.min_total, fa.min_uniq, fa.min_snp, int(fa.write_pair), C.cast(bc, C.c_void_p),
                          C.cast(sm, C.c_void_p), arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                          arrs[3].ctypes.data, p(llks), p(llk0s), p(grid), p(l00), tp,
                          tie_g.ctypes.data if tie_pileup is not None else None, 0.0)
    fin._cell_grid = None
    if cell_grids:                                   # {cell id: llksAB[V][V][A]} of the near-tie-flagged barcodes
        ptrs = (C.c_void_p * len(fa.barcodes))()     # (an argument of dmx_write_doublet_summary_grids since ABI 7, not a member of the struct)
        for c, gr in cell_grids.items():
            gr = np.ascontiguousarray(gr, dtype=np.float64)
            assert gr.size == len(fa.sample_ids) ** 2 * len(alphas)
            keep.append(gr)
            ptrs[int(c)] = gr.ctypes.data
        keep.append(ptrs)
        fin._cell_grid = C.cast(ptrs, C.c_void_p)
    return fin, keep


def write_single(fa: FinalArgs, llks, llk0s, path: str) -> None:
    fin, keep = _final_struct(fa, llks=llks, llk0s=llk0s)
    check(capi.load().dmx_write_single(C.byref(fin), path.encode()))


def write_doublet(fa: FinalArgs, grid, l00, out_prefix: str, tie_pileup: Optional[HostPileup] = None,
                  tie_g: Optional[np.ndarray] = None) -> None:
    fin, keep = _final_struct(fa, grid=grid, l00=l00, tie_pileup=tie_pileup, tie_g=tie_g)
    check(capi.load().dmx_write_doublet(C.byref(fin), out_prefix.encode()))


def near_tie_cells(summary: np.ndarray) -> np.ndarray:
    """ids of the covered cells whose K3 record carries DMX_CELL_NEAR_DOUBLET / _NEAR_SINGLET: a decision of theirs sits within 1e-7 of
    an alternative other than the alpha = 0.5 mirror, and the writers decide it from the cell's grid (Engine.get_cell_grids)."""
    return np.flatnonzero(((summary["flags"] & (capi.DMX_CELL_NEAR_DOUBLET | capi.DMX_CELL_NEAR_SINGLET)) != 0) & (summary["n_pairs"] > 0)).astype(np.int32)


def write_doublet_summary(fa: FinalArgs, sing, l00, summary, out_prefix: str, tie_pileup: Optional[HostPileup] = None,
                          tie_g: Optional[np.ndarray] = None, cell_grids=None) -> None:
    """.sing2/.best from the per-cell records (K3 summaries) instead of the grid — what a multi-GPU run gathers.  `cell_grids`
    {cell id: llksAB[V][V][A]} carries the grids of the near-tie-flagged barcodes (near_tie_cells); without them and with the tie
    pileup such a barcode's grid is re-evaluated on the host."""
    fin, keep = _final_struct(fa, l00=l00, tie_pileup=tie_pileup, tie_g=tie_g, cell_grids=cell_grids)
    sing = np.ascontiguousarray(sing, dtype=np.float64)
    summary = np.ascontiguousarray(summary, dtype=capi.SUMMARY_DTYPE)
    if fin._cell_grid is not None:
        check(capi.load().dmx_write_doublet_summary_grids(C.byref(fin), sing.ctypes.data, summary.ctypes.data, fin._cell_grid, out_prefix.encode()))
    else:
        check(capi.load().dmx_write_doublet_summary(C.byref(fin), sing.ctypes.data, summary.ctypes.data, out_prefix.encode()))


def resolve_tie_order(summary: np.ndarray) -> int:
    """dmx_resolve_tie_order: DMX_CELL_ORDER_RESOLVABLE records -> certified ones, in place (the host libm's log() decides).
    Returns how many stayed unresolved."""
    assert summary.dtype == capi.SUMMARY_DTYPE and summary.flags["C_CONTIGUOUS"]
    return check(capi.load().dmx_resolve_tie_order(summary.ctypes.data, len(summary)))


def demuxlet_run(store, g: np.ndarray, sample_ids: Sequence[str], alphas: Sequence[float], out_prefix: str,
                 doublet_prior: float = 0.5, min_total: int = 0, min_uniq: int = 0, min_snp: int = 0,
                 write_pair: bool = False, device: int = 0, arbiter: bool = True, n_gpus: int = 1, mode: int = capi.DMX_MODE_STRICT,
                 barcodes: Optional[Sequence[str]] = None, timing: bool = False):
    """cmd_cram_demuxlet.cpp:390-881 in one call (dmx_demuxlet_run).  `store` is a Store, or a frozen pileup — a HostPileup or a
    capi.Pileup struct (device-resident arrays: memory = DMX_MEM_DEVICE, one GPU) — together with `barcodes` (dmx_job.pileup).
    With timing=True returns the stage seconds (dmx_job_timing) as a dict."""
    g = np.ascontiguousarray(g, dtype=np.float32)
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    sm, keep = _cstrs(sample_ids)
    tm = capi.JobTiming()
    if isinstance(store, (HostPileup, capi.Pileup)):
        # a frozen pileup: host arrays (HostPileup) or a dmx_pileup struct as it is — DMX_MEM_DEVICE: the five arrays in HBM, the
        # rd_* counters host memory (the caller keeps whatever the pointers refer to alive)
        st = store.as_struct() if isinstance(store, HostPileup) else store
        bc, keep_b = _cstrs(barcodes)
        job = capi.Job(None, g.ctypes.data, g.shape[1], C.cast(sm, C.c_void_p), len(al), al.ctypes.data, doublet_prior,
                       min_total, min_uniq, min_snp, int(write_pair), out_prefix.encode(), device, int(arbiter), n_gpus, mode,
                       C.addressof(st), C.cast(bc, C.c_void_p), C.addressof(tm) if timing else None)
    else:
        job = capi.Job(store.handle, g.ctypes.data, g.shape[1], C.cast(sm, C.c_void_p), len(al), al.ctypes.data, doublet_prior,
                       min_total, min_uniq, min_snp, int(write_pair), out_prefix.encode(), device, int(arbiter), n_gpus, mode,
                       None, None, C.addressof(tm) if timing else None)
    check(capi.load().dmx_demuxlet_run(C.byref(job)))
    if timing:
        return {name: getattr(tm, name) for name, _ in capi.JobTiming._fields_}
