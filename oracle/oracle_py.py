"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/dmx_oracle.c) plus helpers to drive the
reference slice harness (oracle/_ref/ref_slice_harness, dev container only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module. Nothing under
demuxlet_amd/ may import it (tests/test_abi.py::test_product_never_touches_the_oracle enforces that)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "_build" / "liboracle.so"
REF_HARNESS = HERE / "_ref" / "ref_slice_harness"
REF_UNITS = HERE / "_ref" / "libref_units.so"


def build(force: bool = False) -> None:
    """Compile the oracle (gcc) and, when /root/reference is present, the reference harness."""
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < (HERE / "dmx_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-s", "-C", str(HERE), "oracle"])
    if Path("/root/reference").is_dir() and (force or not REF_HARNESS.exists() or not REF_UNITS.exists()):
        subprocess.check_call(["make", "-s", "-C", str(HERE), "ref"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        L.orc_phred_tables.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_phred_prob.restype = C.c_double
        L.orc_phred_prob.argtypes = [C.c_uint32]
        L.orc_geno_from_gt.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p]
        L.orc_geno_from_pl.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_geno_from_gp.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p]
        L.orc_store_new.restype = C.c_void_p
        L.orc_store_free.argtypes = [C.c_void_p]
        L.orc_store_add_cell.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_store_add_cell.restype = C.c_int32
        L.orc_store_count_read.argtypes = [C.c_void_p, C.c_int32]
        L.orc_store_add_read.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.c_int32]
        L.orc_store_add_read.restype = C.c_int32
        L.orc_store_freeze.argtypes = [C.c_void_p]
        for name, rt in [("ncells", C.c_int32), ("npairs", C.c_int64), ("nwords", C.c_int64),
                         ("cell_off", C.c_void_p), ("pair_snp", C.c_void_p), ("pair_off", C.c_void_p),
                         ("words", C.c_void_p), ("totl", C.c_void_p), ("pass", C.c_void_p), ("uniq", C.c_void_p)]:
            f = getattr(L, "orc_store_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = rt
        L.orc_store_barcode.argtypes = [C.c_void_p, C.c_int32]
        L.orc_store_barcode.restype = C.c_char_p
        L.orc_store_sorted_order.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_run.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.orc_run.restype = C.c_int
        L.orc_wall_seconds.restype = C.c_double
        _lib = L
    return _lib


class _Problem(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("n_snps", C.c_int32), ("n_samples", C.c_int32), ("n_alpha", C.c_int32),
                ("alpha", C.c_void_p), ("doublet_prior", C.c_double),
                ("min_total", C.c_int32), ("min_uniq", C.c_int32), ("min_snp", C.c_int32), ("write_pair", C.c_int32),
                ("cell_off", C.c_void_p), ("pair_snp", C.c_void_p), ("pair_off", C.c_void_p), ("words", C.c_void_p),
                ("rd_totl", C.c_void_p), ("rd_pass", C.c_void_p), ("rd_uniq", C.c_void_p),
                ("g", C.c_void_p), ("sample_ids", C.c_void_p), ("barcodes", C.c_void_p), ("singlet_only", C.c_int32)]


class _Raw(C.Structure):
    _fields_ = [("llks", C.c_void_p), ("llk0s", C.c_void_p), ("llksAB", C.c_void_p), ("llks00", C.c_void_p),
                ("processed", C.c_void_p)]


# ----------------------------------------------------------------------------------------------------------------
# Neutral problem description shared by the oracle, the reference harness and the product tests
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class Events:
    """One entry per (read, overlapping SNP) in BAM order; snp = -1 for a read that overlaps no SNP."""
    barcode: List[str]
    snp: np.ndarray       # int32
    umi: List[str]
    allele: np.ndarray    # uint8 in {0,1,2}
    bq: np.ndarray        # uint8
    newread: np.ndarray   # uint8: 1 when this entry starts a new read (counts in RD.TOTL)


@dataclass
class Params:
    alphas: Sequence[float] = (0.0, 0.5)
    doublet_prior: float = 0.5
    min_total: int = 0
    min_uniq: int = 0
    min_snp: int = 0
    write_pair: bool = False


@dataclass
class Problem:
    sample_ids: List[str]
    g: np.ndarray         # float32 [S][V][3]
    events: Events
    params: Params = field(default_factory=Params)

    @property
    def n_snps(self) -> int: return int(self.g.shape[0])

    @property
    def n_samples(self) -> int: return int(self.g.shape[1])


@dataclass
class Csr:
    barcodes: List[str]          # by cell id (first-appearance order)
    cell_off: np.ndarray         # int64 [B+1]
    pair_snp: np.ndarray         # int32 [P]
    pair_off: np.ndarray         # int64 [P+1]
    words: np.ndarray            # uint32 [R]  (allele<<24)|(bq<<16)|count
    rd_totl: np.ndarray
    rd_pass: np.ndarray
    rd_uniq: np.ndarray

    @property
    def n_cells(self) -> int: return len(self.barcodes)

    def sorted_order(self) -> np.ndarray:
        enc = [b.encode() for b in self.barcodes]
        return np.array(sorted(range(len(enc)), key=lambda i: enc[i]), dtype=np.int32)


def store_from_events(ev: Events) -> Csr:
    """Row a1 through the oracle's store (restatement of sc_drop_seq.cpp)."""
    L = lib()
    st = L.orc_store_new()
    try:
        snp = np.asarray(ev.snp, dtype=np.int64)
        for e in range(len(ev.barcode)):
            c = L.orc_store_add_cell(st, ev.barcode[e].encode())
            if ev.newread[e]:
                L.orc_store_count_read(st, c)
            if snp[e] >= 0:
                L.orc_store_add_read(st, int(snp[e]), c, ev.umi[e].encode(), int(ev.allele[e]), int(ev.bq[e]))
        L.orc_store_freeze(st)
        B = L.orc_store_ncells(st)
        Pn = L.orc_store_npairs(st)
        R = L.orc_store_nwords(st)

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()

        return Csr(
            barcodes=[L.orc_store_barcode(st, i).decode() for i in range(B)],
            cell_off=arr(L.orc_store_cell_off(st), B + 1, np.int64),
            pair_snp=arr(L.orc_store_pair_snp(st), Pn, np.int32),
            pair_off=arr(L.orc_store_pair_off(st), Pn + 1, np.int64),
            words=arr(L.orc_store_words(st), R, np.uint32),
            rd_totl=arr(L.orc_store_totl(st), B, np.int32),
            rd_pass=arr(L.orc_store_pass(st), B, np.int32),
            rd_uniq=arr(L.orc_store_uniq(st), B, np.int32),
        )
    finally:
        L.orc_store_free(st)


@dataclass
class RawOut:
    llks: np.ndarray
    llk0s: np.ndarray
    llksAB: Optional[np.ndarray]
    llks00: Optional[np.ndarray]
    processed: Optional[np.ndarray]


def _cstr_array(strs: Sequence[str]):
    arr = (C.c_char_p * max(1, len(strs)))()
    keep = [s.encode() for s in strs]
    for i, s in enumerate(keep):
        arr[i] = s
    return arr, keep


class CsrPlan:
    """Everything orc_run needs, marshalled once; execute() is then the bare C call (ctypes drops the GIL around it, so
    several plans can run on several host threads at the same time)."""

    def __init__(self, csr: Csr, sample_ids: Sequence[str], g: np.ndarray, params: Params, out_prefix: Optional[str] = None,
                 singlet_only: bool = False, want_grid: bool = True):
        self.L = lib()
        g = np.ascontiguousarray(g, dtype=np.float32)
        S, V, _ = g.shape
        B = csr.n_cells
        alphas = np.ascontiguousarray(params.alphas, dtype=np.float64)
        A = len(alphas)
        sm, _k1 = _cstr_array(sample_ids)
        bc, _k2 = _cstr_array(csr.barcodes)
        cell_off = np.ascontiguousarray(csr.cell_off, dtype=np.int64)
        pair_snp = np.ascontiguousarray(csr.pair_snp, dtype=np.int32)
        pair_off = np.ascontiguousarray(csr.pair_off, dtype=np.int64)
        words = np.ascontiguousarray(csr.words, dtype=np.uint32)
        totl = np.ascontiguousarray(csr.rd_totl, dtype=np.int32)
        pas = np.ascontiguousarray(csr.rd_pass, dtype=np.int32)
        uniq = np.ascontiguousarray(csr.rd_uniq, dtype=np.int32)
        if len(pair_snp) and (pair_snp.min() < 0 or pair_snp.max() >= S):
            raise ValueError("pair_snp out of range")
        self.P = _Problem(B, S, V, A, alphas.ctypes.data, params.doublet_prior, params.min_total, params.min_uniq,
                          params.min_snp, int(params.write_pair), cell_off.ctypes.data,
                          pair_snp.ctypes.data if len(pair_snp) else None, pair_off.ctypes.data,
                          words.ctypes.data if len(words) else None, totl.ctypes.data if B else None,
                          pas.ctypes.data if B else None, uniq.ctypes.data if B else None, g.ctypes.data,
                          C.cast(sm, C.c_void_p), C.cast(bc, C.c_void_p), int(singlet_only))
        llks = np.zeros((B, V), dtype=np.float64)
        llk0s = np.zeros(B, dtype=np.float64)
        grid = np.zeros((B, V, V, A), dtype=np.float64) if (want_grid and not singlet_only) else None
        l00 = np.zeros((B, A), dtype=np.float64) if not singlet_only else None
        proc = np.zeros(B, dtype=np.uint8) if not singlet_only else None
        self.R = _Raw(llks.ctypes.data, llk0s.ctypes.data, grid.ctypes.data if grid is not None else None,
                      l00.ctypes.data if l00 is not None else None, proc.ctypes.data if proc is not None else None)
        self.out = RawOut(llks, llk0s, grid, l00, proc)
        self.prefix = out_prefix.encode() if out_prefix else None
        self.n_pairs = int(cell_off[-1]) if B else 0
        self._keep = (g, alphas, sm, _k1, bc, _k2, cell_off, pair_snp, pair_off, words, totl, pas, uniq)

    def execute(self) -> RawOut:
        rc = self.L.orc_run(C.byref(self.P), C.byref(self.R), self.prefix)
        if rc != 0:
            raise RuntimeError(f"orc_run failed rc={rc}")
        return self.out


def run_csr(csr: Csr, sample_ids: Sequence[str], g: np.ndarray, params: Params, out_prefix: Optional[str] = None,
            singlet_only: bool = False, want_grid: bool = True) -> RawOut:
    """Rows a4..a14 through the oracle."""
    return CsrPlan(csr, sample_ids, g, params, out_prefix, singlet_only, want_grid).execute()


def run_problem(pb: Problem, out_prefix: Optional[str] = None, singlet_only: bool = False):
    csr = store_from_events(pb.events)
    return csr, run_csr(csr, pb.sample_ids, pb.g, pb.params, out_prefix, singlet_only)


def geno_from_gt(alleles: np.ndarray, gt_error: float) -> np.ndarray:
    a = np.ascontiguousarray(alleles, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((a.shape[0], 3), dtype=np.float32)
    lib().orc_geno_from_gt(a.ctypes.data, a.shape[0], gt_error, out.ctypes.data)
    return out


def geno_from_pl(pl: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(pl, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), dtype=np.float32)
    lib().orc_geno_from_pl(a.ctypes.data, a.shape[0], out.ctypes.data)
    return out


def geno_from_gp(gp: np.ndarray, gt_error: float) -> np.ndarray:
    a = np.ascontiguousarray(gp, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), dtype=np.float32)
    lib().orc_geno_from_gp(a.ctypes.data, a.shape[0], gt_error, out.ctypes.data)
    return out


def phred_tables():
    mat = np.zeros(256)
    err = np.zeros(256)
    lib().orc_phred_tables(mat.ctypes.data, err.ctypes.data)
    return mat, err


# ----------------------------------------------------------------------------------------------------------------
# Reference slice harness (dev container only)
# ----------------------------------------------------------------------------------------------------------------
def have_ref() -> bool:
    return REF_HARNESS.exists()


def write_spec(pb: Problem, path: str) -> None:
    ev = pb.events
    S, V, _ = pb.g.shape
    seen = []
    seen_set = set()
    for b in ev.barcode:
        if b not in seen_set:
            seen_set.add(b)
            seen.append(b)
    with open(path, "w") as f:
        f.write("DMXSPEC1\n")
        f.write(f"{V} {S} {len(seen)} {len(pb.params.alphas)} {len(ev.barcode)}\n")
        f.write(" ".join(float(a).hex() for a in pb.params.alphas) + "\n")
        p = pb.params
        f.write(f"{float(p.doublet_prior).hex()} {p.min_total} {p.min_uniq} {p.min_snp} {int(p.write_pair)}\n")
        for s in pb.sample_ids:
            f.write(f"SM {s}\n")
        g64 = pb.g.astype(np.float64).reshape(S, V * 3)
        for s in range(S):
            f.write("G " + " ".join(float(x).hex() for x in g64[s]) + "\n")
        for b in seen:
            f.write(f"BC {b}\n")
        for e in range(len(ev.barcode)):
            f.write(f"R {ev.barcode[e]} {int(ev.snp[e])} {ev.umi[e]} {int(ev.allele[e])} {int(ev.bq[e])} {int(ev.newread[e])}\n")


@dataclass
class RefOut:
    files: dict            # suffix -> bytes for single/sing2/best/pair
    barcodes: List[str]    # by id
    counters: np.ndarray   # [B][4] totl,pass,uniq,nsnp
    llks: np.ndarray
    llk0s: np.ndarray
    cell_ids: np.ndarray   # ids of cells that went through the doublet loop, in processing order
    llksAB: np.ndarray     # [len(cell_ids)][V][V][A]
    llks00: np.ndarray     # [len(cell_ids)][A]


def run_ref(pb: Problem, workdir: str, raw: bool = True) -> RefOut:
    if not have_ref():
        raise RuntimeError("reference harness not built (needs /root/reference)")
    os.makedirs(workdir, exist_ok=True)
    spec = os.path.join(workdir, "spec.txt")
    out = os.path.join(workdir, "ref")
    write_spec(pb, spec)
    args = [str(REF_HARNESS), spec, out] + ([] if raw else ["--no-raw"])
    subprocess.run(args, check=True, stderr=subprocess.DEVNULL)
    files = {}
    for suf in ("single", "sing2", "best", "pair"):
        pth = out + "." + suf
        if os.path.exists(pth):
            files[suf] = open(pth, "rb").read()
    V = pb.n_samples
    A = len(pb.params.alphas)
    bcs, cnt = [], []
    for line in open(out + ".raw.barcodes"):
        t = line.rstrip("\n").split("\t")
        bcs.append(t[0])
        cnt.append([int(x) for x in t[1:5]])
    B = len(bcs)
    ids = np.fromfile(out + ".raw.cellid", dtype=np.int32)
    return RefOut(files, bcs, np.array(cnt, dtype=np.int32).reshape(B, 4),
                  np.fromfile(out + ".raw.llks").reshape(B, V), np.fromfile(out + ".raw.llk0s"),
                  ids, np.fromfile(out + ".raw.llksAB").reshape(len(ids), V, V, A),
                  np.fromfile(out + ".raw.llks00").reshape(len(ids), A))
