"""ctypes binding of libdmx.so — a 1:1 mirror of include/dmx.h (no logic here beyond marshalling).

The library is loaded lazily from demuxlet_amd/libdmx.so (built in-tree by demuxlet_amd/build.py).  There is no Python
or CPU fallback for the engine: if the library or the GPU is missing, calls raise DmxError."""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import numpy as np

import os

LIB_PATH = Path(os.environ.get("DMX_LIB", Path(__file__).resolve().parent / "libdmx.so"))   # DMX_LIB: kernel experiments only

DMX_OK = 0
DMX_ERR_ARG, DMX_ERR_HIP, DMX_ERR_STATE, DMX_ERR_IO, DMX_ERR_NOGPU, DMX_ERR_NOMEM = -1, -2, -3, -4, -5, -6   # dmx_status (include/dmx.h)
DMX_MEM_HOST, DMX_MEM_DEVICE = 0, 1
DMX_MODE_STRICT = 0
DMX_MODE_FAST = 1
DMX_CELL_NEAR_DOUBLET, DMX_CELL_NEAR_SINGLET, DMX_CELL_ORDER_CERTIFIED, DMX_CELL_ORDER_RESOLVABLE, DMX_CELL_NEAR_RULE = 1, 2, 4, 8, 16
DMX_ENGINE_NO_CERTIFY = 1

# every symbol include/dmx.h declares (tests/test_abi.py checks the header against this list and the .so against both)
SYMBOLS = [
    "dmx_abi_version", "dmx_last_error", "dmx_device_warm_up", "dmx_phred_tables", "dmx_geno_from_gt", "dmx_geno_from_pl", "dmx_geno_from_gp",
    "dmx_store_new", "dmx_store_free", "dmx_store_add_snp", "dmx_store_add_cell", "dmx_store_count_read",
    "dmx_store_add_read", "dmx_store_n_cells", "dmx_store_n_snps", "dmx_store_barcode", "dmx_store_freeze",
    "dmx_engine_create", "dmx_engine_destroy", "dmx_engine_set_stream", "dmx_engine_set_phred_tables",
    "dmx_engine_set_genotypes", "dmx_engine_set_pileup", "dmx_engine_run_singlet", "dmx_engine_run_doublet", "dmx_engine_run",
    "dmx_engine_sync", "dmx_engine_get_singlet", "dmx_engine_get_doublet", "dmx_engine_device_view",
    "dmx_engine_last_kernel_times", "dmx_engine_algorithmic_bytes", "dmx_write_single", "dmx_write_doublet",
    "dmx_demuxlet_run", "dmx_debug_device_log", "dmx_debug_device_log2", "dmx_debug_device_div", "dmx_debug_log_rate", "dmx_engine_get_sing", "dmx_engine_get_cell_grids", "dmx_write_doublet_summary", "dmx_debug_log_dd",
    "dmx_resolve_tie_order", "dmx_engine_mean_kernel_times", "dmx_store_add_batch", "dmx_write_doublet_summary_grids", "dmx_engine_kernel_names", "dmx_debug_device_log2_lite", "dmx_debug_device_log2_lite32",
    "dmx_engine_format_pair", "dmx_pair_text_get_info", "dmx_pair_text_read", "dmx_pair_text_free",
]


class PairOverride(C.Structure):      # dmx_pair_override
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("n", C.c_int32), ("reserved", C.c_int32), ("llk_ab", C.c_double), ("llk_ba", C.c_double)]


class PairPatch(C.Structure):         # dmx_pair_patch
    _fields_ = [("offset", C.c_int64), ("value", C.c_double), ("out_cell", C.c_int32), ("singlet", C.c_int32)]


class PairRequest(C.Structure):       # dmx_pair_request
    _fields_ = [("n_out", C.c_int32), ("cells", C.c_void_p), ("barcodes", C.c_void_p), ("sample_ids", C.c_void_p), ("host_rows", C.c_void_p), ("ovr", C.c_void_p)]


class PairTextInfo(C.Structure):      # dmx_pair_text_info
    _fields_ = [("n_bytes", C.c_int64), ("n_out", C.c_int32), ("n_patches", C.c_int32), ("cell_off", C.POINTER(C.c_int64)), ("cell_flag", C.POINTER(C.c_uint8)),
                ("patches", C.POINTER(PairPatch)), ("format_ms", C.c_double)]


class DmxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libdmx error {code}: {msg}")
        self.code = code


class Pileup(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("n_snps", C.c_int32), ("n_pairs", C.c_int64), ("n_reads", C.c_int64),
                ("cell_pair_off", C.c_void_p), ("cell_read_off", C.c_void_p), ("pair_snp", C.c_void_p),
                ("pair_nrd", C.c_void_p), ("nrd_width", C.c_int32), ("memory", C.c_int32), ("reads", C.c_void_p),
                ("rd_totl", C.c_void_p), ("rd_pass", C.c_void_p), ("rd_uniq", C.c_void_p)]


class EngineConfig(C.Structure):
    _fields_ = [("n_samples", C.c_int32), ("n_alpha", C.c_int32), ("alpha", C.c_void_p), ("doublet_prior", C.c_double),
                ("device", C.c_int32), ("mode", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32 * 3)]


class CellSummary(C.Structure):
    _fields_ = [("max_llk", C.c_double), ("sum_single", C.c_double), ("sum_double", C.c_double),
                ("sing_llk1", C.c_double), ("sing_llk2", C.c_double),
                ("llk12", C.c_double), ("llk1", C.c_double), ("llk2", C.c_double), ("llk10", C.c_double), ("llk20", C.c_double),
                ("llk00_0", C.c_double), ("llk00_best", C.c_double),
                ("i_sing1", C.c_int32), ("i_sing2", C.c_int32), ("j_best", C.c_int32), ("k_best", C.c_int32),
                ("n_best", C.c_int32), ("n_pairs", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
                ("llk_ab", C.c_double), ("llk_ba", C.c_double), ("llk_ab_alt", C.c_double), ("llk_ba_alt", C.c_double),
                ("ev_x_ab", C.c_double), ("ev_t_ab", C.c_double), ("ev_x_ba", C.c_double), ("ev_t_ba", C.c_double)]


SUMMARY_DTYPE = np.dtype([(n, np.float64) for n in ("max_llk", "sum_single", "sum_double", "sing_llk1", "sing_llk2", "llk12",
                                                    "llk1", "llk2", "llk10", "llk20", "llk00_0", "llk00_best")] +
                         [(n, np.int32) for n in ("i_sing1", "i_sing2", "j_best", "k_best", "n_best", "n_pairs", "flags", "reserved")] +
                         [(n, np.float64) for n in ("llk_ab", "llk_ba", "llk_ab_alt", "llk_ba_alt", "ev_x_ab", "ev_t_ab", "ev_x_ba", "ev_t_ba")])
assert SUMMARY_DTYPE.itemsize == C.sizeof(CellSummary)


class DeviceView(C.Structure):
    _fields_ = [("llks", C.c_void_p), ("llk0s", C.c_void_p), ("llksAB", C.c_void_p), ("llks00", C.c_void_p),
                ("summary", C.c_void_p), ("gp0s", C.c_void_p), ("sing", C.c_void_p)]


class KernelTimes(C.Structure):
    _fields_ = [("gp0_ms", C.c_float), ("singlet_ms", C.c_float), ("doublet_ms", C.c_float), ("reduce_ms", C.c_float)]


class KernelTimeMeans(C.Structure):
    _fields_ = [("singlet_ms", C.c_double), ("doublet_ms", C.c_double), ("reduce_ms", C.c_double), ("certify_ms", C.c_double),
                ("n_singlet", C.c_int32), ("n_doublet", C.c_int32)]


class KernelNames(C.Structure):
    _fields_ = [("singlet", C.c_char * 96), ("doublet", C.c_char * 96), ("certify", C.c_char * 96), ("k1_placement", C.c_int32), ("reserved", C.c_int32 * 3)]


class KernelBytes(C.Structure):
    _fields_ = [("singlet_bytes", C.c_double), ("doublet_bytes", C.c_double), ("reduce_bytes", C.c_double)]


class FinalInput(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("n_samples", C.c_int32), ("n_alpha", C.c_int32), ("alpha", C.c_void_p),
                ("doublet_prior", C.c_double), ("min_total", C.c_int32), ("min_uniq", C.c_int32), ("min_snp", C.c_int32),
                ("write_pair", C.c_int32), ("barcodes", C.c_void_p), ("sample_ids", C.c_void_p),
                ("rd_totl", C.c_void_p), ("rd_pass", C.c_void_p), ("rd_uniq", C.c_void_p), ("n_snp", C.c_void_p),
                ("llks", C.c_void_p), ("llk0s", C.c_void_p), ("llksAB", C.c_void_p), ("llks00", C.c_void_p),
                ("tie_pileup", C.c_void_p), ("tie_g", C.c_void_p), ("tie_tol", C.c_double)]


class Job(C.Structure):
    _fields_ = [("store", C.c_void_p), ("g", C.c_void_p), ("n_samples", C.c_int32), ("sample_ids", C.c_void_p),
                ("n_alpha", C.c_int32), ("alpha", C.c_void_p), ("doublet_prior", C.c_double),
                ("min_total", C.c_int32), ("min_uniq", C.c_int32), ("min_snp", C.c_int32), ("write_pair", C.c_int32),
                ("out_prefix", C.c_char_p), ("device", C.c_int32), ("arbiter", C.c_int32), ("n_gpus", C.c_int32), ("mode", C.c_int32),
                ("pileup", C.c_void_p), ("barcodes", C.c_void_p), ("timing", C.c_void_p)]


class JobTiming(C.Structure):
    _fields_ = [("freeze_s", C.c_double), ("setup_s", C.c_double), ("stage_s", C.c_double), ("wait_s", C.c_double),
                ("write_s", C.c_double), ("total_s", C.c_double), ("kernel_ms", C.c_double),
                ("n_ranges", C.c_int32), ("n_engines", C.c_int32), ("n_cells_grid_fetched", C.c_int32), ("n_cells_single", C.c_int32)]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen libdmx.so; fails loudly when it has not been built (python -m demuxlet_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Load torch FIRST so that libdmx.so binds to
        # that copy: two HIP runtimes in one process leave whichever comes second without a device.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not LIB_PATH.exists():
        raise DmxError(-100, f"{LIB_PATH} is missing — build it with `python -m demuxlet_amd.build` (hipcc, gfx950)")
    L = C.CDLL(str(LIB_PATH))
    vp, i32, dbl = C.c_void_p, C.c_int32, C.c_double
    L.dmx_abi_version.restype = C.c_int
    L.dmx_last_error.restype = C.c_char_p
    sig = {
        "dmx_device_warm_up": [i32, i32], "dmx_phred_tables": [vp, vp], "dmx_geno_from_gt": [vp, i32, dbl, vp], "dmx_geno_from_pl": [vp, i32, vp],
        "dmx_geno_from_gp": [vp, i32, dbl, vp], "dmx_store_free": [vp], "dmx_store_add_snp": [vp],
        "dmx_store_add_cell": [vp, C.c_char_p], "dmx_store_count_read": [vp, i32],
        "dmx_store_add_read": [vp, i32, i32, C.c_char_p, i32, i32], "dmx_store_n_cells": [vp], "dmx_store_n_snps": [vp],
        "dmx_store_barcode": [vp, i32], "dmx_store_freeze": [vp, vp], "dmx_engine_create": [vp, vp],
        "dmx_engine_destroy": [vp], "dmx_engine_set_stream": [vp, vp], "dmx_engine_set_phred_tables": [vp, vp, vp],
        "dmx_engine_set_genotypes": [vp, vp, i32, i32], "dmx_engine_set_pileup": [vp, vp], "dmx_engine_run_singlet": [vp],
        "dmx_engine_run_doublet": [vp], "dmx_engine_run": [vp], "dmx_engine_sync": [vp], "dmx_engine_get_singlet": [vp, vp, vp],
        "dmx_engine_get_doublet": [vp, vp, vp, vp], "dmx_engine_device_view": [vp, vp],
        "dmx_engine_last_kernel_times": [vp, vp], "dmx_engine_algorithmic_bytes": [vp, vp],
        "dmx_write_single": [vp, C.c_char_p], "dmx_write_doublet": [vp, C.c_char_p], "dmx_demuxlet_run": [vp],
        "dmx_debug_device_log": [vp, vp, C.c_int64, i32], "dmx_debug_device_log2": [vp, vp, C.c_int64, i32], "dmx_debug_device_log2_lite": [vp, vp, C.c_int64, i32], "dmx_debug_device_log2_lite32": [vp, vp, C.c_int64, i32],
        "dmx_engine_get_sing": [vp, vp], "dmx_write_doublet_summary": [vp, vp, vp, C.c_char_p],
        "dmx_debug_device_div": [vp, vp, vp, C.c_int64, i32],
        "dmx_debug_log_rate": [i32, i32, i32, vp],
        "dmx_engine_get_cell_grids": [vp, vp, i32, vp],
        "dmx_debug_log_dd": [vp, vp, vp, vp, vp, C.c_int64],
        "dmx_resolve_tie_order": [vp, C.c_int64],
        "dmx_engine_mean_kernel_times": [vp, i32, vp],
        "dmx_store_add_batch": [vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, i32],
        "dmx_write_doublet_summary_grids": [vp, vp, vp, vp, C.c_char_p],
        "dmx_engine_kernel_names": [vp, vp],
        "dmx_engine_format_pair": [vp, vp, vp], "dmx_pair_text_get_info": [vp, vp], "dmx_pair_text_read": [vp, C.c_int64, C.c_int64, vp],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    L.dmx_store_new.restype = vp
    L.dmx_store_new.argtypes = []
    L.dmx_store_free.restype = None
    L.dmx_store_barcode.restype = C.c_char_p
    L.dmx_pair_text_free.restype = None
    L.dmx_pair_text_free.argtypes = [vp]
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise DmxError(rc, load().dmx_last_error().decode(errors="replace"))
    return rc
