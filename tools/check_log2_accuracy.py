"""dmx_log2 (the doublet kernels' log, csrc/dmx_log.hpp) against mpmath at 120 bits on N arguments (default 1e7): likelihood-range values,
a dense sweep around 1 (the bin with c = 1, its neighbours, arguments 2^-10 ... 2^-50 from 1), every bin edge of the 256-bin reduction with
its neighbours, and the whole binary64 range.  The arithmetic evaluated is the host emulation of the exact device operation sequence
(tests/log_emul.cpp; tests/test_dmx_log.py::test_device_k2_log_is_the_emulated_arithmetic shows the device produces the same bits).
    python tools/check_log2_accuracy.py [n_points] [n_processes] [2 | 0]    -> one summary line (2: dmx_log2, the default; 0: the 128-bin dmx_log)"""
import ctypes as C
import subprocess
import sys
import tempfile
from multiprocessing import Pool
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def points(seed, n):
    rng = np.random.default_rng(seed)
    off2 = 0x3FE5F80000000000
    edges = np.array([off2 + (i << 44) + d for i in range(257) for d in (-2, -1, 0, 1, 2)], dtype=np.uint64).view(np.float64)
    return np.concatenate([np.exp(rng.uniform(np.log(1e-8), np.log(2.0), n // 2)), rng.uniform(0.97, 1.03, n // 8),
                           1.0 + rng.uniform(-2 ** -10, 2 ** -9, n // 8), 1.0 + rng.uniform(-2 ** -9, 2 ** -8, n // 16),
                           1.0 + np.ldexp(rng.uniform(-1, 1, n // 16), -rng.integers(10, 50, n // 16)),
                           np.exp(rng.uniform(np.log(1e-300), np.log(1e300), n // 8)), edges])


def work(args):
    seed, n, so, which = args
    import mpmath as mp
    mp.mp.prec = 120
    L = C.CDLL(so)
    fn = L.dmx_log2_emul_n if which == 2 else L.dmx_log_emul_n
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    x = np.ascontiguousarray(points(seed, n))
    y = np.empty_like(x)
    fn(x.ctypes.data, y.ctypes.data, len(x))
    worst, sq, cnt, same = 0.0, 0.0, 0, int(np.sum(y == np.log(x)))
    wx = 0.0
    for xi, yi in zip(x, y):
        t = mp.log(mp.mpf(float(xi)))
        tf = float(t)
        if tf == 0.0:
            assert yi == 0.0
            continue
        e = float(abs(mp.mpf(float(yi)) - t)) / np.spacing(abs(tf))
        if e > worst:
            worst, wx = e, float(xi)
        sq += e * e
        cnt += 1
    return worst, sq, cnt, same, len(x), wx


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # 2: dmx_log2 (doublet kernels), 0: dmx_log (singlet kernels, 128 bins)
    td = tempfile.mkdtemp()
    so = str(Path(td) / "liblogemul.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", f"-I{ROOT / 'demuxlet_amd' / 'csrc'}", str(ROOT / "tests" / "log_emul.cpp"), "-o", so, "-lm"])
    chunks = max(procs, n // 250_000)
    with Pool(procs) as pool:
        res = pool.map(work, [(1000 + i, n // chunks, so, which) for i in range(chunks)])
    worst = max(r[0] for r in res)
    tot = sum(r[2] for r in res)
    npts = sum(r[4] for r in res)
    wx = max(res, key=lambda r: r[0])[5]
    print(f"{'dmx_log2' if which == 2 else 'dmx_log'} vs mpmath (120 bits) on {npts} arguments: max {worst:.4f} ulp (at x = {wx!r} = {wx.hex()}), rms {(sum(r[1] for r in res) / tot) ** 0.5:.4f} ulp; "
          f"identical to this host's glibc log() on {100.0 * sum(r[3] for r in res) / npts:.2f} %")
    assert worst < 1.25
