"""CPU: accuracy of dmx_log()'s arithmetic (host emulation of the exact device operation sequence) against mpmath.
GPU (-m gpu): the device function agrees bit-for-bit with the host emulation (same IEEE operations)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    so = tmp_path_factory.mktemp("logemul") / "liblogemul.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", f"-I{ROOT / 'demuxlet_amd' / 'csrc'}",
                           str(ROOT / "tests" / "log_emul.cpp"), "-o", str(so), "-lm"])
    L = C.CDLL(str(so))
    L.dmx_log_emul_n.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.dmx_log_lite_emul_n.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.dmx_log2_emul_n.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.dmx_log2_lite32_emul_n.argtypes = [C.c_void_p, C.c_void_p, C.c_long]

    def f(x, lite=False, k2=False, lite32=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        (L.dmx_log2_lite32_emul_n if lite32 else L.dmx_log2_emul_n if k2 else (L.dmx_log_lite_emul_n if lite else L.dmx_log_emul_n))(x.ctypes.data, y.ctypes.data, len(x))
        return y
    return f


def sample_points(rng, n):
    xs = [np.exp(rng.uniform(np.log(1e-8), np.log(2.0), n)),          # the range the likelihood terms live in
          rng.uniform(0.97, 1.03, n),                                  # around 1: the polynomial-only bin and its neighbours
          1.0 + rng.uniform(-2 ** -8, 2 ** -7, n // 4),
          np.exp(rng.uniform(np.log(1e-300), np.log(1e300), n // 4))]
    # bin edges of the reduction (z = OFF + i*2^45 mantissa units) and their neighbours
    off = 0x3FE5F00000000000
    edges = np.array([off + (i << 45) + d for i in range(129) for d in (-1, 0, 1)], dtype=np.uint64).view(np.float64)
    return np.concatenate(xs + [edges, np.array([1.0, 0.5, 2.0, np.nextafter(1.0, 0), np.nextafter(1.0, 2)])])


def sample_points_k2(rng, n):
    """sample_points + the bin edges of dmx_log2's 256-bin reduction (z = OFF2 + i 2^44 mantissa units) and a dense sweep around 1."""
    off2 = 0x3FE5F80000000000
    edges = np.array([off2 + (i << 44) + d for i in range(257) for d in (-1, 0, 1)], dtype=np.uint64).view(np.float64)
    near1 = np.concatenate([1.0 + rng.uniform(-2 ** -10, 2 ** -9, n // 2), 1.0 + rng.uniform(-2 ** -9, 2 ** -8, n // 2),
                            1.0 + np.ldexp(rng.uniform(-1, 1, n // 4), -rng.integers(10, 50, n // 4))])
    return np.concatenate([sample_points(rng, n), edges, near1])


def ulp_stats(mp, x, y):
    """max and rms error of y against log(x) in ulps of the correctly rounded result; max |err| / max(|y|, 2^-7); all mpmath at 120 bits."""
    worst, worst_abs, sq = 0.0, 0.0, 0.0
    for xi, yi in zip(x, y):
        t = mp.log(mp.mpf(float(xi)))
        tf = float(t)
        if tf == 0.0:
            assert yi == 0.0
            continue
        d = float(abs(mp.mpf(float(yi)) - t))
        e = d / np.spacing(abs(tf))
        worst = max(worst, e)
        worst_abs = max(worst_abs, d / max(abs(tf), 2.0 ** -7))
        sq += e * e
    return worst, (sq / len(x)) ** 0.5, worst_abs


def test_k2_log_ulp_error_against_mpmath(emul):
    """dmx_log2 — the doublet kernels' log since round 4 (256 bins, log1p's series to r^6/6, 10 FP64 instructions): under 1 ulp on this
    sample, incl. the bin edges, the bin centred on 1 and its neighbours.  The 1e7-point run of the same check
    (tools/check_log2_accuracy.py, profiles/r04_log_accuracy.txt) finds the true maximum: 1.12 ulp just below 1 - 2^-10 (1.05 ulp for
    the 128-bin dmx_log just below 1 - 2^-9) — the bins next to the centre one, where log c and the polynomial cancel."""
    import mpmath as mp
    mp.mp.prec = 120
    rng = np.random.default_rng(4242)
    x = sample_points_k2(rng, 40000)
    worst, rms, worst_abs = ulp_stats(mp, x, emul(x, k2=True))
    print(f"dmx_log2 vs mpmath over {len(x)} points: max {worst:.3f} ulp, rms {rms:.3f} ulp, max |err|/max(|y|,2^-7) = {worst_abs:.2e}")
    assert worst < 1.0 and rms < 0.26 and worst_abs < 1.6e-16


def test_k2_log_matches_libm_to_one_ulp(emul):
    rng = np.random.default_rng(8)
    x = sample_points_k2(rng, 200000)
    y = emul(x, k2=True)
    ref = np.log(x)
    d = np.abs(y - ref) / np.spacing(np.abs(ref) + 1e-300)
    print(f"dmx_log2 vs glibc log: identical {np.mean(y == ref) * 100:.2f} %, max {d.max():.2f} ulp; vs dmx_log: identical {np.mean(y == emul(x)) * 100:.2f} %")
    assert d.max() <= 1.0 and np.mean(y == ref) > 0.95
    with np.errstate(all="ignore"):
        sp = emul(np.array([0.0, -1.0, np.inf, np.nan, 5e-324, 2.2250738585072014e-308]), k2=True)
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and np.isnan(sp[3]) and sp[4] == np.log(5e-324)


def test_compensated_form_against_mpmath(emul):
    """The compensated form of the table walk (w + r as a Fast2Sum, 15 FP64 instructions) that the STRICT kernels carried
    until late in round 2, kept in dmx_log.hpp for the record: max 0.75 ulp.  The kernels' log is the 11-instruction form
    tested below."""
    import mpmath as mp
    mp.mp.prec = 120
    rng = np.random.default_rng(77)
    x = sample_points(rng, 20000)
    y = emul(x, lite=True)
    worst, sq = 0.0, 0.0
    for xi, yi in zip(x, y):
        t = mp.log(mp.mpf(float(xi)))
        tf = float(t)
        if tf == 0.0:
            assert yi == 0.0
            continue
        e = float(abs(mp.mpf(float(yi)) - t)) / np.spacing(abs(tf))
        worst = max(worst, e)
        sq += e * e
    print(f"compensated form vs mpmath over {len(x)} points: max {worst:.3f} ulp, rms {(sq / len(x)) ** 0.5:.3f} ulp")
    assert worst < 0.8


def test_ulp_error_against_mpmath(emul):
    import mpmath as mp
    mp.mp.prec = 120
    rng = np.random.default_rng(2024)
    x = sample_points(rng, 40000)
    y = emul(x)
    worst, worst_abs, sq = 0.0, 0.0, 0.0
    for xi, yi in zip(x, y):
        t = mp.log(mp.mpf(float(xi)))
        tf = float(t)
        if tf == 0.0:
            assert yi == 0.0
            continue
        ulp = np.spacing(abs(tf))
        e = float(abs(mp.mpf(float(yi)) - t)) / ulp
        worst = max(worst, e)
        worst_abs = max(worst_abs, float(abs(mp.mpf(float(yi)) - t)) / max(abs(tf), 2.0 ** -7))
        sq += e * e
    rms = (sq / len(x)) ** 0.5
    print(f"dmx_log vs mpmath over {len(x)} points: max {worst:.3f} ulp, rms {rms:.3f} ulp, max |err|/max(|y|,2^-7) = {worst_abs:.2e}")
    assert worst < 1.0          # under 1 ulp everywhere sampled (glibc's own bound for log is ~0.52 ulp)
    assert worst_abs < 1.6e-16  # the bound DESIGN.md uses for the accumulated-difference estimate


def test_matches_libm_to_one_ulp(emul):
    rng = np.random.default_rng(7)
    x = sample_points(rng, 200000)
    y = emul(x)
    ref = np.log(x)
    d = np.abs(y - ref) / np.spacing(np.abs(ref) + 1e-300)
    print(f"vs glibc log: identical {np.mean(y == ref) * 100:.2f} %, max {d.max():.2f} ulp")
    assert d.max() <= 1.0


def test_special_values(emul):
    with np.errstate(all="ignore"):
        y = emul(np.array([0.0, -1.0, np.inf, np.nan, 5e-324, 2.2250738585072014e-308]))
    assert y[0] == -np.inf and np.isnan(y[1]) and y[2] == np.inf and np.isnan(y[3])
    assert y[4] == np.log(5e-324) and abs(y[5] - np.log(2.2250738585072014e-308)) <= np.spacing(708.0)


def lite32_edge_points():
    """Both ends of every bin of dmx_log2_lite32's reduction (z = OFF32 + i 2^47 mantissa units, OFF32 = 0x3FE64000...), 2^-40 inside, and the mid-points,
    at binary exponents 0, -1, -7, -20, -24: where |r| is largest and keeps one sign — the clustered-argument worst case of ADVICE r5."""
    off32 = 0x3FE6400000000000
    pts = []
    for i in range(32):
        lo, hi = off32 + (i << 47), off32 + ((i + 1) << 47)
        pts += [lo + (1 << 12), hi - (1 << 12), (lo + hi) // 2, lo, hi - 1]
    z = np.array(pts, dtype=np.uint64).view(np.float64)
    return np.concatenate([np.ldexp(z, k) for k in (0, -1, -7, -20, -24)])


def test_lite32_log_accuracy_bound(emul):
    """dmx_log2_lite32 (FAST k_doublet_sym's second log, round 6): the host emulation of the device's operation sequence against mpmath.  Budget stated in
    csrc/dmx_log.hpp: the polynomial is within 8.0e-16 of log1p over the table's r range; with the roundings of r, w and the final fma every value stays
    within 1.2e-15 + 1 ulp of the result ABSOLUTE — tighter than dmx_log2_lite's 6.9e-15, so kLiteLogMaxPairs covers a mix of the two."""
    import mpmath
    mpmath.mp.prec = 120
    rng = np.random.default_rng(41)
    x = np.concatenate([lite32_edge_points(), rng.uniform(1e-6, 1.0, 20000), np.exp(rng.uniform(np.log(1e-30), 0.0, 10000)), rng.uniform(0.9, 1.1, 10000)])
    y = emul(x, lite32=True)
    worst = 0.0
    for xi, yi in zip(x, y):
        t = mpmath.log(mpmath.mpf(float(xi)))
        worst = max(worst, float(abs(mpmath.mpf(float(yi)) - t)) - float(np.spacing(abs(float(t)))))
    assert worst < 1.2e-15, worst
    # next to no bias over the range likelihood terms live in
    lik = rng.uniform(1e-6, 1.0, 400000)
    err = emul(lik, lite32=True) - np.log(lik)
    assert abs(err.mean()) < 2e-16, err.mean()


@pytest.mark.gpu
def test_device_lite32_log_is_the_emulated_arithmetic(emul):
    """The device's dmx_log2_lite32 executes the emulated operation sequence: same bits on 3e5 points incl. every bin edge."""
    from demuxlet_amd import build, capi
    build.build()
    L = capi.load()
    rng = np.random.default_rng(299)
    x = np.concatenate([sample_points_k2(rng, 300000), lite32_edge_points()])
    x = x[np.isfinite(x) & (x >= 2.2250738585072014e-308)]
    y = np.empty_like(x)
    capi.check(L.dmx_debug_device_log2_lite32(x.ctypes.data, y.ctypes.data, len(x), 0))
    h = emul(x, lite32=True)
    assert np.array_equal(y, h), f"{np.sum(y != h)} of {len(x)} differ"


@pytest.mark.gpu
def test_device_log_is_the_emulated_arithmetic(emul):
    """The device executes exactly the operation sequence measured above (IEEE fma/mul/add are deterministic), so the
    host-side ulp measurement IS the device's accuracy."""
    import ctypes as C
    from demuxlet_amd import build, capi
    build.build()
    L = capi.load()
    rng = np.random.default_rng(99)
    x = sample_points(rng, 300000)
    y = np.empty_like(x)
    capi.check(L.dmx_debug_device_log(x.ctypes.data, y.ctypes.data, len(x), 0))
    h = emul(x)
    assert np.array_equal(y, h), f"{np.sum(y != h)} of {len(x)} differ"
    with np.errstate(all="ignore"):
        sp = np.array([0.0, -1.0, np.inf, np.nan, 5e-324])
        ys = np.empty_like(sp)
        capi.check(L.dmx_debug_device_log(sp.ctypes.data, ys.ctypes.data, len(sp), 0))
    assert ys[0] == -np.inf and np.isnan(ys[1]) and ys[2] == np.inf and np.isnan(ys[3]) and abs(ys[4] - np.log(5e-324)) < 1e-12


@pytest.mark.gpu
def test_device_k2_log_is_the_emulated_arithmetic(emul):
    """The same for dmx_log2, the doublet kernels' log."""
    from demuxlet_amd import build, capi
    build.build()
    L = capi.load()
    rng = np.random.default_rng(199)
    x = sample_points_k2(rng, 300000)
    y = np.empty_like(x)
    capi.check(L.dmx_debug_device_log2(x.ctypes.data, y.ctypes.data, len(x), 0))
    h = emul(x, k2=True)
    assert np.array_equal(y, h), f"{np.sum(y != h)} of {len(x)} differ"


@pytest.mark.gpu
def test_fast_mode_log_error_is_small_and_unbiased():
    """DMX_MODE_FAST's phase-2 log (dmx_log2_lite: 6 FP64 instructions, the series cut after r^4/4) — the budget csrc/dmx_log.hpp states: every value
    within 6.5e-15 ABSOLUTE of log(x), and next to no bias over the range likelihood terms live in (the first dropped term r^5/5 has the sign of r,
    which is symmetric inside a table bin; folding ln 2 into one constant leaves k x 2.3e-17 for a binary exponent k): the mean error over
    [1e-7, 1] stays below 2e-16, so the 1e5 terms of the deepest accumulator drift by < 2e-11 where FAST's contract is 1e-9."""
    import mpmath
    from demuxlet_amd import build, capi
    build.build()
    L = capi.load()
    rng = np.random.default_rng(77)
    x = np.concatenate([rng.uniform(1e-6, 1.0, 600000), np.exp(rng.uniform(np.log(1e-30), 0.0, 300000)), rng.uniform(0.9, 1.1, 100000)])
    y = np.empty_like(x)
    capi.check(L.dmx_debug_device_log2_lite(x.ctypes.data, y.ctypes.data, len(x), 0))
    ref = np.log(x)                                    # glibc: < 1 ulp, i.e. < 1.2e-16 relative — far below the budget
    err = y - ref
    assert np.abs(err).max() < 6.5e-15 + 4 * np.finfo(float).eps * np.abs(ref).max(), np.abs(err).max()
    lik = x >= 1e-7                                    # what a likelihood term can be: GL >= 1e-6 / (1 + 3e-6), genotype rows of a few 1e-1
    assert abs(err[lik].mean()) < 2e-16, err[lik].mean()
    mpmath.mp.prec = 100                               # and a few points against real arithmetic
    for v in (0.999, 0.5, 0.3333, 1e-3, 1e-6, 0.7071):
        got = np.empty(1); arg = np.array([v])
        capi.check(L.dmx_debug_device_log2_lite(arg.ctypes.data, got.ctypes.data, 1, 0))
        assert abs(float(mpmath.mpf(got[0]) - mpmath.log(mpmath.mpf(v)))) < 6.5e-15


@pytest.mark.gpu
def test_lite_log_worst_case_is_linear_and_bounded():
    """the lite log's error is unbiased only over VARIED arguments.  Clustered inputs — one argument repeated N times, as a PL field's handful of
    genotype triples and a dozen base-quality levels produce — collect the dropped r^5/5 term linearly.  Pinned here: at BOTH edges of every one of the 256
    table bins (where |r| is largest and keeps one sign) and for binary exponents down to 2^-24, one term is off by at most 6.9e-15 (5.7e-15 of series, the rest
    the result's own rounding), so N identical terms drift by less than 1e-9 up to N = 1.3e5 — the depth beyond which launch_doublet sends a FAST job to the
    STRICT kernels (kLiteLogMaxPairs = 130 000)."""
    import mpmath
    from demuxlet_amd import build, capi
    build.build()
    L = capi.load()
    mpmath.mp.prec = 120
    # bin edges of the reduced argument: the table starts at OFF = 0x3FE5F800... = 2^-1 x (1 + 95.5/256), so in every octave the edges sit at
    # mantissa (i + 0.5) / 256; the points: 2^-40 inside either end of each bin (|r| largest, one sign per end), and the mid-points for contrast
    i = np.arange(-1, 256)
    lo_edge = 1.0 + (i + 0.5) / 256.0 + 2.0 ** -40
    hi_edge = 1.0 + (i + 1.5) / 256.0 - 2.0 ** -40
    z = np.concatenate([lo_edge, hi_edge, 1.0 + (i + 1.0) / 256.0])
    z = z[(z >= 1.0) & (z < 2.0)]
    worst = 0.0
    for k in (0, -1, -7, -20, -24):                               # x = z * 2^k: likelihood terms live in [2^-24, 1]
        x = np.ldexp(z, k - 1)
        y = np.empty_like(x)
        capi.check(L.dmx_debug_device_log2_lite(x.ctypes.data, y.ctypes.data, len(x), 0))
        for xi, yi in zip(x, y):
            worst = max(worst, abs(float(mpmath.mpf(float(yi)) - mpmath.log(mpmath.mpf(float(xi))))))
    assert worst < 6.9e-15, worst
    assert 130000 * worst < 1e-9
    print(f"lite log, clustered arguments at the bin edges: worst |error| of one term {worst:.3e} -> {130000 * worst:.2e} over 130 000 identical terms")


@pytest.mark.gpu
def test_device_shared_reciprocal_division_is_ieee():
    """The kernels divide three (nine) numerators by one denominator with a shared Newton-refined reciprocal; the result
    must be the correctly rounded quotient, i.e. numpy's / bit for bit, over the operand ranges the kernels see."""
    from demuxlet_amd import build, capi
    build.build()
    L = capi.load()
    rng = np.random.default_rng(123)
    n = 400000
    a = np.concatenate([rng.random(n), np.exp(rng.uniform(np.log(1e-200), 0, n)), rng.random(n) * 1e-6 + 1e-6])
    b = np.concatenate([rng.uniform(0.3, 3.0, n), np.exp(rng.uniform(np.log(1e-6), np.log(3.0), n)), 1.0 + rng.random(n) * 3e-6])
    q = np.empty_like(a)
    capi.check(L.dmx_debug_device_div(a.ctypes.data, b.ctypes.data, q.ctypes.data, len(a), 0))
    assert np.array_equal(q, a / b), f"{np.sum(q != a / b)} of {len(a)} quotients differ from IEEE division"


def test_double_double_log_and_libm_brackets():
    """dmx_log_dd / dmx_log_bracket (csrc/dmx_log.hpp; host evaluation of the code the device runs): hi + lo = log(x) to better
    than 2^-65 relative (50-digit check), and the bracket [t_lo, t_hi] — what a libm with < 0.55 ulp error (0.67 ulp in glibc's
    near-1 interval) can return — contains both the correctly rounded value and THIS host's log(), on likelihood-like arguments.
    It is what lets the device certify the reference's accumulators bit for bit (DESIGN.md "Ties")."""
    import mpmath
    from demuxlet_amd import build, capi
    build.build()
    L = capi.load()
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0.93, 1.07, 300000), rng.uniform(0.3, 1.0, 300000), rng.uniform(1e-6, 1e-3, 200000),
                        np.ldexp(rng.uniform(0.5, 1.0, 200000), -rng.integers(0, 60, 200000))])
    n = len(x)
    hi, lo, tl, th = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    capi.check(L.dmx_debug_log_dd(x.ctypes.data, hi.ctypes.data, lo.ctypes.data, tl.ctypes.data, th.ctypes.data, n))
    y = np.log(x)
    assert ((y >= tl) & (y <= th)).all()                            # the host libm never leaves the bracket
    assert (tl <= hi).all() and (hi <= th).all()                      # nor does the correctly rounded value
    amb = tl != th
    assert (np.abs(th[amb] - tl[amb]) <= 1.0001 * np.spacing(np.minimum(np.abs(tl[amb]), np.abs(th[amb])))).all()    # neighbours
    assert 0.02 < amb.mean() < 0.30
    mpmath.mp.prec = 200
    worst = -999.0
    for i in rng.choice(n, 2000, replace=False):
        t = mpmath.log(mpmath.mpf(float(x[i])))
        err = abs(mpmath.mpf(float(hi[i])) + mpmath.mpf(float(lo[i])) - t)
        if err > 0:
            worst = max(worst, float(mpmath.log(err / abs(t), 2)))
        assert float(t) == hi[i] or abs(abs(lo[i]) - 0.5 * np.spacing(abs(hi[i]))) < 1e-3 * np.spacing(abs(hi[i]))    # hi = RN(log x)
    print(f"dmx_log_dd: worst relative error 2^{worst:.1f}; {100 * amb.mean():.1f} % of the arguments leave two candidates")
    assert worst < -65
