#!/usr/bin/env python3
"""Generates demuxlet_amd/csrc/dmx_log_table.inc — the 128-entry {invc, logc} table of dmx_log() and the 256-entry table of
dmx_log2() (dmx_log.hpp; round 4: the doublet kernels' log — bins half as wide, |r| <= 2^-9, so log1p's series stops at r^6/6 and the
evaluation is one FP64 instruction shorter; OFF2 = OFF + 2^43 mantissa units keeps a bin centred on 1 with c = 1 exactly).

Design (ours; the classic table-driven log of Tang 1990, in the fixed-point-logc form popularised by modern libms):
  x = 2^k * z,  z in [OFF, 2*OFF),  OFF = 0x3FE5F000_00000000 (0.68652...), 128 bins of 2^45 mantissa units each, so that
  bin 80 = [1-2^-9, 1+2^-8) is CENTRED on 1 and gets c = 1 exactly (invc = 1, logc = 0: log(x) near 1 is the polynomial
  alone, no cancellation against a table value).
  For every other bin, c = 1/invc where invc is a double near 1/centre chosen among 2^18 candidates such that log(c) is
  within ~2^-66 of a multiple of 2^-43; logc is that multiple, so k*Ln2hi + logc is exact in binary64 (Ln2hi carries
  43 fractional bits) and the table's own rounding error is negligible.
  r = fma(z, invc, -1) = z/c - 1 exactly rounded once, |r| < 2^-8;  log(x) = k*ln2 + logc + log1p(r).
"""
import struct
import sys
from pathlib import Path

import mpmath as mp
import numpy as np

mp.mp.prec = 200
OFF = 0x3FE5F00000000000
N = 128
SHIFT = 45


def d2bits(x): return struct.unpack("<Q", struct.pack("<d", x))[0]
def bits2d(b): return struct.unpack("<d", struct.pack("<Q", b))[0]


def build_rows(off, n_bins, shift):
    """{invc, logc} per bin of z in [off, 2 off) cut into n_bins bins of 2^shift mantissa units; the bin that contains 1 gets c = 1."""
    two43 = mp.mpf(2) ** 43
    rows, worst, rmax, centre_bin = [], 0, 0, None
    for i in range(n_bins):
        lo_b, hi_b = off + (i << shift), off + ((i + 1) << shift)
        zlo, zhi = mp.mpf(bits2d(lo_b)), mp.mpf(bits2d(hi_b))
        if bits2d(lo_b) < 1.0 < bits2d(hi_b):
            rows.append((1.0, 0.0))
            rmax = max(rmax, float(1 - zlo), float(zhi - 1))
            centre_bin = i
            continue
        centre = (zlo + zhi) / 2
        L0 = mp.log(centre)
        Lstar = mp.nint(L0 * two43)
        # stage 1 (fast filter, x87 extended precision): ~2.6e5 candidate multiples around the centre
        ms = np.arange(-(1 << 17), 1 << 17, dtype=np.int64)
        Lm = (np.longdouble(int(Lstar)) + ms.astype(np.longdouble)) / np.longdouble(2.0 ** 43)
        inv_c = (np.longdouble(1) / np.exp(Lm)).astype(np.float64)
        lc_c = -np.log(inv_c.astype(np.longdouble)) * np.longdouble(2.0 ** 43)
        est = np.abs(lc_c - np.rint(lc_c)).astype(np.float64)
        cand = np.argsort(est)[:64]
        # stage 2 (exact): verify the shortlisted doubles with mpmath and keep the best
        best = None
        for ci in cand:
            inv = float(inv_c[ci])
            zl, zh = zlo * mp.mpf(inv) - 1, zhi * mp.mpf(inv) - 1
            if max(abs(zl), abs(zh)) > mp.mpf(2) ** -(52 - shift) * mp.mpf("1.0001"):       # keep |r| within the bin's half width
                continue
            lc = -mp.log(mp.mpf(inv)) * two43
            err = abs(lc - mp.nint(lc))
            if best is None or err < best[0]:
                best = (err, inv, float(mp.nint(lc) / two43))
        err, inv, logc = best
        worst = max(worst, float(err / two43))
        c = 1 / mp.mpf(inv)
        rmax = max(rmax, float(abs(zlo / c - 1)), float(abs(zhi / c - 1)))
        rows.append((inv, logc))
    return rows, worst, rmax, centre_bin


def build_rows32():
    """dmx_log2_lite32 (round 6, FAST's k_doublet_sym only): 32 bins of 2^47 mantissa units over [OFF32, 2 OFF32), OFF32 = 0x3FE64000_00000000;
    bin 19 = [1-2^-7, 1+2^-6) has c = 1.  The table is SPLIT — rc[32] then logc[32], 256 bytes each: a wavefront's ds_read_b64 of either array touches every LDS
    bank at most once (32 x 8 B = the 64 banks), whatever bins its lanes ask for: no bank conflicts, where the 256-bin table's 16-byte gathers cost ~6 extra
    LDS cycles per wavefront and evaluation (profiles/pmc_cfg3_fast.json).  FAST rounds k ln 2 + log c once, so logc is plain RN(-log rc), rc = RN(1 / centre).
    The polynomial is a weighted least-squares (Lawson) near-minimax fit of log1p(r) = r (1 - r/2 + a2 r^2 + ... + a5 r^5) over the table's r range."""
    off, nb, sh = 0x3FE6400000000000, 32, 47
    rc, lc, rmin, rmax, cb = [], [], mp.mpf(0), mp.mpf(0), None
    for i in range(nb):
        zlo, zhi = mp.mpf(bits2d(off + (i << sh))), mp.mpf(bits2d(off + ((i + 1) << sh)))
        if zlo < 1 < zhi:
            inv, cb = 1.0, i
        else:
            inv = float(1 / ((zlo + zhi) / 2))
        rc.append(inv)
        lc.append(float(-mp.log(mp.mpf(inv))) if inv != 1.0 else 0.0)
        rmin, rmax = min(rmin, zlo * mp.mpf(inv) - 1), max(rmax, zhi * mp.mpf(inv) - 1)
    M, n = 400, 5
    xs = [(rmin + rmax) / 2 + (rmax - rmin) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * M)) for k in range(M)]
    w = [mp.mpf(1)] * M
    for _ in range(60):
        A, b = mp.matrix(M, n - 1), mp.matrix(M, 1)
        for k, x in enumerate(xs):
            for c in range(2, n + 1):
                A[k, c - 2] = w[k] * x ** (c + 1)
            b[k] = w[k] * (mp.log1p(x) - x + x * x / 2)
        sol = mp.lu_solve(A.T * A, A.T * b)
        errs = [abs(x - x * x / 2 + sum(sol[c - 2] * x ** (c + 1) for c in range(2, n + 1)) - mp.log1p(x)) for x in xs]
        mx = max(errs)
        w = [w[k] * (errs[k] / mx + mp.mpf("0.05")) for k in range(M)]
        tot = sum(w)
        w = [v * M / tot for v in w]
    coef = [float(v) for v in sol]
    # the error of the ROUNDED coefficients on a dense grid (exact arithmetic; the evaluation's own roundings come on top: tests/test_dmx_log.py)
    worst = mp.mpf(0)
    for k in range(4001):
        x = rmin + (rmax - rmin) * k / 4000
        v = x - x * x / 2 + sum(mp.mpf(coef[c - 2]) * x ** (c + 1) for c in range(2, n + 1))
        worst = max(worst, abs(v - mp.log1p(x)))
    return rc, lc, coef, float(rmin), float(rmax), cb, float(worst)


def main():
    out = Path(__file__).resolve().parents[1] / "demuxlet_amd" / "csrc" / "dmx_log_table.inc"
    two43 = mp.mpf(2) ** 43
    rows, worst, rmax, cb = build_rows(OFF, N, SHIFT)
    assert cb == 80
    rows2, worst2, rmax2, cb2 = build_rows(OFF + (1 << 43), 256, 44)
    assert cb2 == 160 and rmax2 <= 2.0 ** -9 * 1.0001
    ln2 = mp.log(2)
    ln2hi = float(mp.nint(ln2 * two43) / two43)
    ln2lo = float(ln2 - mp.mpf(ln2hi))
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_log_table.py — do not edit.  {invc, logc} per bin; see dmx_log.hpp.\n")
        f.write(f"// worst |log(1/invc) - logc| = {worst:.3e}; max |r| = {rmax:.6e}\n")
        f.write(f"#define DMX_LOG_LN2HI {ln2hi.hex()}\n#define DMX_LOG_LN2LO {ln2lo.hex()}\n")
        f.write("static const double dmx_log_table_host[256] = {\n")
        for inv, logc in rows:
            f.write(f"  {inv.hex()}, {logc.hex()},\n")
        f.write("};\n")
        # second-order parts for the double-double evaluation dmx_log_dd() (the tie-order certificate, DESIGN.md "Ties"):
        # log(1/invc) - logc per bin (|.| <= ~2e-21, stored to 2^-53 relative) and the third piece of ln 2
        ln2lolo = float(ln2 - mp.mpf(ln2hi) - mp.mpf(ln2lo))
        f.write(f"#define DMX_LOG_LN2LOLO {ln2lolo.hex()}\n")
        f.write("static const double dmx_log_table_lo_host[128] = {\n")
        for inv, logc in rows:
            lo = float(-mp.log(mp.mpf(inv)) - mp.mpf(logc)) if inv != 1.0 else 0.0
            f.write(f"  {lo.hex()},\n")
        f.write("};\n")
        f.write("// dmx_log2(): 256 bins of 2^44 mantissa units over [OFF2, 2 OFF2), OFF2 = 0x3FE5F800_00000000; bin 160 = [1-2^-10, 1+2^-9) has c = 1\n")
        f.write(f"// worst |log(1/invc) - logc| = {worst2:.3e}; max |r| = {rmax2:.6e}\n")
        f.write("#define DMX_LOG2_OFF_HI 0x3FE5F800u\n")
        f.write("static const double dmx_log2_table_host[512] = {\n")
        for inv, logc in rows2:
            f.write(f"  {inv.hex()}, {logc.hex()},\n")
        f.write("};\n")
        rc32, lc32, coef32, rmin32, rmax32, cb32, worst32 = build_rows32()
        assert cb32 == 19
        f.write("// dmx_log2_lite32(): 32 bins of 2^47 mantissa units over [OFF32, 2 OFF32), OFF32 = 0x3FE64000_00000000; bin 19 = [1-2^-7, 1+2^-6) has c = 1;\n")
        f.write("// rc[32] then logc[32] (split: conflict-free 8-byte LDS reads); log1p(r) ~ r (1 - r/2 + A2 r^2 + A3 r^3 + A4 r^4 + A5 r^5)\n")
        f.write(f"// r in [{rmin32:.6e}, {rmax32:.6e}]; worst |polynomial - log1p| (exact arithmetic) = {worst32:.3e}\n")
        f.write("#define DMX_LOG32_OFF_HI 0x3FE64000u\n")
        for c, v in enumerate(coef32):
            f.write(f"#define DMX_LOG32_A{c + 2} {v.hex()}\n")
        f.write("static const double dmx_log32_table_host[64] = {\n")
        for v in rc32 + lc32:
            f.write(f"  {v.hex()},\n")
        f.write("};\n")
    print(f"wrote {out}: worst logc error {worst:.3e} / {worst2:.3e}, max|r| {rmax:.6e} / {rmax2:.6e}")


if __name__ == "__main__":
    main()
