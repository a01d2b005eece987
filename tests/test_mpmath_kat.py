"""An independent, arbitrary-precision pin of the MATHEMATICS of rows a4-a14.

The reference slice harness that produced tests/golden goes through I/O stand-ins (htslib is absent), so by the rubric it
pins nothing.  This file evaluates cmd_cram_demuxlet.cpp:390-401 (gp0s), :412-461 (singlet accumulation) and :594-710 (pG,
doublet grid, llks00) in mpmath at 50 digits — real-number arithmetic, no binary64 rounding anywhere, written from the
reference's formulas and not from oracle/dmx_oracle.c — on the SURVEY §4 micro known-answer case and on a 50-barcode problem, and
checks the oracle (CPU) and the HIP engine (GPU, STRICT and FAST) against it to 1e-12.  What it cannot show is the reference's
bit-level summation order (the golden fixtures and the order argument of DESIGN.md §4 cover that); what it does show is that
oracle and kernels compute the right numbers, independently of any stand-in."""
import mpmath
import numpy as np
import pytest

from golden_util import Golden

TOL = 1e-12


def mp_reference(csr, g, alphas):
    """llks[B][V], llk0s[B], llksAB[B][V][V][A], llks00[B][A] as exact real arithmetic sees the reference's formulas."""
    mpmath.mp.dps = 50
    mpf = mpmath.mpf
    S, V, _ = g.shape
    A = len(alphas)
    G = [[[mpf(float(g[s, k, l])) for l in range(3)] for k in range(V)] for s in range(S)]         # the float32 values, exactly
    gp0 = [[sum(G[s][k][l] for k in range(V)) / V for l in range(3)] for s in range(S)]             # :390-401
    err = lambda bq: mpf(10) ** (-mpf(bq) / 10) if bq > 1 else mpf("0.75")                            # PhredHelper.cpp:24-40
    B = csr.n_cells
    llks = np.zeros((B, V)); llk0s = np.zeros(B); grid = np.zeros((B, V, V, A)); l00 = np.zeros((B, A))
    eps = mpf("1e-6")
    for c in range(B):
        acc = [mpf(0)] * V; acc0 = mpf(0)
        accAB = [[[mpf(0)] * A for _ in range(V)] for _ in range(V)]; acc00 = [mpf(0)] * A
        for p in range(int(csr.cell_off[c]), int(csr.cell_off[c + 1])):
            s = int(csr.pair_snp[p])
            gl = [mpf(1)] * 3                                                                       # :427
            pG = [[[mpf(1)] * 3 for _ in range(3)] for _ in range(A)]                               # :597
            nreads = 0
            for w in csr.words[int(csr.pair_off[p]):int(csr.pair_off[p + 1])]:
                al, bq = (int(w) >> 24) & 0xFF, (int(w) >> 16) & 0xFF
                if al == 2:
                    continue                                                                        # :435, :604
                nreads += 1
                e3, mat = err(bq) / 3, 1 - err(bq)
                gl = [gl[0] * (mat if al == 0 else e3), gl[1] * (mpf("0.5") - e3), gl[2] * (mat if al == 1 else e3)]   # :437-439
                pR, pA = (mat if al == 0 else e3), (mat if al == 1 else e3)                         # :606-607
                for n in range(A):
                    for l in range(3):
                        for m in range(3):
                            q = mpf("0.5") * l + (m - l) * mpf("0.5") * mpf(float(alphas[n]))      # :613
                            pG[n][l][m] *= (pR * (1 - q) + pA * q)                                  # :625
            if nreads:
                t = sum(gl); gl = [x / t for x in gl]                                               # the per-read renormalisations telescope
                mx = max(x for a in pG for r in a for x in r)
                pG = [[[x / mx for x in r] for r in a] for a in pG]
            gl = [x + eps for x in gl]; t = sum(gl); gl = [x / t for x in gl]                       # :446-452
            pG = [[[x + eps for x in r] for r in a] for a in pG]                                    # :649
            mx = max(x for a in pG for r in a for x in r)
            pG = [[[x / mx for x in r] for r in a] for a in pG]                                     # :656-663
            for k in range(V):
                acc[k] += mpmath.log(sum(gl[l] * G[s][k][l] for l in range(3)))                     # :456
            acc0 += mpmath.log(sum(gl[l] * gp0[s][l] for l in range(3)))                            # :459
            for n in range(A):
                for j in range(V):
                    for k in range(V):
                        accAB[j][k][n] += mpmath.log(sum(G[s][j][l] * G[s][k][m] * pG[n][l][m] for l in range(3) for m in range(3)))   # :677-683
                acc00[n] += mpmath.log(sum(gp0[s][l] * gp0[s][m] * pG[n][l][m] for l in range(3) for m in range(3)))                  # :702-709
        llks[c] = [float(x) for x in acc]; llk0s[c] = float(acc0)
        grid[c] = [[[float(x) for x in r] for r in a] for a in accAB]; l00[c] = [float(x) for x in acc00]
    return llks, llk0s, grid, l00


def problems(oracle):
    gd = Golden("kat_micro")
    csr = oracle.store_from_events(gd.problem(oracle).events)
    yield "kat_micro (SURVEY §4)", csr, gd.g, gd.alphas
    from demuxlet_amd import synth
    rng = np.random.default_rng(4242)
    B, S, V = 50, 120, 4
    raw = synth.make_raw_genotypes(rng, S, V)
    g = np.stack([oracle.geno_from_gp(x, 0.01) for x in synth.raw_gp_from_alleles(rng, raw.alleles)])
    sp = synth.make_pileup(rng, raw.alleles, B, 0.5, 1.8, dense_layout=False, doublet_rate=0.3)
    words = ((sp.reads >> 7).astype(np.uint32) << 24) | ((sp.reads & 0x7F).astype(np.uint32) << 16) | 1
    csr2 = oracle.Csr([f"c{i:03d}" for i in range(B)], sp.cell_pair_off, sp.pair_snp, np.concatenate([[0], np.cumsum(sp.pair_nrd.astype(np.int64))]),
                      words.astype(np.uint32), sp.rd_totl, sp.rd_pass, sp.rd_uniq)
    yield "50 barcodes x 120 SNPs x 4 samples, GP, alpha {0, 0.25, 0.5}", csr2, g, (0.0, 0.25, 0.5)


@pytest.fixture(scope="module")
def exact(oracle):
    return [(name, csr, g, al, mp_reference(csr, g, al)) for name, csr, g, al in problems(oracle)]


def test_oracle_against_50_digit_arithmetic(oracle, exact):
    for name, csr, g, al, (llks, llk0s, grid, l00) in exact:
        out = oracle.run_csr(csr, [f"s{j}" for j in range(g.shape[1])], g, oracle.Params(tuple(al), 0.5))
        cov = np.diff(csr.cell_off) > 0
        d = max(np.abs(out.llks - llks).max(), np.abs(out.llk0s - llk0s).max(), np.abs(out.llksAB[cov] - grid[cov]).max(),
                np.abs(out.llks00[cov] - l00[cov]).max())
        print(f"{name}: oracle vs 50-digit arithmetic, max |delta| = {d:.2e} (|llk| up to {np.abs(grid).max():.1f})")
        assert d < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_engine_against_50_digit_arithmetic(oracle, exact, mode):
    from demuxlet_amd import build, capi, engine
    from golden_util import printed_mask
    build.build()
    for name, csr, g, al, (llks, llk0s, grid, l00) in exact:
        reads = (((csr.words >> 24) & 0xFF).astype(np.uint8) << 7) | ((csr.words >> 16) & 0x7F).astype(np.uint8)
        keep = ((csr.words >> 24) & 0xFF) != 2
        nrd = np.array([int(keep[csr.pair_off[p]:csr.pair_off[p + 1]].sum()) for p in range(len(csr.pair_snp))], dtype=np.uint8)
        cell_read_off = np.concatenate([[0], np.cumsum([nrd[csr.cell_off[c]:csr.cell_off[c + 1]].sum() for c in range(csr.n_cells)])]).astype(np.int64)
        pl = engine.HostPileup(csr.n_cells, g.shape[0], csr.cell_off, cell_read_off, csr.pair_snp, nrd, reads[keep], csr.rd_totl, csr.rd_pass, csr.rd_uniq)
        e = engine.Engine(g.shape[1], al, 0.5, mode=capi.DMX_MODE_FAST if mode == "fast" else capi.DMX_MODE_STRICT)
        e.set_genotypes(g); e.set_pileup(pl)
        e.run_singlet(); e.run_doublet()
        a, b = e.get_singlet()
        gr, l0, _ = e.get_doublet()
        e.close()
        cov = np.diff(csr.cell_off) > 0
        dg = np.abs(gr[cov] - grid[cov])
        if mode == "fast":                       # FAST computes the printed entries (for the default grid {0, 0.5})
            dg = dg[np.broadcast_to(printed_mask(g.shape[1], len(al))[None], dg.shape)] if tuple(al) == (0.0, 0.5) else dg
        d = max(np.abs(a - llks).max(), np.abs(b - llk0s).max(), dg.max(), np.abs(l0[cov] - l00[cov]).max())
        print(f"{name}: HIP engine ({mode}) vs 50-digit arithmetic, max |delta| = {d:.2e}")
        assert d < TOL
