"""CPU: the product's HOST logic through the C-ABI (no GPU): phred LUT (a2), genotype transforms (a3), UMI store (a1),
finaliser/writers (a6, a10..a14) and the tie arbiter — against the reference's own outputs (tests/golden) and the oracle."""
import numpy as np
import pytest

from golden_util import CASES, GOLDEN, Golden


@pytest.fixture(scope="module")
def eng():
    from demuxlet_amd import build, engine
    build.build()
    return engine


def test_phred_tables_match_reference(eng):
    z = np.load(GOLDEN / "ref_units.npz")
    mat, err = eng.phred_tables()
    assert np.array_equal(mat, z["phred_mat"]) and np.array_equal(err, z["phred_err"])


def test_store_trace_matches_reference_units(eng):
    """sc_drop_seq.cpp compiled alone (ref_units.npz): return values, ids, counters, iteration order."""
    z = np.load(GOLDEN / "ref_units.npz")
    st = eng.Store()
    for _ in range(12):
        st.add_snp()
    rets, ids = [], []
    for c, s, u, a, b in zip(z["ev_cell"], z["ev_snp"], z["ev_umi"], z["ev_allele"], z["ev_bq"]):
        cid = st.add_cell(str(c))
        ids.append(cid)
        rets.append(int(st.add_read(int(s), cid, str(u), int(a), int(b))))
    assert np.array_equal(rets, z["ret_new"]) and np.array_equal(ids, z["ret_cellid"])
    pl = st.freeze()
    cnt = z["counters"]
    assert np.array_equal(pl.rd_pass, cnt[:, 0]) and np.array_equal(pl.rd_uniq, cnt[:, 1])
    assert np.array_equal(pl.n_snp_per_cell, cnt[:, 2]) and np.array_equal(pl.n_snp_per_cell, z["cell_npairs"])
    assert np.array_equal(pl.pair_snp, z["flat_snp"])
    words = z["flat_words"]
    al, bq = (words >> 24) & 0xFF, (words >> 16) & 0xFF
    keep = al != 2
    assert np.array_equal(pl.reads, ((al[keep] << 7) | bq[keep]).astype(np.uint8))
    pair_of_word = np.repeat(np.arange(len(z["flat_nper"])), z["flat_nper"])
    assert np.array_equal(pl.pair_nrd, np.bincount(pair_of_word[keep], minlength=len(z["flat_nper"])))


@pytest.mark.parametrize("threads,pieces", [(1, 1), (4, 1), (8, 3), (3, 7)])
def test_store_batch_insert_equals_the_single_calls(eng, threads, pieces):
    """dmx_store_add_batch (observations of different cell shards inserted on different host threads) against the reference trace
    (sc_drop_seq.cpp compiled alone) and against dmx_store_add_read called one by one: same return values, counters and CSR — also
    when the trace arrives in several batches and single calls are mixed in between."""
    z = np.load(GOLDEN / "ref_units.npz")
    n = len(z["ev_cell"])
    st = eng.Store()
    for _ in range(12):
        st.add_snp()
    ids = np.array([st.add_cell(str(c)) for c in z["ev_cell"]], dtype=np.int32)
    assert np.array_equal(ids, z["ret_cellid"])
    rets = np.zeros(n, dtype=np.int64)
    cuts = np.linspace(0, n, pieces + 1).astype(int)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b - a > 2:                                 # the last item of a piece goes through the single call
            rets[a:b - 1] = st.add_batch(z["ev_snp"][a:b - 1], ids[a:b - 1], [str(u) for u in z["ev_umi"][a:b - 1]], z["ev_allele"][a:b - 1],
                                         z["ev_bq"][a:b - 1], n_threads=threads)
            a = b - 1
        for i in range(a, b):
            rets[i] = int(st.add_read(int(z["ev_snp"][i]), int(ids[i]), str(z["ev_umi"][i]), int(z["ev_allele"][i]), int(z["ev_bq"][i])))
    assert np.array_equal(rets, z["ret_new"])
    pl = st.freeze()
    cnt = z["counters"]
    assert np.array_equal(pl.rd_pass, cnt[:, 0]) and np.array_equal(pl.rd_uniq, cnt[:, 1]) and np.array_equal(pl.n_snp_per_cell, cnt[:, 2])
    assert np.array_equal(pl.pair_snp, z["flat_snp"])
    words = z["flat_words"]
    al, bq = (words >> 24) & 0xFF, (words >> 16) & 0xFF
    keep = al != 2
    assert np.array_equal(pl.reads, ((al[keep] << 7) | bq[keep]).astype(np.uint8))


def test_store_batch_insert_on_a_large_random_trace(eng):
    """2e5 observations over 3 000 barcodes with 30 % repeated keys: the batch path on 8 threads == the single calls."""
    rng = np.random.default_rng(11)
    n, B, S = 200_000, 3000, 5000
    cell = rng.integers(0, B, n).astype(np.int32); snp = rng.integers(0, S, n).astype(np.int32)
    umi = [f"U{x:05d}" for x in rng.integers(0, 40, n)]
    dup = rng.random(n) < 0.3
    src = rng.integers(0, n, n)
    for i in np.flatnonzero(dup):
        j = src[i] % (i + 1)
        cell[i], snp[i], umi[i] = cell[j], snp[j], umi[j]
    allele = rng.integers(0, 3, n).astype(np.uint8); bq = rng.integers(2, 41, n).astype(np.uint8)
    stores = []
    for mode in ("single", "batch"):
        st = eng.Store()
        for _ in range(S):
            st.add_snp()
        for c in range(B):
            assert st.add_cell(f"BC{c:05d}") == c
        if mode == "single":
            r = np.array([int(st.add_read(int(snp[i]), int(cell[i]), umi[i], int(allele[i]), int(bq[i]))) for i in range(n)])
        else:
            r = np.concatenate([st.add_batch(snp[a:a + 70_000], cell[a:a + 70_000], umi[a:a + 70_000], allele[a:a + 70_000], bq[a:a + 70_000], n_threads=8)
                                for a in range(0, n, 70_000)])
        stores.append((r, st.freeze()))
    (r0, p0), (r1, p1) = stores
    assert np.array_equal(r0, r1) and r0.sum() < n
    for f in ("cell_pair_off", "cell_read_off", "pair_snp", "pair_nrd", "reads", "rd_pass", "rd_uniq"):
        assert np.array_equal(getattr(p0, f), getattr(p1, f)), f


def build_store(eng, pb):
    st = eng.Store()
    for _ in range(pb.n_snps):
        st.add_snp()
    ev = pb.events
    for e in range(len(ev.barcode)):
        c = st.add_cell(ev.barcode[e])
        if ev.newread[e]:
            st.count_read(c)
        if ev.snp[e] >= 0:
            st.add_read(int(ev.snp[e]), c, ev.umi[e], int(ev.allele[e]), int(ev.bq[e]))
    return st


@pytest.mark.parametrize("name", CASES)
def test_store_matches_oracle_csr(eng, oracle, name):
    gd = Golden(name)
    pb = gd.problem(oracle)
    csr = oracle.store_from_events(pb.events)
    st = build_store(eng, pb)
    pl = st.freeze()
    assert st.barcodes() == csr.barcodes == gd.ref_barcodes
    assert np.array_equal(pl.cell_pair_off, csr.cell_off)
    assert np.array_equal(pl.pair_snp, csr.pair_snp)
    al, bq = (csr.words >> 24) & 0xFF, (csr.words >> 16) & 0xFF
    keep = al != 2
    assert np.array_equal(pl.reads, ((al[keep] << 7) | bq[keep]).astype(np.uint8))
    pair_of_word = np.repeat(np.arange(len(csr.pair_snp)), np.diff(csr.pair_off))
    assert np.array_equal(pl.pair_nrd, np.bincount(pair_of_word[keep], minlength=len(csr.pair_snp)))
    cnt = gd.z["ref_counters"]
    assert np.array_equal(pl.rd_totl, cnt[:, 0]) and np.array_equal(pl.rd_pass, cnt[:, 1]) and np.array_equal(pl.rd_uniq, cnt[:, 2])
    rd_off = np.concatenate([[0], np.cumsum(np.bincount(np.searchsorted(csr.cell_off, pair_of_word[keep], side="right") - 1,
                                                       minlength=csr.n_cells))])
    assert np.array_equal(pl.cell_read_off, rd_off)


def final_args(eng, gd, n_snp):
    cnt = gd.z["ref_counters"]
    return eng.FinalArgs(gd.ref_barcodes, gd.sample_ids, gd.alphas, gd.doublet_prior, cnt[:, 0], cnt[:, 1], cnt[:, 2], n_snp,
                         gd.min_total, gd.min_uniq, gd.min_snp, gd.write_pair)


@pytest.mark.parametrize("name", CASES)
def test_finaliser_reproduces_reference_files_from_reference_arrays(eng, name, tmp_path):
    """The writers alone: reference raw arrays in, the reference's four files out, byte for byte."""
    gd = Golden(name)
    fa = final_args(eng, gd, gd.z["ref_counters"][:, 3])
    eng.write_single(fa, gd.z["ref_llks"], gd.z["ref_llk0s"], str(tmp_path / "o.single"))
    eng.write_doublet(fa, gd.z["ref_llksAB"], gd.z["ref_llks00"], str(tmp_path / "o"))
    for suf, ref in gd.files.items():
        assert (tmp_path / f"o.{suf}").read_bytes() == ref, suf
    assert (tmp_path / "o.pair").exists() == gd.write_pair


@pytest.mark.parametrize("name", ["gt_v4_a2_pair", "gp_v8_a2_minsnp", "gt_v5_dense", "gt_v3_alpha_quirk"])
def test_tie_arbiter_restores_reference_order_under_noise(eng, oracle, name, tmp_path):
    """Perturb the grid at the 1e-11 level (what a different log() does) — without the arbiter some DBL-a-b rows flip to
    DBL-b-a; with it (host re-evaluation of the near-tied entries) .best is the reference's again."""
    gd = Golden(name)
    pb = gd.problem(oracle)
    st = build_store(eng, pb)
    pl = st.freeze()
    fa = final_args(eng, gd, pl.n_snp_per_cell)
    rng = np.random.default_rng(5)
    grid = gd.z["ref_llksAB"] + rng.normal(0, 2e-11, size=gd.z["ref_llksAB"].shape) * gd.z["ref_processed"][:, None, None, None]
    eng.write_doublet(fa, grid, gd.z["ref_llks00"], str(tmp_path / "noisy"))
    eng.write_doublet(fa, grid, gd.z["ref_llks00"], str(tmp_path / "fixed"), tie_pileup=pl, tie_g=gd.g)
    ref = gd.files["best"].decode().splitlines()
    noisy = (tmp_path / "noisy.best").read_text().splitlines()
    fixed = (tmp_path / "fixed.best").read_text().splitlines()
    col = lambda rows, i: [r.split("\t")[i] for r in rows]
    assert col(fixed, 5) == col(ref, 5)                       # BEST
    assert col(fixed, 11) == col(ref, 11) and col(fixed, 12) == col(ref, 12)   # DBL.1ST DBL.2ND
    assert fixed == ref
    if name == "gt_v4_a2_pair":
        assert col(noisy, 11) != col(ref, 11), "the noise should have flipped at least one doublet order in this fixture"


def test_geno_transforms_match_oracle(eng, oracle):
    """Row a3.  PARITY-UNPINNED by the reference (parse_posteriors needs htslib): product vs oracle restatement plus
    hand-derived values."""
    rng = np.random.default_rng(11)
    for V in (1, 2, 5, 32):
        a = (rng.random((V, 2)) < 0.4).astype(np.int32)
        a[rng.random(V) < 0.2] = -1
        a[rng.random(V) < 0.1, 1] = -1
        for e in (0.0, 0.01, 0.1):
            assert np.array_equal(eng.geno_from_gt(a, e), oracle.geno_from_gt(a, e))
        pl = rng.integers(0, 256, size=(V, 3)).astype(np.int32)
        pl[rng.random(V) < 0.2] = np.iinfo(np.int32).min
        assert np.array_equal(eng.geno_from_pl(pl), oracle.geno_from_pl(pl))
        gp = rng.random((V, 3)).astype(np.float32)
        for e in (0.0, 0.01):
            assert np.array_equal(eng.geno_from_gp(gp, e), oracle.geno_from_gp(gp, e))
    # hand-derived: GT 0/1 at eps=0.01 -> (0.005, 0.99, 0.005) as float32
    g = eng.geno_from_gt(np.array([[0, 1]]), 0.01)
    assert np.array_equal(g, np.array([[np.float32(0.005), np.float32(0.99), np.float32(0.005)]]))
    # missing genotype with one called het sample: an=2, ac=(1,1) -> af=(1.5/3) each -> HWE (0.25, 0.5, 0.25)
    g = eng.geno_from_gt(np.array([[0, 1], [-1, -1]]), 0.01)
    assert np.allclose(g[1], [0.25, 0.5, 0.25], atol=1e-7)
    # GP: one sample (1,0,0): normalised (1,0,0); mean = ((.25,.5,.25)+(1,0,0))/2; out = .99*gp + .01*mean
    g = eng.geno_from_gp(np.array([[1.0, 0.0, 0.0]], dtype=np.float32), 0.01)
    assert np.allclose(g[0], [0.99 + 0.01 * 0.625, 0.01 * 0.25, 0.01 * 0.125], atol=1e-7)
    # PL (0,30,60) for every sample: posterior concentrates on hom-ref, sums to 1
    g = eng.geno_from_pl(np.tile(np.array([[0, 30, 60]], dtype=np.int32), (4, 1)))
    assert np.allclose(g.sum(axis=1), 1.0, atol=1e-6) and (g[:, 0] > 0.99).all()


def test_store_rejects_bad_arguments(eng):
    from demuxlet_amd import capi
    st = eng.Store()
    st.add_snp()
    c = st.add_cell("A")
    for args in ((1, c, "u", 0, 30), (0, 5, "u", 0, 30), (0, c, "u", 3, 30), (0, c, "u", 0, 200)):
        with pytest.raises(capi.DmxError):
            st.add_read(*args)
    pl = st.freeze()
    assert pl.n_cells == 1 and len(pl.pair_nrd) == 0


def test_row_formatter_equals_printf(eng, tmp_path, monkeypatch):
    """The writers format `%.5lf`/`%.4lf`/`%d` columns without printf (exact 128-bit decimal rounding, csrc/dmx_host.cpp
    put_fixed) and on several host threads; the bytes must be what printf produces.  Python's % operator is correctly
    rounded (round-half-even on the exact binary value) like glibc's printf, so it is the checker.  Covered: exact ties at
    the 5th decimal (k/2^n), values that round up across a power of ten, negative zero, subnormals, |v| >= 2^43 (printf
    fallback), and a few hundred thousand random magnitudes."""
    rng = np.random.default_rng(7)
    ties = np.array([k / 2.0 ** n for n in range(1, 20) for k in (1, 3, 5, 7, 2 ** n - 1, 2 ** n + 1)])
    edge = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-6, 4.999995e-6, 5.000005e-6, 0.999995, 0.9999949999, 9.999995, 99999.999995,
                     -123456.789015, 2.0 ** 43, -(2.0 ** 43), 2.0 ** 42 + 0.5, 1e15, -1e22, 1e300, 0.1, 0.15, 0.25, 0.35, 1e-300])
    rnd = np.concatenate([-rng.uniform(0, 1e5, 100000), rng.normal(0, 1, 100000), rng.uniform(-1, 1, 50000) * 10.0 ** rng.integers(-12, 13, 50000),
                          np.round(rng.uniform(-1e3, 1e3, 50000), 5), np.round(rng.uniform(-10, 10, 50000), 5) + 5e-6])
    vals = np.concatenate([ties, -ties, ties * 1000 + 7, edge, rnd])
    B = len(vals)
    z = rng.integers(0, 2 ** 31 - 1, B).astype(np.int32)
    fa = eng.FinalArgs([f"BC{i:07d}" for i in range(B)], ["S0"], (0.0, 0.5), 0.5, z, z, z, z)
    want = ["BARCODE\tSM_ID\tRD.TOTL\tRD.PASS\tRD.UNIQ\tN.SNP\tLLK1\tLLK0\tPOSTPRB\n"]
    for i in range(B):
        want.append("BC%07d\tS0\t%d\t%d\t%d\t%d\t%.5f\t%.5f\t1\n" % (i, z[i], z[i], z[i], z[i], vals[i], -vals[i]))
    want = "".join(want)
    for threads in ("1", "5"):
        monkeypatch.setenv("DMX_THREADS", threads)
        path = str(tmp_path / f"fmt{threads}.single")
        eng.write_single(fa, vals.reshape(B, 1), -vals, path)
        got = open(path).read()
        assert got == want, next((a, b) for a, b in zip(got.split("\n"), want.split("\n")) if a != b)


def test_geno_transforms_hand_vectors(eng, oracle):
    """Row a3 is unpinned by the reference (parse_posteriors needs htslib), so beyond product == oracle it is pinned on values
    derived by hand from bcf_filtered_reader.cpp:186-242 (allele counts), :381-400 (GT), :255-311 (PL, 10 EM rounds) and
    :421-448 (GP) — exact rationals or 50-digit arithmetic where an EM is involved, compared after rounding to float32."""
    import mpmath
    f32 = np.float32
    MISS = np.iinfo(np.int32).min                      # bcf_int32_missing; toProb(uint32) sends it to phred2Prob[255] (PhredHelper.h:40)
    # ---- GT (:381-400).  called genotype g: 1-e on g, e/2 elsewhere, as float32
    assert np.array_equal(eng.geno_from_gt(np.array([[1, 1]]), 0.0), [[0.0, 0.0, 1.0]])
    assert np.array_equal(eng.geno_from_gt(np.array([[0, 0]]), 0.1), np.array([[f32(0.9), f32(0.05), f32(0.05)]]))
    assert np.array_equal(eng.geno_from_gt(np.array([[1, 0]]), 0.01), np.array([[f32(0.005), f32(0.99), f32(0.005)]]))   # 1|0 is a het too
    # a column in which EVERY genotype is missing: an = 0, ac = (0, 0) -> (0 + 1/2)/(0 + 1) per allele -> HWE (1/4, 1/2, 1/4)
    g = eng.geno_from_gt(np.array([[-1, -1], [-1, -1], [-1, -1]]), 0.01)
    assert np.array_equal(g, np.tile(np.array([[0.25, 0.5, 0.25]], dtype=f32), (3, 1)))
    # half-missing (0/.) counts its called allele (an = 1, ac = (1, 0)) and is itself "missing" (bcf_filtered_reader.h:144-149):
    #   l0 = (1.5/2)^2, l1 = 2 (0.5/2)(1.5/2), l2 = (0.5/2)^2, evaluated left to right in binary64
    g = eng.geno_from_gt(np.array([[0, -1]]), 0.01)
    want = [f32(1.0 * 1.5 / 2.0 * 1.5 / 2.0), f32(2.0 * 0.5 / 2.0 * 1.5 / 2.0), f32(1.0 * 0.5 / 2.0 * 0.5 / 2.0)]
    assert np.array_equal(g[0], want) and np.array_equal(g[0], np.array([0.5625, 0.375, 0.0625], dtype=f32))
    # 0/0, 0/0, 0/1, ./. : an = 6, ac = (5, 1) -> missing row ((5.5/7)^2, 2 (1.5/7)(5.5/7), (1.5/7)^2)
    g = eng.geno_from_gt(np.array([[0, 0], [0, 0], [0, 1], [-1, -1]]), 0.01)
    want = np.array([f32(1.0 * 5.5 / 7.0 * 5.5 / 7.0), f32(2.0 * 1.5 / 7.0 * 5.5 / 7.0), f32(1.0 * 1.5 / 7.0 * 1.5 / 7.0)])
    assert np.array_equal(g[3], want) and abs(float(g[3].sum()) - 1.0) < 1e-6
    assert np.array_equal(g[2], np.array([f32(0.005), f32(0.99), f32(0.005)]))
    # 0/0, 0/1, 1/1, ./. : an = 6, ac = (3, 3) -> (3.5/7)^2 = 1/4 etc.: exact
    g = eng.geno_from_gt(np.array([[0, 0], [0, 1], [1, 1], [-1, -1]]), 0.01)
    assert np.array_equal(g[3], np.array([0.25, 0.5, 0.25], dtype=f32))

    # ---- PL (:255-311): 10 EM rounds from af = (1/2, 1/2); the LAST round's posteriors are kept
    def em_pl(pl_rows):                               # 50-digit arithmetic, no float rounding anywhere
        mpmath.mp.dps = 50
        L = [[mpmath.mpf(10) ** (-mpmath.mpf(min(p, 255) if p >= 0 else 255) / 10) for p in row] for row in pl_rows]
        af = [mpmath.mpf(1) / 2, mpmath.mpf(1) / 2]
        out = None
        for _ in range(10):
            new, out = [mpmath.mpf(0), mpmath.mpf(0)], []
            for Ls in L:
                gp = [af[0] * af[0] * Ls[0], 2 * af[1] * af[0] * Ls[1], af[1] * af[1] * Ls[2]]
                s = sum(gp)
                gp = [x / s for x in gp]
                new[0] += 2 * gp[0] + gp[1]
                new[1] += gp[1] + 2 * gp[2]
                out.append(gp)
            af = [x / (2 * len(L)) for x in new]
        return np.array([[float(x) for x in r] for r in out])
    # every PL missing: the three likelihoods are equal, af stays (1/2, 1/2): (1/4, 1/2, 1/4) exactly
    assert np.array_equal(eng.geno_from_pl(np.array([[MISS, MISS, MISS]], dtype=np.int32)), np.array([[0.25, 0.5, 0.25]], dtype=f32))
    # PL above 255 is read as 255
    assert np.array_equal(eng.geno_from_pl(np.array([[0, 300, 1000]], dtype=np.int32)), eng.geno_from_pl(np.array([[0, 255, 255]], dtype=np.int32)))
    for rows in ([[0, 30, 60], [60, 30, 0]], [[0, 255, 255]], [[232, 14, 0], [MISS, MISS, MISS]], [[10, 0, 10], [0, 3, 30], [40, 0, 25], [MISS, 5, 0]]):
        got = eng.geno_from_pl(np.array(rows, dtype=np.int32))
        want = em_pl(rows)
        assert np.allclose(got, want.astype(f32), rtol=3e-7, atol=1e-30), (rows, got, want)
    # --geno-error does not touch the PL path (cmd_cram_demuxlet.cpp passes it, parse_likelihoods ignores it): one entry point, no error argument

    # ---- GP (:421-448): per-sample float32 normalisation, pseudo-sample HWE(1/2), mean over (n + 1) with an INTEGER divisor, mix
    gp = np.array([[0.2, 0.3, 0.5]], dtype=f32)
    s = f32(f32(f32(0) + gp[0, 0]) + gp[0, 1]) + gp[0, 2]
    norm = gp[0] / s                                   # float32 divisions
    assert np.array_equal(eng.geno_from_gp(gp, 0.0), [norm])                      # gt_error = 0: (1.0) * gp + 0.0 * mean = gp
    two = np.array([[1, 0, 0], [0, 0, 1]], dtype=f32)
    mean = np.array([f32(f32(0.25) + f32(1)) / f32(3), f32(0.5) / f32(3), f32(f32(0.25) + f32(1)) / f32(3)], dtype=f32)   # float / (int32_t)(2 + 1.0)
    want = np.array([[f32((1.0 - 0.1) * 1.0 + 0.1 * float(mean[0])), f32(0.1 * float(mean[1])), f32(0.1 * float(mean[2]))],
                     [f32(0.1 * float(mean[0])), f32(0.1 * float(mean[1])), f32((1.0 - 0.1) * 1.0 + 0.1 * float(mean[2]))]])
    assert np.array_equal(eng.geno_from_gp(two, 0.1), want)
    # unnormalised input (2, 2, 4) == (0.25, 0.25, 0.5)
    assert np.array_equal(eng.geno_from_gp(np.array([[2, 2, 4]], dtype=f32), 0.01), eng.geno_from_gp(np.array([[0.25, 0.25, 0.5]], dtype=f32), 0.01))

    # ---- the tutorial VCF's first records (tutorial/README.MD; FORMAT GT:GQ:DP:PL:AD, samples jurkat / 293T_RTG), by hand:
    #   1:700513  0/1:3:14:232,14,0   .:.:.:.:.      1:713914  0/1:2:4:54,3,0   .     1:761606  0/1:2:18:261,2,0   .
    import gzip
    from golden_util import GOLDEN
    recs = [l.rstrip("\n").split("\t") for l in gzip.open(GOLDEN / "tutorial_jurkat_293T_first4000.vcf.gz", "rt") if not l.startswith("#")][:4]
    assert [r[1] for r in recs] == ["700513", "713914", "714061", "761606"] and recs[0][9].startswith("0/1:3:14:232,14,0") and recs[0][10].startswith(".")
    for r, pl in ((recs[0], [232, 14, 0]), (recs[1], [54, 3, 0]), (recs[3], [261, 2, 0])):
        assert r[9].split(":")[3] == ",".join(map(str, pl)) and r[9].split(":")[0] == "0/1"
        # GT: jurkat het, 293T missing: an = 2, ac = (1, 1) -> 293T gets ((1.5/3)^2, 2 (1.5/3)^2, (1.5/3)^2) = (1/4, 1/2, 1/4)
        g = eng.geno_from_gt(np.array([[0, 1], [-1, -1]]), 0.01)
        assert np.array_equal(g, np.array([[f32(0.005), f32(0.99), f32(0.005)], [0.25, 0.5, 0.25]], dtype=f32))
        # PL: jurkat's three likelihoods against a sample whose PL is missing
        got = eng.geno_from_pl(np.array([pl, [MISS, MISS, MISS]], dtype=np.int32))
        assert np.allclose(got, em_pl([pl, [MISS, MISS, MISS]]).astype(f32), rtol=3e-7, atol=1e-30)
        assert got[0, 2] > got[0, 1] > got[0, 0]       # PL ... ,0: hom-alt is the most likely genotype of jurkat here
    # the oracle restatement agrees on all of the above (it is what the GPU tests check the product against)
    for a, e in ((np.array([[0, -1]]), 0.01), (np.array([[0, 0], [0, 0], [0, 1], [-1, -1]]), 0.01)):
        assert np.array_equal(eng.geno_from_gt(a, e), oracle.geno_from_gt(a, e))


def test_resolve_tie_order_asks_the_host_libm(eng):
    """dmx_resolve_tie_order (K3b's RESOLVABLE records, DESIGN.md "Ties"): the accumulator is the lower candidate when the host
    libm's log() of the recorded argument equals the recorded lower log value, the upper one when it is the next double, and
    the record stays unresolved otherwise.  The order then follows the reference's strict-< scan (:799-814)."""
    import math
    from demuxlet_amd import capi
    x = 0.37251
    L = math.log(x)
    up = np.nextafter(L, np.inf)
    dn = np.nextafter(L, -np.inf)
    s = np.zeros(4, dtype=capi.SUMMARY_DTYPE)
    s["n_pairs"] = 10; s["n_best"] = 1
    s["j_best"] = 3; s["k_best"] = 7
    s["llk1"] = -1.0; s["llk2"] = -2.0; s["llk10"] = -10.0; s["llk20"] = -20.0
    s["flags"] = capi.DMX_CELL_ORDER_RESOLVABLE
    # 0: (a,b) open, libm returns the lower candidate -> llk_ab; (b,a) certain and smaller -> order stays (3, 7)
    s[0]["llk_ab"], s[0]["llk_ab_alt"], s[0]["ev_x_ab"], s[0]["ev_t_ab"] = -100.0, -99.5, x, L
    s[0]["llk_ba"] = s[0]["llk_ba_alt"] = -100.25
    # 1: (a,b) open, libm returns the upper candidate -> llk_ab_alt = -100.5 < (b,a) = -100.25 -> (b,a) wins: order (7, 3)
    s[1]["llk_ab"], s[1]["llk_ab_alt"], s[1]["ev_x_ab"], s[1]["ev_t_ab"] = -101.0, -100.5, x, dn
    s[1]["llk_ba"] = s[1]["llk_ba_alt"] = -100.25
    # 2: both open
    s[2]["llk_ab"], s[2]["llk_ab_alt"], s[2]["ev_x_ab"], s[2]["ev_t_ab"] = -50.0, -49.0, x, L
    s[2]["llk_ba"], s[2]["llk_ba_alt"], s[2]["ev_x_ba"], s[2]["ev_t_ba"] = -50.0, -49.0, x, dn
    # 3: the libm's answer is neither candidate -> stays unresolved
    s[3]["llk_ab"], s[3]["llk_ab_alt"], s[3]["ev_x_ab"], s[3]["ev_t_ab"] = -5.0, -4.0, x, up
    s[3]["llk_ba"] = s[3]["llk_ba_alt"] = -6.0
    left = eng.resolve_tie_order(s)
    assert left == 1
    C, R = capi.DMX_CELL_ORDER_CERTIFIED, capi.DMX_CELL_ORDER_RESOLVABLE
    assert [int(f) for f in s["flags"]] == [C, C, C, R]
    assert (s[0]["j_best"], s[0]["k_best"], s[0]["llk12"], s[0]["llk1"], s[0]["llk10"]) == (3, 7, -100.0, -1.0, -10.0)
    assert (s[1]["j_best"], s[1]["k_best"], s[1]["llk12"], s[1]["llk1"], s[1]["llk2"], s[1]["llk10"], s[1]["llk20"]) == (7, 3, -100.25, -2.0, -1.0, -20.0, -10.0)
    assert (s[2]["llk_ab"], s[2]["llk_ba"]) == (-50.0, -49.0) and (s[2]["j_best"], s[2]["k_best"], s[2]["llk12"]) == (7, 3, -49.0)
    assert (s[3]["j_best"], s[3]["k_best"]) == (3, 7)
