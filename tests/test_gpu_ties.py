"""GPU (-m gpu): the tie-heavy regime — panels with genotype-IDENTICAL samples and barcodes that cover a handful of
SNPs, the normal state of a low-coverage droplet.  There the reference's scans (cmd_cram_demuxlet.cpp:746-758 top-2 singlets, :799-814
best doublet over (j, k, alpha) with a strict `<`, :837-857 the SNG/DBL/AMB rule) meet many EXACT ties (duplicate samples give
bit-identical accumulators: the first in scan order wins) and many last-bit near-ties (the alphas of a doublet of two identical samples are
one number mathematically and differ in the last bit of libm's log() terms).  The device cannot promise libm's last bit, so K3 flags such
barcodes (DMX_CELL_NEAR_DOUBLET / _NEAR_SINGLET) and the host arbiter decides them with the host libm.  What is pinned here: whatever
the flags and certificates do, the FILES are the oracle's byte for byte — through dmx_demuxlet_run (1 and 3 engines, STRICT and FAST) and
through the gathered-records path (get_sing + K3 records + dmx_write_doublet_summary, what rank 0 of a multi-GPU job writes from)."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def eng():
    from demuxlet_amd import build, capi, engine
    build.build()
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    capi.load()
    return engine


def oracle_files(oracle, pl, g, alphas, barcodes, sample_ids, prefix, write_pair=False, prior=0.5):
    """The oracle's four files for a C-ABI pileup (words rebuilt from the packed read bytes); returns its raw arrays too."""
    words = ((pl.reads >> 7).astype(np.uint32) << 24) | ((pl.reads & 0x7F).astype(np.uint32) << 16) | 1
    pair_snp = pl.pair_snp if pl.pair_snp is not None else np.tile(np.arange(pl.n_snps, dtype=np.int32), pl.n_cells)
    csr = oracle.Csr(list(barcodes), pl.cell_pair_off, pair_snp, np.concatenate([[0], np.cumsum(pl.pair_nrd.astype(np.int64))]),
                     words.astype(np.uint32), pl.rd_totl, pl.rd_pass, pl.rd_uniq)
    return oracle.run_csr(csr, list(sample_ids), g, oracle.Params(tuple(alphas), prior, 0, 0, 0, write_pair), str(prefix))


def assert_same_file(got: Path, want: Path, what):
    a, b = got.read_bytes(), want.read_bytes()
    if a == b:
        return
    la, lb = a.decode().splitlines(), b.decode().splitlines()
    assert len(la) == len(lb), (what, got.name, len(la), len(lb))
    bad = [(i, x, y) for i, (x, y) in enumerate(zip(la, lb)) if x != y]
    raise AssertionError(f"{what}: {got.name} differs from the oracle's in {len(bad)} of {len(la)} rows; first:\n  got  {bad[0][1]}\n  want {bad[0][2]}")


def tie_problem(eng, seed, V, field, B, S=60, groups=((0, 1), (2, 3, 4)), delta=0.15):
    """S SNPs x V samples in which the samples of every group carry IDENTICAL genotype rows (same raw field values, so the float32 rows are
    bit-identical), and B barcodes covering about S*delta SNPs each (3-20 for the defaults), a third of them doublets."""
    from demuxlet_amd import synth
    rng = np.random.default_rng(seed)
    raw = synth.make_raw_genotypes(rng, S, V)
    al = raw.alleles.copy()
    for grp in groups:
        for j in grp[1:]:
            if j < V:
                al[:, j] = al[:, grp[0]]
    if field == "GT":
        g = np.stack([eng.geno_from_gt(al[s], 0.01) for s in range(S)])
    else:
        gp = synth.raw_gp_from_alleles(rng, al)
        for grp in groups:
            for j in grp[1:]:
                if j < V:
                    gp[:, j] = gp[:, grp[0]]
        g = np.stack([eng.geno_from_gp(gp[s], 0.01) for s in range(S)])
    for grp in groups:
        for j in grp[1:]:
            if j < V:
                assert np.array_equal(g[:, j], g[:, grp[0]])
    sp = synth.make_pileup(rng, al, B, delta, 1.3, dense_layout=False, doublet_rate=0.35)
    pl = eng.HostPileup(sp.n_cells, sp.n_snps, sp.cell_pair_off, sp.cell_read_off, sp.pair_snp, sp.pair_nrd, sp.reads, sp.rd_totl, sp.rd_pass, sp.rd_uniq)
    return g, pl


def run_both_paths(eng, oracle, tmp_path, g, pl, alphas, mode, what, min_fetched_frac=None):
    """dmx_demuxlet_run (1 and 3 engines, arbiter on, with and without --write-pair) and the records path against the oracle's files."""
    from demuxlet_amd import capi
    V, B = g.shape[1], pl.n_cells
    bcs = [f"BC{(i * 7919) % 100003:06d}-1" for i in range(B)]          # not in sorted order: the writers sort (std::map order)
    sms = [f"SM{j:02d}" for j in range(V)]
    md = capi.DMX_MODE_FAST if mode == "fast" else capi.DMX_MODE_STRICT
    ref = oracle_files(oracle, pl, g, alphas, bcs, sms, tmp_path / "ref", write_pair=True)
    fetched = None
    for n_gpus, wp in ((1, False), (3, False), (1, True)):
        pre = tmp_path / f"run{n_gpus}{int(wp)}"
        tm = eng.demuxlet_run(pl, g, sms, alphas, str(pre), write_pair=wp, arbiter=True, n_gpus=n_gpus, mode=md, barcodes=bcs, timing=True)
        for suf in ("single", "sing2", "best") + (("pair",) if wp else ()):
            assert_same_file(Path(f"{pre}.{suf}"), tmp_path / f"ref.{suf}", f"{what}, dmx_demuxlet_run, {n_gpus} engine(s), write_pair={wp}")
        if not wp and n_gpus == 1:
            fetched = tm["n_cells_grid_fetched"]
    # the gathered-records path: nothing but (llks, llk0s, sing, llks00, K3 record) per barcode leaves the engine
    e = eng.Engine(V, alphas, 0.5, mode=md)
    e.set_genotypes(g); e.set_pileup(pl)
    e.run()
    llks, llk0s = e.get_singlet()
    _, l00, summ = e.get_doublet(want_grid=False)
    sing = e.get_sing()
    near_ids = eng.near_tie_cells(summ)
    near_grids = e.get_cell_grids(near_ids)          # what a rank adds to the gather for its flagged barcodes (demuxlet_amd/dist.py)
    e.close()
    fa = eng.FinalArgs(bcs, sms, alphas, 0.5, pl.rd_totl, pl.rd_pass, pl.rd_uniq, pl.n_snp_per_cell)
    eng.write_single(fa, llks, llk0s, str(tmp_path / "rec.single"))
    assert_same_file(tmp_path / "rec.single", tmp_path / "ref.single", f"{what}, records path")
    # (a) records + the flagged barcodes' device grids + the arbiter: what rank 0 of a multi-GPU job does
    eng.write_doublet_summary(fa, sing, l00, summ, str(tmp_path / "rec"), tie_pileup=pl, tie_g=g,
                              cell_grids={int(c): gr for c, gr in zip(near_ids, near_grids)})
    # (b) records + the arbiter only: the flagged barcodes' grids are re-evaluated on the host
    eng.write_doublet_summary(fa, sing, l00, summ, str(tmp_path / "rech"), tie_pileup=pl, tie_g=g)
    for pre in ("rec", "rech"):
        for suf in ("sing2", "best"):
            assert_same_file(tmp_path / f"{pre}.{suf}", tmp_path / f"ref.{suf}", f"{what}, records path ({'device grids' if pre == 'rec' else 'host grids'})")
    covered = int((summ["n_pairs"] > 0).sum())
    # DMX_CELL_NEAR_RULE (round 5): K3 flags exactly the barcodes with a comparison of the BEST rule (:837,:844) within 1e-7 of flipping
    cov = summ["n_pairs"] > 0
    s1, s2, l12, l1, l2 = (summ[k].astype(np.float64) for k in ("sing_llk1", "sing_llk2", "llk12", "llk1", "llk2"))
    with np.errstate(invalid="ignore"):
        want_rule = (np.abs(l12 - (s1 + 2)) < 1e-7) | (np.abs(s1 - (s2 + 2)) < 1e-7) | (((np.abs(l12 - l1) < 1e-7) | (np.abs(l12 - l2) < 1e-7)) & (l12 > s1 + 2 - 1e-7))
    has_rule = (summ["flags"] & capi.DMX_CELL_NEAR_RULE) != 0
    certified = (summ["flags"] & capi.DMX_CELL_ORDER_CERTIFIED) != 0   # (K3b rewrites llk12 with the reference's bits: 1e-11 beside a 1e-7 window)
    assert np.array_equal(has_rule[cov & ~certified], want_rule[cov & ~certified]), (int(has_rule.sum()), int(want_rule.sum()))
    near = int(((summ["flags"] & (capi.DMX_CELL_NEAR_DOUBLET | capi.DMX_CELL_NEAR_SINGLET)) != 0).sum())
    print(f"{what}: {int(has_rule[cov].sum())} of {covered} barcodes flagged DMX_CELL_NEAR_RULE")
    print(f"{what}: {covered} covered barcodes, {near} flagged near-tie by K3, grid fetched for {fetched} "
          f"({100.0 * fetched / max(covered, 1):.1f} %)")
    # the flags are what routes a barcode to the arbiter: every K3-flagged barcode had its grid fetched, and nothing but flagged
    # (or uncertified-order) barcodes did; a regression in the flags shows here as a jump to 0 or to `covered`
    assert fetched >= near and fetched <= covered
    if min_fetched_frac is not None:
        assert fetched >= min_fetched_frac * covered, (fetched, covered)
    return ref, summ, fetched, covered


def test_fuzz_case_9334_368(eng, oracle, tmp_path):
    """The case round 3's STRICT sweep stopped at (profiles/r03_fuzz_summary.txt): V = 33, A = 5 (0, .22, .38, .41, .5), GT, 7 SNPs, 16
    barcodes; two samples with the same genotypes, every alpha of their doublet equal to the last bit.  K3 named alpha index 4 where the
    oracle's libm-based grid names 2 and flagged the barcode: the arbiter owns the decision and the files must be the oracle's."""
    from demuxlet_amd import capi
    z = np.load(ROOT / "tests" / "golden" / "fuzz_9334_case368.npz")
    g, alphas = z["g"], tuple(float(a) for a in z["alphas"])
    pl = eng.HostPileup(int(z["n_cells"]), int(z["n_snps"]), z["cell_pair_off"], z["cell_read_off"], None if bool(z["dense"]) else z["pair_snp"],
                        z["pair_nrd"], z["reads"], z["rd_totl"], z["rd_pass"], z["rd_uniq"])
    assert g.shape == (7, 33, 3) and len(alphas) == 5 and pl.n_cells == 16
    ref, summ, fetched, covered = run_both_paths(eng, oracle, tmp_path, g, pl, alphas, "strict", "fuzz 9334/368")
    # the barcode(s) whose device choice of alpha differs from the oracle's are flagged (that is the contract the arbiter relies on)
    from golden_util import summary_from_grid
    n_named_differently = 0
    for c in np.flatnonzero(ref.processed.astype(bool)):
        want = summary_from_grid(ref.llksAB[c], ref.llks00[c], alphas, 0.5, int(summ[c]["n_pairs"]), summ.dtype)
        if int(summ[c]["n_best"]) != int(want["n_best"]) or {int(summ[c]["j_best"]), int(summ[c]["k_best"])} != {int(want["j_best"]), int(want["k_best"])}:
            n_named_differently += 1
            assert summ[c]["flags"] & capi.DMX_CELL_NEAR_DOUBLET, c
    print(f"fuzz 9334/368: {n_named_differently} barcode(s) where K3's own choice differs from the oracle's; all flagged")


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("V,field,alphas", [(8, "GT", (0.0, 0.5)), (8, "GP", (0.0, 0.5)), (33, "GT", (0.0, 0.5)), (33, "GP", (0.0, 0.1, 0.2, 0.35, 0.5)),
                                            (64, "GT", (0.0, 0.1, 0.2, 0.35, 0.5)), (64, "GP", (0.0, 0.5)), (8, "GT", (0.0, 0.1, 0.2, 0.35, 0.5))])
def test_identical_samples_and_few_snps(eng, oracle, tmp_path, V, field, alphas, mode):
    """Genotype-identical pairs and triples in the panel, 200 barcodes covering 3-20 SNPs each: all files byte-identical to the oracle's."""
    g, pl = tie_problem(eng, 7000 + V + len(alphas), V, field, 200)
    n = np.diff(pl.cell_pair_off)
    assert 3 <= np.percentile(n, 10) and np.percentile(n, 90) <= 20
    run_both_paths(eng, oracle, tmp_path, g, pl, alphas, mode, f"V={V} {field} A={len(alphas)} {mode}", min_fetched_frac=0.05)


def test_every_sample_duplicated(eng, oracle, tmp_path):
    """The degenerate panel: samples 2i and 2i+1 identical for every i (V = 16) and a panel of four copies of two genotypes (V = 8);
    every decision of every barcode is an exact tie with at least one other candidate."""
    for V, groups in ((16, tuple((2 * i, 2 * i + 1) for i in range(8))), (8, ((0, 2, 4, 6), (1, 3, 5, 7)))):
        for field in ("GT", "GP"):
            g, pl = tie_problem(eng, 7100 + V, V, field, 120, S=40, groups=groups, delta=0.2)
            d = tmp_path / f"{V}{field}"
            d.mkdir()
            run_both_paths(eng, oracle, d, g, pl, (0.0, 0.5), "strict", f"all-duplicated V={V} {field}", min_fetched_frac=0.5)


def _dist_worker(rank, world, port, seed, V, field, outdir):
    import os
    import sys
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    from demuxlet_amd import capi, engine
    from demuxlet_amd import dist as ddist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    g, pl = tie_problem(engine, seed, V, field, 200)
    bcs = [f"BC{(i * 7919) % 100003:06d}-1" for i in range(pl.n_cells)]
    sms = [f"SM{j:02d}" for j in range(V)]
    res = ddist.run_sharded(pl, bcs, V, 2, ddist.engine_compute(g, (0.0, 0.5), device=rank), capi.SUMMARY_DTYPE, device=torch.device("cuda", rank))
    if rank == 0:
        order, rec = res
        assert len(rec.near_cells) > 10
        ddist.write_from_records(order, rec, pl, bcs, sms, (0.0, 0.5), os.path.join(outdir, "o"), g=g)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("V,field", [(8, "GT"), (33, "GP")])
def test_sharded_run_over_rccl_with_the_engine(eng, oracle, tmp_path, V, field):
    """demuxlet_amd/dist.py end to end on the GPU(s) of this box: one process per GPU (all there are, at most 8; on a 1-GPU box a 1-rank RCCL
    group), the real engine per rank, ONE gather of the records with the near-tie-flagged barcodes' grids riding along, rank 0 writes
    — the tie-heavy problem's files are the oracle's byte for byte (cmd_cram_demuxlet.cpp:576: barcodes are independent)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    world = max(1, min(8, torch.cuda.device_count()))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    seed = 7300 + V
    mp.spawn(_dist_worker, args=(world, port, seed, V, field, str(tmp_path)), nprocs=world, join=True)
    g, pl = tie_problem(eng, seed, V, field, 200)
    bcs = [f"BC{(i * 7919) % 100003:06d}-1" for i in range(pl.n_cells)]
    oracle_files(oracle, pl, g, (0.0, 0.5), bcs, [f"SM{j:02d}" for j in range(V)], tmp_path / "ref")
    for suf in ("single", "sing2", "best"):
        assert_same_file(tmp_path / f"o.{suf}", tmp_path / f"ref.{suf}", f"dist.py over RCCL, world {world}, V={V} {field}")
