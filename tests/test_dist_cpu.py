"""CPU: the multi-GPU plumbing (demuxlet_amd/dist.py) — sharding of the sorted barcodes, the single gather of per-cell
records (gloo, world_size 2), and the finaliser that works from those records — against the reference's outputs.
The per-shard numbers come from the oracle here (the engine has no CPU path); on GPUs the same code runs the engine."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from golden_util import CASES, Golden, summary_from_grid

ROOT = Path(__file__).resolve().parents[1]


def test_balanced_ranges_properties():
    from demuxlet_amd import dist
    rng = np.random.default_rng(0)
    for n, w in ((0, 2), (1, 2), (5, 8), (1000, 8), (37, 3)):
        cost = rng.integers(1, 1000, size=n).astype(float)
        r = dist.balanced_ranges(cost, w)
        assert len(r) == w and r[0][0] == 0 and r[-1][1] == n
        assert all(r[i][1] == r[i + 1][0] for i in range(w - 1)) and all(a <= b for a, b in r)
        if n >= 8 * w:
            sums = [cost[a:b].sum() for a, b in r]
            assert max(sums) <= cost.sum() / w + cost.max()
    order = dist.sorted_barcode_order(["b", "a", "B", "ab", "a-1"])
    assert [["b", "a", "B", "ab", "a-1"][i] for i in order] == ["B", "a", "a-1", "ab", "b"]


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


@pytest.mark.parametrize("name", [c for c in CASES if c not in ("gt_v64_a2",)])
def test_summary_writer_reproduces_reference_files(oracle, name, tmp_path):
    """.sing2/.best from per-cell records (what rank 0 holds after the gather) == the reference's files."""
    from demuxlet_amd import build, capi, engine
    build.build()
    gd = Golden(name)
    cnt = gd.z["ref_counters"]
    B = len(gd.ref_barcodes)
    grid, l00 = gd.z["ref_llksAB"], gd.z["ref_llks00"]
    summ = np.zeros(B, dtype=capi.SUMMARY_DTYPE)
    for c in range(B):
        if gd.z["ref_processed"][c]:
            summ[c] = summary_from_grid(grid[c], l00[c], gd.alphas, gd.doublet_prior, cnt[c, 3], capi.SUMMARY_DTYPE)
    fa = engine.FinalArgs(gd.ref_barcodes, gd.sample_ids, gd.alphas, gd.doublet_prior, cnt[:, 0], cnt[:, 1], cnt[:, 2], cnt[:, 3],
                          gd.min_total, gd.min_uniq, gd.min_snp, False)
    engine.write_doublet_summary(fa, grid[:, :, 0, 0], l00, summ, str(tmp_path / "o"))
    assert (tmp_path / "o.best").read_bytes() == gd.files["best"]
    assert (tmp_path / "o.sing2").read_bytes() == gd.files["sing2"]


@pytest.mark.parametrize("name", ["gt_v4_a2_pair", "pl_v12_a6_pair", "gt_v3_alpha_quirk", "gp_v8_a2_minsnp"])
def test_summary_writer_near_tie_records(oracle, name, tmp_path):
    """A record flagged DMX_CELL_NEAR_* is not decided from the record: (a) with the barcode's grid in
    dmx_final_input.cell_grid the reference's scans run on it; (b) without a grid but with the tie pileup the barcode's whole grid is
    re-evaluated on the host in the reference's operation order.  Either way a record whose own decisions are WRONG (as a device
    that lost a last-bit tie to libm would produce) still gives the reference's files, byte for byte."""
    from demuxlet_amd import build, capi, engine
    build.build()
    gd = Golden(name)
    st = build_store(engine, gd.problem(oracle))
    pl = st.freeze()
    assert st.barcodes() == gd.ref_barcodes
    cnt = gd.z["ref_counters"]
    B, V = len(gd.ref_barcodes), len(gd.sample_ids)
    grid, l00 = gd.z["ref_llksAB"], gd.z["ref_llks00"]
    summ = np.zeros(B, dtype=capi.SUMMARY_DTYPE)
    for c in range(B):
        if gd.z["ref_processed"][c]:
            summ[c] = summary_from_grid(grid[c], l00[c], gd.alphas, gd.doublet_prior, cnt[c, 3], capi.SUMMARY_DTYPE)
    covered = np.flatnonzero(summ["n_pairs"] > 0)
    bad = summ.copy()
    for c in covered:                                   # sabotage every decision of the record and flag it
        bad[c]["flags"] = capi.DMX_CELL_NEAR_DOUBLET | capi.DMX_CELL_NEAR_SINGLET
        bad[c]["j_best"], bad[c]["k_best"] = (int(summ[c]["k_best"]) + 1) % V, int(summ[c]["j_best"])
        bad[c]["n_best"] = 1 + int(summ[c]["n_best"]) % (len(gd.alphas) - 1)
        bad[c]["i_sing1"], bad[c]["i_sing2"] = int(summ[c]["i_sing2"]), int(summ[c]["i_sing1"])
        bad[c]["llk12"] += 0.25; bad[c]["max_llk"] += 0.125; bad[c]["sum_double"] *= 1.5
    fa = engine.FinalArgs(gd.ref_barcodes, gd.sample_ids, gd.alphas, gd.doublet_prior, cnt[:, 0], cnt[:, 1], cnt[:, 2], cnt[:, 3],
                          gd.min_total, gd.min_uniq, gd.min_snp, False)
    sing = np.ascontiguousarray(grid[:, :, 0, 0]) + 0.5         # the singlet column of the records is not to be trusted either
    assert list(engine.near_tie_cells(bad)) == list(covered)
    # (a) grids handed over (here the reference's own; on a GPU box Engine.get_cell_grids)
    engine.write_doublet_summary(fa, sing, l00, bad, str(tmp_path / "a"), cell_grids={int(c): grid[c] for c in covered})
    # (b) no grids: the host re-evaluates the flagged barcodes from the pileup
    engine.write_doublet_summary(fa, sing, l00, bad, str(tmp_path / "b"), tie_pileup=pl, tie_g=gd.g)
    for pre in ("a", "b"):
        assert (tmp_path / f"{pre}.best").read_bytes() == gd.files["best"], pre
        assert (tmp_path / f"{pre}.sing2").read_bytes() == gd.files["sing2"], pre
    # and with neither, the sabotaged record is what gets printed (nothing silently recomputes)
    engine.write_doublet_summary(fa, sing, l00, bad, str(tmp_path / "c"))
    assert (tmp_path / "c.best").read_bytes() != gd.files["best"]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, name, outdir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    from demuxlet_amd import capi, engine
    from demuxlet_amd import dist as ddist
    from golden_util import Golden, summary_from_grid
    from oracle import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gd = Golden(name)
    pb = gd.problem(O)
    st = build_store(engine, pb)
    pl = st.freeze()
    barcodes = st.barcodes()
    V, A = len(gd.sample_ids), len(gd.alphas)

    def compute(shard):          # stands in for the GPU engine: same inputs (a dmx_pileup slice), same record layout
        words = ((shard.reads >> 7).astype(np.uint32) << 24) | ((shard.reads & 0x7F).astype(np.uint32) << 16) | 1
        csr = O.Csr([f"x{i}" for i in range(shard.n_cells)], shard.cell_pair_off, shard.pair_snp,
                    np.concatenate([[0], np.cumsum(shard.pair_nrd.astype(np.int64))]), words, shard.rd_totl, shard.rd_pass, shard.rd_uniq)
        r = O.run_csr(csr, gd.sample_ids, gd.g, O.Params(gd.alphas, gd.doublet_prior))
        summ = np.zeros(shard.n_cells, dtype=capi.SUMMARY_DTYPE)
        for c in range(shard.n_cells):
            if shard.n_snp_per_cell[c] > 0:
                summ[c] = summary_from_grid(r.llksAB[c], r.llks00[c], gd.alphas, gd.doublet_prior, shard.n_snp_per_cell[c], capi.SUMMARY_DTYPE)
        # every third covered cell plays a near-tie: its record is flagged and sabotaged, its grid rides along in the gather
        near = np.flatnonzero(summ["n_pairs"] > 0)[::3].astype(np.int32)
        for c in near:
            summ[c]["flags"] |= capi.DMX_CELL_NEAR_DOUBLET
            summ[c]["j_best"], summ[c]["k_best"] = int(summ[c]["k_best"]), (int(summ[c]["j_best"]) + 1) % V
            summ[c]["llk12"] -= 3.0
        return ddist.CellRecords(r.llks, r.llk0s, np.ascontiguousarray(r.llksAB[:, :, 0, 0]), r.llks00, summ, near, r.llksAB[near])

    res = ddist.run_sharded(pl, barcodes, V, A, compute, capi.SUMMARY_DTYPE)
    if rank == 0:
        order, rec = res
        inv = np.empty_like(order); inv[order] = np.arange(len(order))      # records are in sorted-barcode order
        fa = engine.FinalArgs(barcodes, gd.sample_ids, gd.alphas, gd.doublet_prior, pl.rd_totl, pl.rd_pass, pl.rd_uniq,
                              pl.n_snp_per_cell, gd.min_total, gd.min_uniq, gd.min_snp, False)
        engine.write_single(fa, rec.llks[inv], rec.llk0s[inv], os.path.join(outdir, "o.single"))
        assert len(rec.near_cells) > 0 and rec.near_grids.shape == (len(rec.near_cells), V, V, A)
        grids = {int(order[i]): gr for i, gr in zip(rec.near_cells, rec.near_grids)}      # sorted position -> cell id
        assert set(grids) == set(int(c) for c in engine.near_tie_cells(rec.summary[inv]))
        engine.write_doublet_summary(fa, rec.sing[inv], rec.llks00[inv], rec.summary[inv], os.path.join(outdir, "o"), cell_grids=grids)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["gt_v4_a2_pair", "gp_v8_a2_minsnp"])
def test_sharded_run_world2_gloo(name, tmp_path):
    import torch.multiprocessing as mp
    from demuxlet_amd import build
    from oracle import oracle_py
    build.build(); oracle_py.build()
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    gd = Golden(name)
    for suf in ("single", "sing2", "best"):
        assert (tmp_path / f"o.{suf}").read_bytes() == gd.files[suf], suf


def _near_rule(l12, l1, l2, s1, s2, tol=1e-7):
    """k_reduce's DMX_CELL_NEAR_RULE predicate (dmx::near_rule): a comparison of cmd_cram_demuxlet.cpp:837,:844 within tol of flipping."""
    return abs(l12 - (s1 + 2)) < tol or abs(s1 - (s2 + 2)) < tol or ((abs(l12 - l1) < tol or abs(l12 - l2) < tol) and l12 > s1 + 2 - tol)


@pytest.mark.parametrize("certified", [False, True])
def test_best_rule_thresholds_follow_the_exact_values(oracle, certified, tmp_path):
    """the BEST rule (cmd_cram_demuxlet.cpp:837 `LLK12 > LLK1 && LLK12 > LLK2 && LLK12 > SNG.LLK1 + 2`, :844 `SNG.LLK1 >
    SNG.LLK2 + 2`) is evaluated on device-derived accumulators.  Records whose values sit 1e-12 on the WRONG side of a threshold — what a
    device that lost the last bit of a log to libm produces when the exact margin is that small — must still print the reference's
    SNG- / DBL- / AMB-: the record carries DMX_CELL_NEAR_RULE (margin < 1e-7) and the writers re-evaluate the entries involved in the
    reference's operation order.  Six kinds of sabotage over the golden jobs; every file byte-identical to the reference's; and without
    the pileup (no arbiter) the sabotaged record is what gets printed, so the test has teeth."""
    from demuxlet_amd import build, capi, engine
    build.build()
    kinds_seen = set()
    for name in ("gt_v4_a2_pair", "gp_v8_a2_minsnp", "pl_v12_a6_pair", "gt_v5_dense", "gt_v24_a2_deep", "gp_v32_a2_dense"):
        gd = Golden(name)
        st = build_store(engine, gd.problem(oracle))
        pl = st.freeze()
        assert st.barcodes() == gd.ref_barcodes
        cnt = gd.z["ref_counters"]
        B, V, A = len(gd.ref_barcodes), len(gd.sample_ids), len(gd.alphas)
        grid, l00 = gd.z["ref_llksAB"], gd.z["ref_llks00"]
        summ = np.zeros(B, dtype=capi.SUMMARY_DTYPE)
        for c in range(B):
            if gd.z["ref_processed"][c]:
                summ[c] = summary_from_grid(grid[c], l00[c], gd.alphas, gd.doublet_prior, cnt[c, 3], capi.SUMMARY_DTYPE)
        sing = np.ascontiguousarray(grid[:, :, 0, 0]).copy()
        bad = summ.copy()
        eps = 1e-12
        n_sab = 0
        for q, c in enumerate(np.flatnonzero(summ["n_pairs"] > 0)):
            r = summ[c]
            i1, i2, jb, kb, nb = (int(r[k]) for k in ("i_sing1", "i_sing2", "j_best", "k_best", "n_best"))
            s1, s2, l12, l1, l2 = float(r["sing_llk1"]), float(r["sing_llk2"]), float(r["llk12"]), float(r["llk1"]), float(r["llk2"])
            dbl = l12 > l1 and l12 > l2 and l12 > s1 + 2
            sng = (not dbl) and s1 > s2 + 2
            if certified:                               # as after K3b: both accumulators of the alpha = 0.5 best pair known
                if gd.alphas[nb] != 0.5:
                    continue
                a, b = min(jb, kb), max(jb, kb)
                bad[c]["flags"] |= capi.DMX_CELL_ORDER_CERTIFIED
                bad[c]["llk_ab"], bad[c]["llk_ba"] = grid[c, a, b, nb], grid[c, b, a, nb]
            kind = None
            if dbl:                                     # the record loses the doublet by 1e-12 on one of the three comparisons
                kind = ("dbl_vs_l1", "dbl_vs_l2", "dbl_vs_sng")[q % 3]
                if kind == "dbl_vs_l1": bad[c]["llk1"] = l12 + eps          # LLK1 itself is [jb][0][0]: the writer re-evaluates it
                elif kind == "dbl_vs_l2": bad[c]["llk2"] = l12 + eps
                else: bad[c]["llk12"] = s1 + 2 - eps
            elif sng and q % 2 == 0:                     # SNG in the reference, AMB in the record
                kind = "sng_to_amb"
                sing[c, i2] = s1 - 2 + eps; bad[c]["sing_llk2"] = sing[c, i2]
            elif sng and l1 <= s1 and l2 <= s1:          # SNG in the reference, DBL in the record
                kind = "sng_to_dbl"
                bad[c]["llk12"] = s1 + 2 + eps
            elif (not dbl) and (not sng) and s1 - 2 - eps > np.partition(sing[c], -3)[-3] if V >= 3 else False:   # AMB in the reference, SNG in the record
                kind = "amb_to_sng"
                sing[c, i2] = s1 - 2 - eps; bad[c]["sing_llk2"] = sing[c, i2]
            if kind is None:
                continue
            assert _near_rule(float(bad[c]["llk12"]), float(bad[c]["llk1"]), float(bad[c]["llk2"]), float(sing[c, i1]), float(sing[c, i2]))
            bad[c]["flags"] |= capi.DMX_CELL_NEAR_RULE   # what k_reduce sets on such a record
            kinds_seen.add(kind); n_sab += 1
        if not n_sab:
            continue
        fa = engine.FinalArgs(gd.ref_barcodes, gd.sample_ids, gd.alphas, gd.doublet_prior, cnt[:, 0], cnt[:, 1], cnt[:, 2], cnt[:, 3],
                              gd.min_total, gd.min_uniq, gd.min_snp, False)
        pre = str(tmp_path / f"{name}_arb")
        engine.write_doublet_summary(fa, sing, l00, bad, pre, tie_pileup=pl, tie_g=gd.g)
        assert Path(pre + ".best").read_bytes() == gd.files["best"], name
        assert Path(pre + ".sing2").read_bytes() == gd.files["sing2"], name
        pre = str(tmp_path / f"{name}_raw")              # no arbiter: the record's own (wrong) comparisons are printed
        engine.write_doublet_summary(fa, sing, l00, bad, pre)
        got, ref = Path(pre + ".best").read_bytes().split(b"\n"), gd.files["best"].split(b"\n")
        assert len(got) == len(ref) and sum(x.split(b"\t")[5][:3] != y.split(b"\t")[5][:3] for x, y in zip(got[1:-1], ref[1:-1])) == n_sab, name
    assert {"dbl_vs_l1", "dbl_vs_l2", "dbl_vs_sng", "sng_to_amb"} <= kinds_seen, kinds_seen


def test_records_only_fallback_is_bounded(oracle, tmp_path):
    """a near-tie-flagged barcode without its grid has its whole grid re-evaluated on the host — a fall-back, not a way of
    life: beyond 2e9 host log() calls the writer refuses before it writes anything and names the entry point that takes the grids."""
    from demuxlet_amd import build, capi, engine
    build.build()
    gd = Golden("gt_v4_a2_pair")
    st = build_store(engine, gd.problem(oracle))
    pl = st.freeze()
    cnt = gd.z["ref_counters"]
    B = len(gd.ref_barcodes)
    grid, l00 = gd.z["ref_llksAB"], gd.z["ref_llks00"]
    summ = np.zeros(B, dtype=capi.SUMMARY_DTYPE)
    for c in range(B):
        if gd.z["ref_processed"][c]:
            summ[c] = summary_from_grid(grid[c], l00[c], gd.alphas, gd.doublet_prior, cnt[c, 3], capi.SUMMARY_DTYPE)
    c0 = int(np.flatnonzero(summ["n_pairs"] > 0)[0])
    summ[c0]["flags"] |= capi.DMX_CELL_NEAR_DOUBLET
    summ[c0]["n_pairs"] = 2_000_000_000              # (a record claiming 2e9 covered SNPs: 2e9 x 4 x 4 x 2 logs)
    fa = engine.FinalArgs(gd.ref_barcodes, gd.sample_ids, gd.alphas, gd.doublet_prior, cnt[:, 0], cnt[:, 1], cnt[:, 2], cnt[:, 3],
                          gd.min_total, gd.min_uniq, gd.min_snp, False)
    with pytest.raises(capi.DmxError) as ei:
        engine.write_doublet_summary(fa, grid[:, :, 0, 0], l00, summ, str(tmp_path / "o"), tie_pileup=pl, tie_g=gd.g)
    assert ei.value.code == capi.DMX_ERR_ARG and "dmx_write_doublet_summary_grids" in str(ei.value)
    assert not (tmp_path / "o.best").exists() and not (tmp_path / "o.sing2").exists()
