"""GPU (-m gpu): BASELINE.json's configurations at their FULL sizes (the workloads bench.py times), checked two ways:

* a sample of barcodes (first / spread / last of the launch) is copied back and pushed through the oracle at the full SNP depth
  -> every log-likelihood within 1e-9 absolute, the per-cell K3 indices equal to the reference's scans on the device grid;
* size-independent properties on ALL barcodes: the genotype-class kernels equal the general kernels bit-for-bit, a re-run is
  bit-identical (no atomics, no order dependence on scheduling), and the sampled barcodes computed ALONE (another launch
  geometry: other wavefronts, other cells-per-wavefront) give the same bits as inside the full launch.

Nothing here reads /root/reference; the oracle is the checker only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def mods():
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from demuxlet_amd import build, capi, engine, synth, synth_torch
    build.build()
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    capi.load()
    import bench
    from oracle import oracle_py as O
    O.build()
    return dict(torch=torch, engine=engine, synth=synth, st=synth_torch, bench=bench, O=O)


def sample_cells(B, n, seed):
    rng = np.random.default_rng(seed)
    k = max(1, n // 3)
    mid = rng.choice(np.arange(k, B - k), size=n - 2 * k, replace=False)
    return np.unique(np.concatenate([np.arange(k), mid, np.arange(B - k, B)])).astype(np.int64)


def slice_device_pileup(m, dp, cells):
    """The chosen cells of a DevicePileup as (a) host arrays for the oracle, (b) a new DevicePileup with only those cells."""
    torch = m["torch"]
    po = dp.cell_pair_off.cpu().numpy()
    ro = dp.cell_read_off.cpu().numpy()
    pidx = torch.cat([torch.arange(po[c], po[c + 1], device=dp.pair_nrd.device) for c in cells])
    ridx = torch.cat([torch.arange(ro[c], ro[c + 1], device=dp.reads.device) for c in cells])
    npair = po[cells + 1] - po[cells]
    nread = ro[cells + 1] - ro[cells]
    z = np.zeros(1, dtype=np.int64)
    sub = m["st"].DevicePileup(len(cells), dp.n_snps,
                               torch.from_numpy(np.concatenate([z, np.cumsum(npair)])).to(dp.reads.device),
                               torch.from_numpy(np.concatenate([z, np.cumsum(nread)])).to(dp.reads.device),
                               None if dp.pair_snp is None else dp.pair_snp[pidx].contiguous(),
                               dp.pair_nrd[pidx].contiguous(), dp.reads[ridx].contiguous(), dp.truth[torch.from_numpy(cells).to(dp.truth.device)])
    return sub


def oracle_on(m, sub, g, cfg):
    O = m["O"]
    n = sub.n_cells
    reads = sub.reads.cpu().numpy()
    words = ((reads >> 7).astype(np.uint32) << 24) | ((reads & 0x7F).astype(np.uint32) << 16) | 1
    pair_snp = (sub.pair_snp.cpu().numpy() if sub.pair_snp is not None
                else np.tile(np.arange(sub.n_snps, dtype=np.int32), n))
    nrd = sub.pair_nrd.cpu().numpy().astype(np.int64)
    csr = O.Csr([f"c{i:07d}" for i in range(n)], sub.cell_pair_off.cpu().numpy(), pair_snp,
                np.concatenate([[0], np.cumsum(nrd)]), words, np.zeros(n, np.int32), np.zeros(n, np.int32),
                np.zeros(n, np.int32))
    return O.run_csr(csr, [f"s{j}" for j in range(g.shape[1])], g, O.Params(tuple(cfg["alphas"]), 0.5), None,
                     singlet_only=not cfg["doublet"], want_grid=cfg["doublet"])


def device_results(m, eng, B, V, A, doublet, rows=None):
    torch, st = m["torch"], m["st"]
    dev = torch.device("cuda", 0)
    v = eng.device_view()
    pick = (lambda t: t) if rows is None else (lambda t: t[torch.from_numpy(rows).to(dev)])
    out = dict(llks=pick(st.tensor_from_ptr(v.llks, (B, V), torch.float64, dev)).cpu().numpy(),
               llk0s=pick(st.tensor_from_ptr(v.llk0s, (B,), torch.float64, dev)).cpu().numpy())
    if doublet:
        out["grid"] = pick(st.tensor_from_ptr(v.llksAB, (B, V, V, A), torch.float64, dev)).cpu().numpy()
        out["l00"] = pick(st.tensor_from_ptr(v.llks00, (B, A), torch.float64, dev)).cpu().numpy()
        words = m["engine"].capi.SUMMARY_DTYPE.itemsize // 8
        sm = pick(st.tensor_from_ptr(v.summary, (B, words), torch.float64, dev)).cpu().numpy()
        out["summ"] = np.ascontiguousarray(sm).view(m["engine"].capi.SUMMARY_DTYPE).reshape(-1)
    return out


def run_full(m, cfg_id, n_sample, check_general, barcodes=0, fast=False):
    torch, engine, bench = m["torch"], m["engine"], m["bench"]
    cfg = dict(bench.CONFIGS[cfg_id])
    if barcodes:
        cfg["B"] = barcodes
    B, S, V, A = cfg["B"], cfg["S"], cfg["V"], len(cfg["alphas"])
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0xD3A00000 + cfg_id)          # bench.py's panel and pileup for this config
    raw, g = bench.genotype_matrix(engine, m["synth"], rng, S, V, cfg["field"])
    dosage = torch.from_numpy(np.clip(raw.alleles, 0, 1).sum(axis=2).astype(np.float32)).to(dev)
    dp = m["st"].make_device_pileup(dosage, B, cfg["delta"], cfg["rbar"], seed=0xD3A0 + 1000 * cfg_id, device=dev)
    torch.cuda.synchronize()

    def run(pileup, env=None, one_call=False):
        """one_call: dmx_engine_run — what bench.py times and dmx_demuxlet_run calls (K1 on its own stream beside K2, or after it beside K3 + K3b);
        otherwise run_singlet, then run_doublet, one after the other on the engine's stream."""
        old = {}
        for k, val in (env or {}).items():
            old[k] = os.environ.get(k)
            os.environ[k] = val
        try:
            e = engine.Engine(V, cfg["alphas"], 0.5, device=0, mode=engine.capi.DMX_MODE_FAST if fast else engine.capi.DMX_MODE_STRICT)
            e.set_genotypes(g)
            e.set_pileup_struct(pileup.as_struct(), keep=pileup)
            if one_call and cfg["doublet"]:
                e.run()
            else:
                e.run_singlet()
                if cfg["doublet"]:
                    e.run_doublet()
            e.sync()
            return e
        finally:
            for k, val in old.items():
                if val is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = val

    cells = sample_cells(B, n_sample, 1234 + cfg_id)
    # the oracle-compared pass is the path the driver times, at the size it is timed at: dmx_engine_run at full size, where
    # K1 really overlaps K2 (or K3 + K3b) — a fork / join or event-ordering fault shows here as a wrong or a changing value
    e = run(dp, one_call=True)
    if cfg["doublet"]:
        names = e.kernel_names()
        # DESIGN 6 "K1 beside K2": K1 starts after K2 under k_doublet_clsp (cfg4) and under k_doublet_sym with ONE long K1 launch (cfg3 FAST;
        # a K1 walked in SNP blocks — cfg5 — fits in between), beside K2 everywhere else
        k2 = names["doublet"]
        want_place = 2 if (k2.startswith("k_doublet_clsp<") or (k2.startswith("k_doublet_sym<") and cfg["delta"] >= 1.0)) else 1
        assert names["k1_placement"] == want_place, names
    full = device_results(m, e, B, V, A, cfg["doublet"])
    # (1) the sampled barcodes against the oracle at full depth
    sub = slice_device_pileup(m, dp, cells)
    want = oracle_on(m, sub, g, cfg)
    d1 = np.abs(full["llks"][cells] - want.llks).max()
    d0 = np.abs(full["llk0s"][cells] - want.llk0s).max()
    assert d1 <= TOL and d0 <= TOL, (d1, d0)
    worst = max(d1, d0)
    if cfg["doublet"]:
        dgrid = np.abs(full["grid"][cells] - want.llksAB)
        if fast:                                 # FAST computes the entries demuxlet prints or decides on (golden_util.printed_mask)
            from golden_util import printed_mask
            dgrid = dgrid[np.broadcast_to(printed_mask(V, A)[None], dgrid.shape)]
        dg = dgrid.max()
        dl = np.abs(full["l00"][cells] - want.llks00).max()
        assert dg <= TOL and dl <= TOL, (dg, dl)
        worst = max(worst, dg, dl)
        # K3's indices are the reference's scans (strict <, first maximum) of the device grid
        from golden_util import summary_from_grid
        for c in cells:
            sm = full["summ"][c]
            ref = summary_from_grid(full["grid"][c], full["l00"][c], cfg["alphas"], 0.5, int(sm["n_pairs"]),
                                    m["engine"].capi.SUMMARY_DTYPE)
            assert sm["max_llk"] == ref["max_llk"]
            for f in ("i_sing1", "i_sing2", "n_best"):
                assert sm[f] == ref[f], (c, f)
            if sm["flags"] & m["engine"].capi.DMX_CELL_ORDER_CERTIFIED:      # K3b may have turned the pair into the reference's order
                assert {int(sm["j_best"]), int(sm["k_best"])} == {int(ref["j_best"]), int(ref["k_best"])}, c
            else:
                assert (sm["j_best"], sm["k_best"]) == (ref["j_best"], ref["k_best"]), c
        # ... and the calls they lead to are the ORACLE's calls: same best/next singlet, same doublet pair (its two samples may
        # come in either order at alpha = 0.5, which is what the host tie arbiter settles, DESIGN.md "Ties"), same alpha
        for i, c in enumerate(cells):
            sm = full["summ"][c]
            want_s = summary_from_grid(want.llksAB[i], want.llks00[i], cfg["alphas"], 0.5, int(sm["n_pairs"]), m["engine"].capi.SUMMARY_DTYPE)
            assert (sm["i_sing1"], sm["i_sing2"], sm["n_best"]) == (want_s["i_sing1"], want_s["i_sing2"], want_s["n_best"]), c
            assert {int(sm["j_best"]), int(sm["k_best"])} == {int(want_s["j_best"]), int(want_s["k_best"])}, c
            if sm["flags"] & m["engine"].capi.DMX_CELL_ORDER_CERTIFIED:      # certified: the oracle's order and LLK12 bits
                assert (int(sm["j_best"]), int(sm["k_best"])) == (int(want_s["j_best"]), int(want_s["k_best"])), c
                assert sm["llk12"] == want_s["llk12"], c
    # (2) a second run — the kernels one after the other (run_singlet; run_doublet) — is bit-identical to the one-call run
    e2 = run(dp)
    again = device_results(m, e2, B, V, A, cfg["doublet"])
    for k in ("llks", "llk0s") + (("grid", "l00") if cfg["doublet"] else ()):
        assert np.array_equal(full[k], again[k]), k
    e2.close()
    # (3) the sampled barcodes alone: another launch geometry, same bits
    e3 = run(sub)
    alone = device_results(m, e3, len(cells), V, A, cfg["doublet"])
    for k in ("llks", "llk0s") + (("grid", "l00") if cfg["doublet"] else ()):
        assert np.array_equal(alone[k], full[k][cells]), k
    e3.close()
    # (4) genotype-class kernels == general kernels on every barcode (GT inputs only)
    if check_general:
        e4 = run(dp, env={"DMX_NO_CLASSES": "1"})
        gen = device_results(m, e4, B, V, A, cfg["doublet"])
        for k in ("llks", "llk0s") + (("grid", "l00") if cfg["doublet"] else ()):
            assert np.array_equal(full[k], gen[k]), k
        e4.close()
    e.close()
    assert np.abs(want.llks).min() > 0 and np.isfinite(full["llks"]).all()
    print(f"cfg{cfg_id}: {len(cells)} of {B} barcodes at S={S} through the oracle, max |delta| = {worst:.3e} "
          f"(|llk| up to {np.abs(want.llks).max():.4e})")
    return worst


def test_cfg2_full_size(mods):
    """10k barcodes x 50k SNPs x 8 samples, GT, singlet-only: 48 barcodes through the oracle + properties on all 10k."""
    run_full(mods, 2, 48, check_general=True)


def test_cfg3_full_size(mods):
    """10k x 50k x 32, GP, alpha {0, 0.5}: 4 barcodes x 1.0e8 pair-evaluations through the oracle + properties on all."""
    run_full(mods, 3, 4, check_general=False)


def test_cfg5_full_size(mods):
    """20k x 200k x 16, PL, sparse: 12 barcodes through the oracle + properties on all."""
    run_full(mods, 5, 12, check_general=False)


def test_cfg4_full_depth_and_panel(mods):
    """cfg4's SNP depth and panel (100k SNPs x 64 samples, GT) on 1 000 of its barcodes (its 12.5k-barcode shard repeats this
    work 12.5x; the oracle needs 15 s per barcode here): 2 barcodes x 8.2e8 pair-evaluations through the oracle, class kernels
    == general kernels on all 1 000."""
    run_full(mods, 4, 2, check_general=True, barcodes=1000)


def test_cfg3_full_size_fast_mode(mods):
    """DMX_MODE_FAST at cfg3's full depth (50 k SNPs per barcode, |LLK| ~ 1e5): the factored doublet term keeps every sampled
    log-likelihood within 1e-9 of the oracle, re-runs stay bit-identical, barcodes computed alone give the same bits."""
    worst = run_full(mods, 3, 4, check_general=False, fast=True)
    assert worst < 1e-9


def test_cfg4_full_depth_fast_mode(mods):
    """DMX_MODE_FAST on cfg4's depth and panel (100k SNPs x 64 samples, GT -> k_doublet_clsp<FAST>): every printed entry of the two
    sampled barcodes within 1e-9 of the oracle, the calls the oracle's, re-runs and other launch geometries bit-identical."""
    worst = run_full(mods, 4, 2, check_general=False, barcodes=1000, fast=True)
    assert worst < 1e-9


def test_cfg5_full_size_fast_mode(mods):
    """DMX_MODE_FAST on cfg5 (20k x 200k x 16, PL, sparse -> k_doublet_sym, one wavefront per barcode)."""
    worst = run_full(mods, 5, 12, check_general=False, fast=True)
    assert worst < 1e-9


def test_cfg6_full_size(mods):
    """SURVEY 8(d)'s realistic-coverage shape (bench.py's cfg6: 20k barcodes x 100k SNPs x 16 samples, GT, delta = 0.02 -> ~2 000 covered SNPs
    per barcode, what a 10x droplet looks like; cmd_cram_demuxlet.cpp:592): 12 barcodes through the oracle + the size-independent properties
    on all 20k, genotype-class kernels == general kernels bit for bit."""
    run_full(mods, 6, 12, check_general=True)


def test_cfg6_full_size_fast_mode(mods):
    """DMX_MODE_FAST on cfg6 (GT, 16 samples, sparse -> k_doublet_clsym)."""
    worst = run_full(mods, 6, 12, check_general=False, fast=True)
    assert worst < 1e-9


def test_cfg4_slice_on_distinct_devices_with_write_pair(mods, tmp_path, monkeypatch):
    """The same slice with dmx_job.n_gpus = min(8, visible devices) (what `demuxlet --gpus N --write-pair` runs): two engines per
    DEVICE, ranges of the sorted barcodes alternating between them, the `.pair` rows appended in barcode order.  Skipped on a 1-GPU box
    (there the multi-engine tests put every engine on device 0)."""
    n = min(8, mods["torch"].cuda.device_count())
    if n < 2:
        pytest.skip("one visible device")
    run_cfg4_slice(mods, tmp_path, monkeypatch, n_gpus=n)


def test_cfg4_slice_streamed_with_write_pair(mods, tmp_path, monkeypatch):
    run_cfg4_slice(mods, tmp_path, monkeypatch, n_gpus=1)


def run_cfg4_slice(mods, tmp_path, monkeypatch, n_gpus):
    """A slice of cfg4's per-GPU shard (2 000 of its barcodes x 100 k SNPs x 64 samples, GT, dense) through dmx_demuxlet_run with
    `--write-pair` and a forced 16 MiB range budget: the barcodes stream through the engine in ~8 ranges of the sorted order, the
    4.2 M rows are appended range by range.  Three sampled barcodes are pushed through the oracle at full depth: their rows of all
    four files must be the oracle's (strings and BEST identical, numbers to the last printed digit)."""
    torch, engine, bench, O = mods["torch"], mods["engine"], mods["bench"], mods["O"]
    cfg = dict(bench.CONFIGS[4])
    B, S, V = 2000, cfg["S"], cfg["V"]
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0xD3A00000 + 4)
    raw, g = bench.genotype_matrix(engine, mods["synth"], rng, S, V, cfg["field"])
    dosage = torch.from_numpy(np.clip(raw.alleles, 0, 1).sum(axis=2).astype(np.float32)).to(dev)
    dp = mods["st"].make_device_pileup(dosage, B, cfg["delta"], cfg["rbar"], seed=0xD3A0 + 4000, device=dev)
    h = dp.host_slice(0, B)
    nrd = h["pair_nrd"].astype(np.int64)
    per_cell_reads = np.diff(h["cell_read_off"]).astype(np.int32)
    pl = engine.HostPileup(B, S, h["cell_pair_off"], h["cell_read_off"], None, h["pair_nrd"], h["reads"], per_cell_reads, per_cell_reads,
                           per_cell_reads)
    barcodes = [f"BC{(i * 7919) % B:05d}-1" for i in range(B)]          # sorted order != id order
    sm = [f"SM{j:02d}" for j in range(V)]
    del dp, dosage
    torch.cuda.empty_cache()
    monkeypatch.setenv("DMX_RANGE_BYTES", str(16 << 20))
    tm = engine.demuxlet_run(pl, g, sm, cfg["alphas"], str(tmp_path / "o"), write_pair=True, arbiter=True, barcodes=barcodes, timing=True,
                             n_gpus=n_gpus)
    assert tm["n_ranges"] >= 8 and tm["n_engines"] == (2 * n_gpus if tm["n_ranges"] > n_gpus else n_gpus)
    print("cfg4 slice, --write-pair, streamed:", {k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items()})
    cells = np.array([3, 977, 1999])
    pidx = np.concatenate([np.arange(h["cell_pair_off"][c], h["cell_pair_off"][c + 1]) for c in cells])
    ridx = np.concatenate([np.arange(h["cell_read_off"][c], h["cell_read_off"][c + 1]) for c in cells])
    reads = h["reads"][ridx]
    words = ((reads >> 7).astype(np.uint32) << 24) | ((reads & 0x7F).astype(np.uint32) << 16) | 1
    z = np.zeros(1, dtype=np.int64)
    csr = O.Csr([barcodes[c] for c in cells], np.concatenate([z, np.cumsum(np.full(len(cells), S, dtype=np.int64))]),
                np.tile(np.arange(S, dtype=np.int32), len(cells)), np.concatenate([z, np.cumsum(nrd[pidx])]), words,
                per_cell_reads[cells], per_cell_reads[cells], per_cell_reads[cells])
    O.run_csr(csr, sm, g, O.Params(tuple(cfg["alphas"]), 0.5, 0, 0, 0, True), str(tmp_path / "r"))
    want_bc = {barcodes[c] for c in cells}
    for suf in ("single", "sing2", "best", "pair"):
        got_all = (tmp_path / f"o.{suf}").read_text().splitlines()
        want = (tmp_path / f"r.{suf}").read_text().splitlines()
        assert got_all[0] == want[0]
        keys = [r.split("\t")[0] for r in got_all[1:]]
        assert keys == sorted(keys)                                      # ascending barcode order across the appended ranges
        got = [got_all[0]] + [r for r in got_all[1:] if r.split("\t")[0] in want_bc]
        assert len(got) == len(want), suf
        for a, b in zip(got[1:], want[1:]):
            fa, fb = a.split("\t"), b.split("\t")
            assert len(fa) == len(fb)
            for x, y in zip(fa, fb):
                try:
                    fx, fy = float(x), float(y)
                    assert abs(fx - fy) <= 1e-3 * max(1e-3, abs(fy)) + 1.01e-4, (suf, a, b)
                except ValueError:
                    assert x == y, (suf, a, b)
    n_pair_rows = sum(1 for _ in open(tmp_path / "o.pair")) - 1
    assert n_pair_rows == B * (V + V * (V - 1) // 2)
