"""GPU (-m gpu): the `.pair` rows of `--write-pair` (cmd_cram_demuxlet.cpp:772-797) formatted ON THE DEVICE (dmx_engine_format_pair, ABI 8;
csrc/dmx_format.hpp) are the host formatter's bytes.

* crafted grids: every magnitude a log-likelihood and a posterior can take (negative zero, subnormals, 1e-3 ... 8e12, the denormal range of exp, values that
  underflow for certain), nan / inf / |v| >= 2^43 (left to the host: flag 2) — field by field against Python's '%.5f' / '%.5g' (glibc printf);
* real problems: the text with its patches spliced in == the file dmx_write_doublet writes from the same grid;
* the job: dmx_demuxlet_run with write_pair, device formatting (default) vs the host formatter (DMX_PAIR_ON_HOST=1, fenced): four files byte for byte, one
  range and many, with barcodes whose rows stay on the host (duplicate samples -> near-tie flags)."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from demuxlet_amd import build, capi, engine
    build.build()
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    capi.load()
    return engine


def small_engine(eng, V, alphas, B, S=60, seed=3, dup=False):
    from demuxlet_amd import synth
    rng = np.random.default_rng(seed)
    raw = synth.make_raw_genotypes(rng, S, V)
    if dup and V >= 4:
        raw.alleles[:, 1] = raw.alleles[:, 0]
    g = np.stack([eng.geno_from_gt(raw.alleles[s], 0.01) for s in range(S)])
    sp = synth.make_pileup(rng, raw.alleles, B, 0.5, 1.5, doublet_rate=0.3)
    pl = eng.HostPileup(sp.n_cells, sp.n_snps, sp.cell_pair_off, sp.cell_read_off, sp.pair_snp, sp.pair_nrd, sp.reads, sp.rd_totl, sp.rd_pass, sp.rd_uniq)
    e = eng.Engine(V, alphas, 0.5)
    e.set_genotypes(g); e.set_pileup(pl); e.run(); e.sync()
    return e, g, sp, pl


def rows_of(V, alphas):
    out = []
    for j in range(V):
        out.append((j, j, 0))
        for k in range(V):
            for a in range(1, len(alphas)):
                if j == k or (j > k and alphas[a] == 0.5):
                    continue
                out.append((j, k, a))
    return out


def splice(text, patches, summ, cells, V, A, prior):
    """The text with the fields the device left to the host's libm filled in (what dmx_demuxlet_run's writer does)."""
    out, pos = [], 0
    for off, v, oc, singlet in patches:
        out.append(text[pos:off]); pos = off
        sm = summ[cells[oc]]
        tot = float(sm["sum_single"]) + float(sm["sum_double"])
        x = v - float(sm["max_llk"])
        e = math.exp(x)
        p = e * (1. - prior) / V / tot if singlet else e * prior / V / (V - 1) / (A - 1) / tot
        out.append(("%.5g" % p).encode())
    out.append(text[pos:])
    return b"".join(out)


def test_crafted_values_field_by_field(eng):
    import torch
    from demuxlet_amd import synth_torch as st
    V, alphas, B = 8, (0.0, 0.5), 3000
    A = len(alphas)
    e, g, sp, pl = small_engine(eng, V, alphas, B)
    dev = torch.device("cuda", 0)
    view = e.device_view()
    grid = st.tensor_from_ptr(view.llksAB, (B, V, V, A), torch.float64, dev)
    words = eng.capi.SUMMARY_DTYPE.itemsize // 8
    summ_t = st.tensor_from_ptr(view.summary, (B, words), torch.float64, dev)
    rng = np.random.default_rng(11)
    vals = np.empty((B, V, V, A))
    kind = rng.integers(0, 8, size=vals.shape)
    vals[:] = -rng.uniform(0, 760, size=vals.shape)                                   # posterior from 1 down through the denormal range to 0
    vals[kind == 1] = -np.exp(rng.uniform(np.log(1e-3), np.log(8e12), size=(kind == 1).sum()))
    vals[kind == 2] = np.exp(rng.uniform(np.log(1e-320), np.log(1e-3), size=(kind == 2).sum()))   # subnormal ... tiny positive
    vals[kind == 3] = -rng.uniform(0, 50, size=(kind == 3).sum())                       # posteriors that print in fixed notation
    vals[kind == 4] = np.round(-rng.uniform(0, 1e5, size=(kind == 4).sum()), 5)         # decimal-looking values (ties of the 6th digit)
    vals[kind == 5] = -rng.uniform(800, 2e5, size=(kind == 5).sum())                    # posteriors that underflow for certain: "0" without exp()
    vals.reshape(-1)[:8] = [-0.0, 0.0, -5e-324, 2.5e-6, -2.5e-6, 1.5e-5, -99999.999995, -123456.785]
    special = np.zeros(B, dtype=bool)
    for c, v in ((5, np.nan), (6, np.inf), (7, -np.inf), (8, -9e12), (9, 2.0 ** 43)):
        vals[c, 3, 4, 1] = v; special[c] = True
    grid.copy_(torch.from_numpy(vals).to(dev))
    summ = np.ascontiguousarray(summ_t.cpu().numpy()).view(eng.capi.SUMMARY_DTYPE).reshape(-1).copy()
    summ["max_llk"] = 0.0
    summ["sum_single"] = rng.uniform(0.2, 0.9, size=B); summ["sum_double"] = rng.uniform(0.0, 0.5, size=B)
    summ["max_llk"][100:200] = rng.uniform(-3, 3, size=100)
    summ_t.copy_(torch.from_numpy(summ.view(np.float64).reshape(B, words)).to(dev))
    torch.cuda.synchronize()
    cells = np.arange(B, dtype=np.int32)[::-1].copy()                                   # any output order
    bcs = [f"BC{c:05d}-1" for c in cells]
    sms = [f"S{j}x" * (1 + j % 3) for j in range(V)]
    text, off, flag, patches, ms = e.format_pair(cells, bcs, sms)
    assert [int(f) for f in flag] == [2 if special[c] else 0 for c in cells]
    assert all(off[i + 1] == off[i] for i in range(B) if flag[i])
    pat = {}
    for o, v, oc, sg in patches:
        pat[o] = (v, oc, sg)
    rows = rows_of(V, alphas)
    n_fields = n_patched = 0
    for i, c in enumerate(cells):
        if flag[i]:
            continue
        seg = text[off[i]:off[i + 1]]
        lines = seg.split(b"\n")
        assert lines[-1] == b"" and len(lines) == len(rows) + 1
        pos = int(off[i])
        tot = float(summ["sum_single"][c]) + float(summ["sum_double"][c])
        for (j, k, a), ln in zip(rows, lines):
            f = ln.split(b"\t")
            v = vals[c, j, 0 if a == 0 else k, a]
            assert f[0] == bcs[i].encode() and f[1] == sms[j].encode() and f[2] == sms[k].encode() and f[3] == (b"%.3f" % alphas[a])
            assert f[4] == ("%.5f" % v).encode(), (c, j, k, a, v, f[4])
            post_at = pos + len(ln) - len(f[5])
            x = v - float(summ["max_llk"][c])
            if post_at in pat and f[5] == b"":
                assert pat[post_at] == (v, i, 1 if a == 0 else 0)
                n_patched += 1
            else:
                assert post_at not in pat
                ex = math.exp(x)
                p = ex * (1. - 0.5) / V / tot if a == 0 else ex * 0.5 / V / (V - 1) / (A - 1) / tot
                assert f[5] == ("%.5g" % p).encode(), (c, j, k, a, v, p, f[5])
                n_fields += 1
            pos += len(ln) + 1
        assert pos == off[i + 1]
    assert n_fields > 50_000 and 0 < n_patched < n_fields // 4
    print(f"{n_fields} POSTPRB fields printed on the device, {n_patched} left to the host; {ms:.2f} ms")
    e.close()


@pytest.mark.parametrize("V,alphas,dup", [(4, (0.0, 0.5), False), (12, (0.0, 0.25, 0.5), False), (7, (0.1, 0.5, 0.9), False), (33, (0.0, 0.5), True)])
def test_device_text_is_the_host_formatters_file(eng, tmp_path, V, alphas, dup):
    B = 150
    e, g, sp, pl = small_engine(eng, V, alphas, B, S=40 if V < 30 else 25, seed=V, dup=dup)
    grid, l00, summ = e.get_doublet()
    A = len(alphas)
    bcs = [f"BC{(i * 7919) % 1000:04d}-1" for i in range(B)]
    sms = [f"SM{j:02d}" for j in range(V)]
    fa = eng.FinalArgs(bcs, sms, alphas, 0.5, sp.rd_totl, sp.rd_pass, sp.rd_uniq, pl.n_snp_per_cell, write_pair=True)
    eng.write_doublet(fa, grid, l00, str(tmp_path / "h"))                                # host formatter, no arbiter: the device grid as it is
    order = sorted(range(B), key=lambda i: bcs[i].encode())
    cells = np.array([i for i in order if pl.n_snp_per_cell[i] > 0], dtype=np.int32)
    # what the host formatter prints for the certified entries of an alpha = 0.5 best doublet (no near-tie flag): the certificate's values
    ovr = []
    for c in cells:
        sm = summ[c]
        if (sm["flags"] & eng.capi.DMX_CELL_ORDER_RESOLVABLE):
            one = summ[c:c + 1].copy(); eng.resolve_tie_order(one); sm = one[0]
        if (sm["flags"] & eng.capi.DMX_CELL_ORDER_CERTIFIED) and not (sm["flags"] & 3):
            ovr.append((min(sm["j_best"], sm["k_best"]), max(sm["j_best"], sm["k_best"]), sm["n_best"], 0, sm["llk_ab"], sm["llk_ba"]))
        else:
            ovr.append(None)
    text, off, flag, patches, ms = e.format_pair(cells, [bcs[c] for c in cells], sms, ovr=ovr)
    assert not flag.any()
    # the host formatter takes maxLLK and the two sums from the grid with the host libm; the device text uses K3's record — equal to the last digit printed
    got = b"BARCODE\tSM1.ID\tSM2.ID\tLLK12\tPOSTPRB\n" + splice(text, patches, summ, cells, V, A, 0.5)
    want = (tmp_path / "h.pair").read_bytes()
    assert got == want
    e.close()


@pytest.mark.parametrize("V,field,mode,rb", [(6, "GT", "strict", None), (6, "GT", "strict", "20000"), (16, "GP", "fast", "60000"), (5, "GT", "strict", "9000")])
def test_job_with_write_pair_device_vs_host_formatter(eng, tmp_path, monkeypatch, V, field, mode, rb):
    from demuxlet_amd import synth, capi
    rng = np.random.default_rng(900 + V)
    S, B = 300, 400
    raw = synth.make_raw_genotypes(rng, S, V)
    if V == 5:                                      # duplicate samples: near-tie flags -> barcodes whose rows stay on the host formatter
        raw.alleles[:, 1] = raw.alleles[:, 0]; raw.alleles[:, 3] = raw.alleles[:, 2]
    if field == "GT":
        g = np.stack([eng.geno_from_gt(raw.alleles[s], 0.01) for s in range(S)])
    else:
        g = np.stack([eng.geno_from_gp(x, 0.01) for x in synth.raw_gp_from_alleles(rng, raw.alleles)])
    sp = synth.make_pileup(rng, raw.alleles, B, 0.05 if V == 5 else 0.3, 1.5, doublet_rate=0.4)
    pl = eng.HostPileup(sp.n_cells, sp.n_snps, sp.cell_pair_off, sp.cell_read_off, sp.pair_snp, sp.pair_nrd, sp.reads, sp.rd_totl, sp.rd_pass, sp.rd_uniq)
    bcs = [f"BC{(i * 7919) % 100003:06d}-1" for i in range(B)]
    sms = [f"S{j:02d}" for j in range(V)]
    md = capi.DMX_MODE_FAST if mode == "fast" else capi.DMX_MODE_STRICT
    monkeypatch.setenv("DMX_EXPERIMENTS", "1")
    if rb: monkeypatch.setenv("DMX_RANGE_BYTES", rb)
    else: monkeypatch.delenv("DMX_RANGE_BYTES", raising=False)
    monkeypatch.delenv("DMX_PAIR_ON_HOST", raising=False)
    td = eng.demuxlet_run(pl, g, sms, (0.0, 0.5), str(tmp_path / "d"), write_pair=True, barcodes=bcs, mode=md, timing=True)
    monkeypatch.setenv("DMX_PAIR_ON_HOST", "1")
    th = eng.demuxlet_run(pl, g, sms, (0.0, 0.5), str(tmp_path / "h"), write_pair=True, barcodes=bcs, mode=md, timing=True)
    assert td["n_ranges"] == th["n_ranges"] and (rb is None or td["n_ranges"] > 2)
    for suf in ("single", "sing2", "best", "pair"):
        assert (tmp_path / f"d.{suf}").read_bytes() == (tmp_path / f"h.{suf}").read_bytes(), suf
    n_rows = sum(1 for _ in open(tmp_path / "d.pair")) - 1
    covered = int((pl.n_snp_per_cell > 0).sum())
    assert n_rows == covered * (V + V * (V - 1) // 2)
