"""GPU box: STRICT K2 on wide soft-field panels — k_doublet_a2<256,16> with up to 150 KB of LDS per workgroup (129 <= V <= 384, round 4)
against the generic kernel those panels used to take (DMX_K2_GENERIC=1).  Prints both times and whether the grids are bit-identical."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa
from demuxlet_amd import build, engine, synth
build.build()
for V in (160, 192, 256, 384, 512, 768, 1024):
    rng = np.random.default_rng(V)
    S, B = 20000, (256 if V <= 384 else 64)
    raw = synth.make_raw_genotypes(rng, S, V)
    g = np.stack([engine.geno_from_gp(x, 0.01) for x in synth.raw_gp_from_alleles(rng, raw.alleles)])
    sp = synth.make_pileup(rng, raw.alleles, B, 0.1, 1.25, dense_layout=False, doublet_rate=0.3)
    pl = engine.HostPileup(B, S, sp.cell_pair_off, sp.cell_read_off, sp.pair_snp, sp.pair_nrd, sp.reads, sp.rd_totl, sp.rd_pass, sp.rd_uniq)
    res = {}
    runs = [("a2", None, None), ("generic", "DMX_K2_GENERIC", "1")] + [(f"a2/tp{c}", "DMX_A2_TP", str(c)) for c in (32, 16, 8)]
    for name, env, val in runs:
        if env: os.environ[env] = val
        e = engine.Engine(V, (0.0, 0.5), 0.5, device=0)
        e.set_genotypes(g); e.set_pileup(pl)
        e.run_singlet(); e.run_doublet(); e.get_doublet()
        e.reset_kernel_times()
        for _ in range(2):
            e.run_doublet()
        grid, l00, summ = e.get_doublet()
        km = e.mean_kernel_times()
        res[name] = (km.doublet_ms, grid, l00)
        e.close()
        if env: del os.environ[env]
    same = np.array_equal(res["a2"][1], res["generic"][1]) and np.array_equal(res["a2"][2], res["generic"][2])
    n = float(sp.cell_pair_off[-1]) * V * V * 2
    print(f"V={V}: " + " ".join(f"{k}={v[0]:.1f}ms" for k, v in res.items()), flush=True)
    print(f"V={V}: k_doublet_a2<256,16> {res['a2'][0]:.1f} ms ({n / res['a2'][0] / 1e6:.0f} G entries/s), k_doublet_generic {res['generic'][0]:.1f} ms, "
          f"bit-identical grids: {same}", flush=True)
