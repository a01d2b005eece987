"""GPU box: FAST K2 on wide soft-field panels — k_doublet_sym's slab form (V <= 512) against the STRICT kernel those panels run with DMX_NO_SYM_WIDE=1."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa
from demuxlet_amd import build, engine, synth
build.build()
for V in (192, 256, 384, 512):
    rng = np.random.default_rng(V)
    S, B = 20000, 256
    raw = synth.make_raw_genotypes(rng, S, V)
    g = np.stack([engine.geno_from_gp(x, 0.01) for x in synth.raw_gp_from_alleles(rng, raw.alleles)])
    sp = synth.make_pileup(rng, raw.alleles, B, 0.25, 1.25, dense_layout=False, doublet_rate=0.3)
    pl = engine.HostPileup(B, S, sp.cell_pair_off, sp.cell_read_off, sp.pair_snp, sp.pair_nrd, sp.reads, sp.rd_totl, sp.rd_pass, sp.rd_uniq)
    res = {}
    for name, env in (("sym", None), ("strict_a2", "DMX_NO_SYM_WIDE")):
        if env: os.environ[env] = "1"
        e = engine.Engine(V, (0.0, 0.5), 0.5, device=0, mode=engine.capi.DMX_MODE_FAST)
        e.set_genotypes(g); e.set_pileup(pl)
        e.run_singlet(); e.run_doublet(); e.get_doublet()
        e.reset_kernel_times()
        for _ in range(3):
            e.run_doublet()
        grid, l00, summ = e.get_doublet()
        km = e.mean_kernel_times()
        res[name] = (km.doublet_ms, grid)
        e.close()
        if env: del os.environ[env]
    m = np.zeros((V, V, 2), bool); m[:, 0, 0] = True; m[:, :, 1] = True
    d = np.abs(res["sym"][1] - res["strict_a2"][1])[np.broadcast_to(m[None], res["sym"][1].shape)].max()
    print(f"V={V}: k_doublet_sym slabs {res['sym'][0]:.2f} ms, STRICT k_doublet_a2 (what FAST ran there before this change: the generic kernel) {res['strict_a2'][0]:.2f} ms, max |delta| on printed entries {d:.2e}", flush=True)
