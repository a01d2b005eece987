"""FAST k_doublet_sym: product-of-terms form against the per-term form and against STRICT (experiment; python tools/probe_product.py [cfg] [barcodes])."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
os.environ["DMX_EXPERIMENTS"] = "1"
import torch
import bench
from demuxlet_amd import build, capi, engine, synth, synth_torch as st
from golden_util import printed_mask
build.build(); capi.load()
cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = dict(bench.CONFIGS[cfg_id]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S, V, A = cfg["S"], cfg["V"], len(cfg["alphas"])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0xD3A00000 + cfg_id)
raw, g = bench.genotype_matrix(engine, synth, rng, S, V, cfg["field"])
dosage = torch.from_numpy(np.clip(raw.alleles, 0, 1).sum(axis=2).astype(np.float32)).to(dev)
dp = st.make_device_pileup(dosage, B, cfg["delta"], cfg["rbar"], seed=0xD3A0 + 1000 * cfg_id, device=dev)
def run(mode, env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    e = engine.Engine(V, cfg["alphas"], 0.5, device=0, mode=mode)
    for k in (env or {}): os.environ.pop(k)
    e.set_genotypes(g); e.set_pileup_struct(dp.as_struct(), keep=dp); e.run_singlet(); e.run_doublet(); e.sync()
    v = e.device_view()
    return st.tensor_from_ptr(v.llksAB, (B, V, V, A), torch.float64, dev).cpu().numpy().copy(), e.kernel_names()
gs, _ = run(capi.DMX_MODE_STRICT)
gp, kn = run(capi.DMX_MODE_FAST)
gt, _ = run(capi.DMX_MODE_FAST, {"DMX_SYM_NO_PRODUCT": "1"})
print(kn.doublet if hasattr(kn, "doublet") else kn)
m = np.broadcast_to(printed_mask(V, A)[None], gs.shape)
for name, x in (("product", gp), ("per-term", gt)):
    d = (x - gs)[m]
    print(name, "max|d| %.3e mean d %.3e rms %.3e" % (np.abs(d).max(), d.mean(), np.sqrt((d * d).mean())), "grid magnitude %.3e" % np.abs(gs[m]).max())
    i = np.argmax(np.abs((x - gs) * m)); print("  worst at", np.unravel_index(i, gs.shape), (x - gs).flat[i], gs.flat[i])
