"""PCIe-inclusive timing of the boundary with HOST buffers (DESIGN.md "PCIe-inclusive rate"): cfg2-shaped pileup in pageable
host memory -> dmx_engine_set_pileup(DMX_MEM_HOST) -> run_singlet -> get_singlet.  Run on a GPU box: python tools/measure_host_path.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from demuxlet_amd import build, engine, synth, synth_torch
import bench

build.build()
cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = dict(bench.CONFIGS[cfg_id])
if len(sys.argv) > 2:
    cfg["B"] = int(sys.argv[2])
B, S, V = cfg["B"], cfg["S"], cfg["V"]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0xD3A00000 + cfg_id)
raw, g = bench.genotype_matrix(engine, synth, rng, S, V, cfg["field"])
dosage = torch.from_numpy(np.clip(raw.alleles, 0, 1).sum(axis=2).astype(np.float32)).to(dev)
dp = synth_torch.make_device_pileup(dosage, B, cfg["delta"], cfg["rbar"], seed=0xD3A0 + 1000 * cfg_id, device=dev)
h = dp.host_slice(0, B)
z = np.zeros(B, np.int32)
pl = engine.HostPileup(B, S, h["cell_pair_off"], h["cell_read_off"], h["pair_snp"], h["pair_nrd"], h["reads"], z, z, z)
nbytes = pl.pair_nrd.nbytes + pl.reads.nbytes + (0 if pl.pair_snp is None else pl.pair_snp.nbytes) + 16 * (B + 1)
del dp
torch.cuda.empty_cache()
e = engine.Engine(V, cfg["alphas"], 0.5, device=0)
e.set_genotypes(g)
for it in range(3):
    t0 = time.perf_counter(); e.set_pileup(pl); t1 = time.perf_counter()
    e.run_singlet(); e.sync(); t2 = time.perf_counter()
    if cfg["doublet"]:
        e.run_doublet(); e.sync()
    t3 = time.perf_counter()
    llks, llk0s = e.get_singlet(); t4 = time.perf_counter()
    print(f"pass {it}: set_pileup(HOST) {1e3*(t1-t0):.1f} ms ({nbytes/1e9/(t1-t0):.1f} GB/s of {nbytes/1e9:.2f} GB), singlet {1e3*(t2-t1):.1f} ms, "
          f"doublet {1e3*(t3-t2):.1f} ms, get {1e3*(t4-t3):.1f} ms; end to end {B*S*V*(1 if cfg['delta']>=1 else cfg['delta'])/(t4-t0):.3e} triples/s")
