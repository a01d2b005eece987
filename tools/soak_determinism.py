"""Runs engine configurations repeatedly on the GPU and checks that every run returns the same bits: a race in the LDS-direct
double buffers, the pipelined class kernels or the certificate would show up as a rare difference.
python tools/soak_determinism.py [runs]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import bench
from demuxlet_amd import engine, synth, synth_torch

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
for cfgno, cells, fast in ((3, 3000, True), (3, 1500, False), (4, 1500, True), (4, 1000, False), (5, 6000, True), (5, 4000, False), (2, 10000, False)):
    cfg = dict(bench.CONFIGS[cfgno]); B = cells
    rng = np.random.default_rng(0xD3A00000 + cfgno)
    raw, g = bench.genotype_matrix(engine, synth, rng, cfg["S"], cfg["V"], cfg["field"])
    dosage = torch.from_numpy(np.clip(raw.alleles, 0, 1).sum(axis=2).astype(np.float32)).to(dev)
    dp = synth_torch.make_device_pileup(dosage, B, cfg["delta"], cfg["rbar"], seed=77 + cfgno, device=dev, cell_id_base=0)
    del dosage
    torch.cuda.synchronize()
    e = engine.Engine(cfg["V"], cfg["alphas"], 0.5, mode=engine.capi.DMX_MODE_FAST if fast else engine.capi.DMX_MODE_STRICT)
    e.set_genotypes(g)
    e.set_pileup_struct(dp.as_struct(), keep=dp)
    ref = None
    for r in range(runs):
        e.run_singlet()
        h = [x.tobytes() for x in e.get_singlet()]
        if cfg["doublet"]:
            e.run_doublet()
            grid, l00, summ = e.get_doublet()
            h += [grid.tobytes(), l00.tobytes(), summ.tobytes()]
        if ref is None:
            ref = h
        elif h != ref:
            print(f"cfg{cfgno} cells={cells} fast={fast}: run {r} DIFFERS from run 0")
            sys.exit(1)
    print(f"cfg{cfgno} cells={cells} fast={fast}: {runs} runs bit-identical")
    e.close()
    del dp
print("soak ok")
