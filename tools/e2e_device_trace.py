import os, sys, json, tempfile
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import bench
from demuxlet_amd import engine, synth, synth_torch
cfg = dict(bench.CONFIGS[3]); B,S,V = cfg["B"],cfg["S"],cfg["V"]
dev = torch.device("cuda",0)
rng = np.random.default_rng(0xD3A00003)
raw, g = bench.genotype_matrix(engine, synth, rng, S, V, cfg["field"])
dosage = torch.from_numpy(np.clip(raw.alleles,0,1).sum(axis=2).astype(np.float32)).to(dev)
dp = synth_torch.make_device_pileup(dosage, B, 1.0, 1.25, seed=0xD3A0+3000, device=dev)
h = dp.host_slice(0,B); nreads = np.diff(h["cell_read_off"]).astype(np.int32)
bcs=[f"BC{i:07d}-1" for i in range(B)]; sms=[f"SM{j:02d}" for j in range(V)]
ds = dp.as_struct(); ds.rd_totl = ds.rd_pass = ds.rd_uniq = nreads.ctypes.data
engine.device_warm_up(0,1)
with tempfile.TemporaryDirectory() as td:
    for rep in range(2):
        for nm, md in (("fast", engine.capi.DMX_MODE_FAST),("strict", engine.capi.DMX_MODE_STRICT)):
            for rpg in (None, "1", "2"):
                if rpg: os.environ["DMX_RANGES_PER_GPU"]=rpg
                else: os.environ.pop("DMX_RANGES_PER_GPU",None)
                if rep==1 and nm=="fast" and rpg is None: os.environ["DMX_E2E_TRACE"]="1"
                else: os.environ.pop("DMX_E2E_TRACE",None)
                tm = engine.demuxlet_run(ds, g, sms, cfg["alphas"], os.path.join(td,"o"), barcodes=bcs, timing=True, mode=md)
                print(rep, nm, "ranges/gpu", rpg, {k: round(v,4) if isinstance(v,float) else v for k,v in tm.items()}, flush=True)
