import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
t0 = time.time()
from demuxlet_amd import engine as eng, capi
capi.load()
t1 = time.time()
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t2 = time.time()
V, S = 32, 50000
g = np.random.default_rng(1).random((S, V, 3), dtype=np.float32)
ts = []
for i in range(3):
    a = time.time(); e = eng.Engine(V, (0.0, 0.5), 0.5, mode=capi.DMX_MODE_FAST); b = time.time(); e.set_genotypes(g); c = time.time()
    ts.append((b - a, c - b)); e.close()
print("load", t1 - t0, "torch init", t2 - t1, "create/set_genotypes:", ts)
