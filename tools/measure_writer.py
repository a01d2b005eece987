"""Throughput of the .pair/.sing2/.best writer (host threads, ordered output): python tools/measure_writer.py <barcodes> <samples>"""
import sys, time, os
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from demuxlet_amd import engine, capi
import ctypes as C
B, V, A = int(sys.argv[1]), int(sys.argv[2]), 2
rng = np.random.default_rng(1)
grid = -rng.uniform(1000, 2000, size=(B, V, V, A))
l00 = -rng.uniform(1000, 2000, size=(B, A))
z = np.full(B, 100, np.int32)
fa = engine.FinalArgs([f"BC{i:08d}-1" for i in range(B)], [f"SAMPLE{j:03d}" for j in range(V)], (0.0, 0.5), 0.5, z, z, z, z, write_pair=True)
fin, keep = engine._final_struct(fa, grid=grid, l00=l00)
L = capi.load()
for t in (1, 4, 16):
    os.environ["DMX_THREADS"] = str(t)
    t0 = time.perf_counter()
    L.dmx_write_doublet(C.byref(fin), b"/tmp/wb_out")
    dt = time.perf_counter() - t0
    rows = B * (V + V * (V - 1) // 2)
    print(f"threads {t}: {rows} pair rows in {dt:.2f} s = {rows/dt:.3e} rows/s")
