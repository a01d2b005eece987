"""Replays tools/fuzz_parity.py's generator to case N of a seed and stores that problem's inputs as a fixture (no GPU needed).
    python tools/extract_fuzz_case.py <seed> <case> <out.npz> [scale]
Used for tests/golden/fuzz_9334_case368.npz: the case the round-3 STRICT sweep stopped at (profiles/r03_fuzz_summary.txt)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
from demuxlet_amd import build
build.build()
from fuzz_gen import gen_case

seed, case, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(seed)
for _ in range(case + 1):
    cs = gen_case(rng, scale)
sp = cs["sp"]
print({k: cs[k] for k in ("V", "A", "alphas", "field", "dense", "S", "B", "rbar")})
np.savez_compressed(out, g=cs["g"], alphas=np.array(cs["alphas"]), n_cells=sp.n_cells, n_snps=sp.n_snps, cell_pair_off=sp.cell_pair_off,
                    cell_read_off=sp.cell_read_off, pair_snp=sp.pair_snp if sp.pair_snp is not None else np.zeros(0, np.int32),
                    dense=sp.pair_snp is None, pair_nrd=sp.pair_nrd, reads=sp.reads, rd_totl=sp.rd_totl, rd_pass=sp.rd_pass, rd_uniq=sp.rd_uniq,
                    seed=seed, case=case)
