"""GPU (-m gpu): bench.py's contract — stdout is exactly one JSON line with the driver's keys, the roofline and CPU-baseline
objects, and sane values — on a reduced barcode count so that the test takes seconds."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def run_bench(*extra):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", *extra],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-500:]          # C-level banners (RCCL, ...) must not reach stdout
    return json.loads(lines[0])


def test_bench_line_singlet():
    d = run_bench("--cells", "1024")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 10 * c["value"] and d["ms_per_step"] > 0
    assert d["config"]["barcodes_per_gpu"] == 1024 and d["config"]["mode"] == "strict"


def test_bench_line_doublet_configs():
    d = run_bench("--config", "3", "--cells", "64", "--no-cpu-baseline")
    assert d["pair_evals_per_s"] > 0 and d["roofline"]["kernel"] == "k_doublet" and "cpu_baseline" not in d
    f = run_bench("--config", "3", "--cells", "64", "--no-cpu-baseline", "--fast")
    assert f["config"]["mode"] == "fast"
