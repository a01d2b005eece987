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
    assert len(lines[0]) < 6144, len(lines[0])       # the driver could not parse a 27 KB line
    assert "note" not in lines[0]                    # no prose on stdout
    d = json.loads(lines[0])
    full = json.loads((ROOT / d["full_record"]).read_text())      # the full record lies beside bench.py
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "fp64_valu" in full
    return d


def test_bench_line_singlet():
    d = run_bench("--config", "2", "--cells", "1024")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "none"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 * r["frac"] and r["achieved"] > 0      # (the line carries 6 significant digits)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 10 * c["value"] and d["ms_per_step"] > 0
    assert d["config"]["barcodes_per_gpu"] == 1024 and d["config"]["mode"] == "strict"
    # the timed path's rows against the oracle on the barcodes the CPU leg evaluates
    p = d["parity_check"]
    assert p["barcodes"] >= 1 and p["max_abs_delta"] <= 1e-9 and p["calls_identical"] is True and p["ok"] is True


def test_bench_line_doublet_configs():
    d = run_bench("--config", "3", "--cells", "64", "--no-cpu-baseline")
    assert d["pair_evals_per_s"] > 0 and d["roofline"]["kernel"] == "k_doublet" and "cpu_baseline" not in d and "parity_check" not in d
    f = run_bench("--config", "3", "--cells", "64", "--no-cpu-baseline", "--fast")
    assert f["config"]["mode"] == "fast"


def test_bench_parity_check_on_the_timed_path_doublet():
    """cfg3's depth and panel on 8 barcodes, STRICT and FAST: the line carries the comparison of dmx_engine_run's rows (the timed steps' own
    results) with the oracle — values within 1e-9 (FAST: the printed entries), K3's records making the oracle's calls."""
    for extra in ((), ("--fast",)):
        d = run_bench("--config", "3", "--cells", "8", "--only", *extra)
        p = d["parity_check"]
        assert 1 <= p["barcodes"] <= 8 and p["max_abs_delta"] <= 1e-9 and p["calls_identical"] is True and p["ok"] is True, p
        assert d["cpu_baseline"]["value"] > 0


def test_bench_default_line_is_cfg3_with_nested_records():
    """The driver's N=1 command (no --config): cfg3 STRICT as the line, cfg3-FAST / cfg2 / cfg5 as nested records."""
    d = run_bench("--cells", "64")
    assert d["config"]["workload"].startswith("cfg3") and d["config"]["mode"] == "strict" and d["pair_evals_per_s"] > 0
    assert d["roofline"]["kernel"] == "k_doublet" and d["scaling"] == "none"
    names = [a["workload"] for a in d["also"]]
    assert names == ["cfg3/fast", "cfg2/strict", "cfg5/strict", "cfg5/fast", "cfg6/strict", "cfg6/fast", "cfg4/strict", "cfg4/fast", "cfg4-shard/strict"]
    assert set(d["end_to_end"]["cfg6"]) >= {"strict", "fast"} and 0 <= d["end_to_end"]["cfg6"]["fast"]["grid_fetched_frac"] <= 1
    # the counter-derived fractions are quoted only for the kernel the committed counters were collected on (dmx_engine_kernel_names)
    assert d["roofline"]["kernel_launched"].startswith("k_doublet_a2")   # (cfg3 STRICT at full size: k_doublet_a2u; smaller panels: k_doublet_a2)
    assert d["parity_check"]["ok"] is True and d["parity_check"]["max_abs_delta"] <= 1e-9
    for a in d["also"]:
        if a["workload"] != "cfg4/strict":            # [barcodes, max |delta| vs the oracle, calls identical] of that record's own timed steps
            assert a["parity"][0] >= 1 and a["parity"][1] <= 1e-9 and a["parity"][2] is True, a
        assert a["value"] > 0 and a["roofline_frac"] > 0 and a["kernel_ms"] > 0 and a["ms_per_step"] >= a["kernel_ms"] * 0.999
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.001 and d["roofline"]["counts"].startswith("profiles/pmc_cfg3_strict")
    assert d["roofline_valu"]["kernel"].startswith("k_doublet")


def test_bench_sharded_path_with_a_one_rank_group():
    """The N>1 code path (cfg4 ranges + the end-of-step RCCL gather) on the one GPU this box has: a 1-rank process group."""
    import os
    env = dict(os.environ, DMX_BENCH_FORCE_DIST="1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "4",
                        "--cells", "96", "--e2e-write-pair"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert d["ranks_seen"] == 1 and len(d["per_rank_ms_per_step"]) == 1 and d["gather_ms"] >= 0
    assert d["config"]["workload"].startswith("cfg4") and d["config"]["barcodes_total"] == 96
    # an N > 1 line is judged like the N = 1 line: it carries its own CPU baseline (rank 0, after the timed region) and the ranks' devices
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 1 and len(d["rank_devices"]) == 1
    assert d["parity_check"]["ok"] is True
    # ... and the job with --write-pair on every rank's range at once (rows formatted on the GPUs): stage seconds, the slowest rank's total
    wp = d["end_to_end"]["cfg4_shard_write_pair"]
    assert wp["pair_rows"] == 96 * (64 + 64 * 63 // 2) and wp["strict"]["total_s"] > 0 and wp["slowest_rank_total_s"]["strict"] >= wp["strict"]["total_s"] * 0.999


def test_bench_gpus_n_without_a_launcher():
    """`python bench.py --gpus N` (no torchrun): more ranks than visible devices fails loudly before anything is launched; with enough
    devices it starts N ranks itself and the line says so (the N >= 2 leg needs a multi-GPU box)."""
    import os
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n + 1), "--cells", "96"], capture_output=True, text=True, cwd=ROOT,
                       timeout=600, env=env)
    assert r.returncode != 0 and f"{n + 1} GPUs requested, {n} visible" in r.stderr and r.stdout.strip() == ""
    if n >= 2:
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cells", "192", "--only"],
                           capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
        assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["barcodes_per_gpu"] == 96
