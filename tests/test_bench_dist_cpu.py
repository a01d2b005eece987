"""CPU (gloo, world_size 2 and 3): the bookkeeping of bench.py's N > 1 line — the contiguous barcode ranges, the padded gather of
the per-barcode records to rank 0 and the max-over-ranks timing — with an injected compute (the record matrices are made here;
on GPUs they are views of the engine's device buffers and the backend is RCCL).  Covers what the driver's `--gpus N` run relies on
before it first runs it."""
import os
import socket
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B_total, ncols, outdir):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    lo, hi = bench.shard_range(B_total, world, rank)
    counts = [bench.shard_range(B_total, world, r)[1] - bench.shard_range(B_total, world, r)[0] for r in range(world)]
    assert sum(counts) == B_total and max(counts) - min(counts) <= 1 and counts[rank] == hi - lo
    # the injected compute: record row of barcode b = [b, b + 0.5, ...] so that rank 0 can tell every row's origin
    rec = (torch.arange(lo, hi, dtype=torch.float64)[:, None] + 0.5 * torch.arange(ncols, dtype=torch.float64)[None, :]).contiguous()
    outs = None
    for _ in range(3):                                  # every step gathers again (same shapes, fresh buffers)
        outs = bench.gather_records(torch, dist, rec, counts, rank, world)
    elapsed, total_pairs, per_rank = bench.collect_timing(torch, dist, dev, elapsed=1.0 + 0.25 * rank, own_elapsed=0.5 + 0.1 * rank,
                                                          n_pairs=(hi - lo) * 7, steps=2, world=world)
    assert elapsed == pytest.approx(1.0 + 0.25 * (world - 1))          # the slowest rank's time is the job's
    assert total_pairs == B_total * 7
    assert per_rank == pytest.approx([1e3 * (0.5 + 0.1 * r) / 2 for r in range(world)])
    if rank == 0:
        assert len(outs) == world and all(o.shape == (max(counts), ncols) for o in outs)
        b = 0
        for r in range(world):
            real = outs[r][:counts[r]]
            assert torch.equal(real[:, 0], torch.arange(b, b + counts[r], dtype=torch.float64))     # rank order == barcode order
            assert torch.equal(real[:, ncols - 1], real[:, 0] + 0.5 * (ncols - 1))
            assert (outs[r][counts[r]:] == 0).all()                                                 # the padding rows
            b += counts[r]
        assert b == B_total
        Path(outdir, "ok").write_text("ok")
    else:
        assert outs is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B_total", [(2, 100_000), (2, 12_501), (3, 10), (3, 100_000)])
def test_bench_shard_gather_and_timing(world, B_total, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), B_total, 9, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def _run_bench(*argv, env_extra=None, timeout=300):
    import json
    import subprocess
    env = dict(os.environ, DMX_BENCH_INJECT="1", **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if not (env_extra and k in env_extra):
            env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_bench_gpus_n_launches_n_ranks_itself(n, tmp_path):
    """`python bench.py --gpus N` with no launcher in the environment (how the driver runs N = 1) starts N ranks itself and the line says
    n_gpus = ranks_seen = N.  Injected compute over gloo: the launch, the checks, the ranges, the gather, the timing
    and the compact line are the product's; only the engine is replaced."""
    import json
    full = tmp_path / "full.json"
    r = _run_bench("--gpus", str(n), "--steps", "3", "--warmup", "1", "--cells", "1001", env_extra={"DMX_BENCH_FULL": str(full)})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-1000:]
    assert len(lines[0]) < 6144
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["ranks_seen"] == n and len(d["per_rank_ms_per_step"]) == n and d["scaling"] == "strong"
    assert d["config"]["barcodes_total"] == 1001 and d["config"]["workload"].startswith("cfg4") and d["steps"] == 3 and d["warmup"] == 1
    assert "INJECTED" in d["data"]
    assert json.loads(full.read_text())["n_gpus"] == n


def test_bench_gpus_must_match_the_launchers_world_size():
    r = _run_bench("--gpus", "2", env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "WORLD_SIZE=3" in r.stderr
    assert r.stdout.strip() == ""


def test_bench_refuses_more_gpus_than_visible():
    """Without the test hook: this container has no GPU, so any --gpus N > 1 must fail loudly before launching anything."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DMX_BENCH_INJECT")}
    import torch
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n + 1 if n else 2), "--cells", "96"], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=300)
    assert r.returncode != 0 and "GPUs requested" in r.stderr and f"{n} visible" in r.stderr
    assert r.stdout.strip() == ""


def test_compact_line_stays_small_on_a_full_size_record():
    """The r03 full record (27 KB, which the driver could not parse) through compact_line: < 6 KB, driver keys + roofline + cpu_baseline kept."""
    import json
    import bench
    full = json.loads((ROOT / "profiles" / "r03_bench_default_n1.json").read_text())
    for a in [full] + full.get("also", []):                 # r03 records have no tag: derive it the way run_config does now
        a["config"].setdefault("tag", a["config"]["workload"][:4] + "/" + a["config"]["mode"])
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < 6144, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "also"):
        assert k in d, k
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["cpu_baseline"]["cores"] == 1
    assert len(d["also"]) == len(full["also"]) and all(len(json.dumps(a)) < 400 for a in d["also"])
