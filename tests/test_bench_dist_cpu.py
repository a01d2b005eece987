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
