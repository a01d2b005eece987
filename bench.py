#!/usr/bin/env python3
"""bench.py — throughput of the likelihood engine on synthetic pileups of the BASELINE.json configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|5] [--cells B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one resident batch: config 2 (the N=1 headline, BASELINE.json configs[1]):
K1 singlet accumulation over 10k barcodes x 50k SNPs x 8 samples (dense, GT field); configs 3/5 add the doublet grid
(K2) and the per-cell reduction (K3).  Inputs are generated in HBM before the timed region.  With N>1 every rank owns
its own 10k-barcode shard (weak scaling: barcodes are independent, cmd_cram_demuxlet.cpp:576) and each step ends with
the one RCCL gather of the per-cell records to rank 0.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

CONFIGS = {
    # name: B, S, V, field, alphas, delta, rbar, doublet
    2: dict(B=10_000, S=50_000, V=8, field="GT", alphas=(0.0, 0.5), delta=1.0, rbar=1.25, doublet=False,
            name="cfg2: 10k barcodes x 50k SNPs x 8 samples, GT field, singlet-only, dense (delta=1, rbar=1.25)"),
    3: dict(B=10_000, S=50_000, V=32, field="GP", alphas=(0.0, 0.5), delta=1.0, rbar=1.25, doublet=True,
            name="cfg3: 10k barcodes x 50k SNPs x 32 samples, GP field, doublet grid alpha 0,0.5, dense"),
    4: dict(B=12_500, S=100_000, V=64, field="GT", alphas=(0.0, 0.5), delta=1.0, rbar=1.25, doublet=True,
            name="cfg4 (per-GPU shard): 12.5k of 100k barcodes x 100k SNPs x 64 samples, GT field, doublet grid, dense"),
    5: dict(B=20_000, S=200_000, V=16, field="PL", alphas=(0.0, 0.5), delta=0.05, rbar=2.0, doublet=True,
            name="cfg5: 20k barcodes x 200k SNPs x 16 samples, PL field, doublet grid, sparse (delta=0.05, rbar=2)"),
    6: dict(B=20_000, S=100_000, V=16, field="GT", alphas=(0.0, 0.5), delta=0.02, rbar=1.25, doublet=True,
            name="cfg6 (not in BASELINE.json; SURVEY 8d's realistic 10x shape): 20k barcodes x 100k SNPs x 16 samples, GT, doublet "
                 "grid, sparse (delta=0.02 -> ~2000 covered SNPs per barcode, rbar=1.25)"),
}
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6  # FP64 vector peak (spec; half the 157.3 TF FP32 vector rate), FMA = 2 flop
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4   # FP64 wave64 instructions/s: 1024 SIMDs, 16 FP64 lanes/clk each (= 78.6 TF / 128)


def genotype_matrix(engine, synth, rng, S, V, field):
    raw = synth.make_raw_genotypes(rng, S, V)
    g = np.empty((S, V, 3), dtype=np.float32)
    if field == "GT":
        for s in range(S):
            g[s] = engine.geno_from_gt(raw.alleles[s], 0.01)
    elif field == "GP":
        gp = synth.raw_gp_from_alleles(rng, raw.alleles)
        for s in range(S):
            g[s] = engine.geno_from_gp(gp[s], 0.01)
    else:
        pl = synth.raw_pl_from_alleles(rng, raw.alleles)
        for s in range(S):
            g[s] = engine.geno_from_pl(pl[s])
    return raw, g


def cpu_baseline(dp, g, cfg, target_s=12.0):
    """The oracle (CPU restatement of the reference, 1 thread) on the first cells of the SAME workload."""
    from oracle import oracle_py as O
    V = cfg["V"]

    def prepare(c0, ncells):
        h = dp.host_slice(c0, ncells)
        words = ((h["reads"] >> 7).astype(np.uint32) << 24) | ((h["reads"] & 0x7F).astype(np.uint32) << 16) | 1
        pair_snp = h["pair_snp"] if h["pair_snp"] is not None else np.tile(np.arange(dp.n_snps, dtype=np.int32), ncells)
        csr = O.Csr([f"c{i:07d}" for i in range(ncells)], h["cell_pair_off"], pair_snp,
                    np.concatenate([[0], np.cumsum(h["pair_nrd"].astype(np.int64))]), words,
                    np.zeros(ncells, np.int32), np.zeros(ncells, np.int32), np.zeros(ncells, np.int32))
        return O.CsrPlan(csr, [f"s{j}" for j in range(V)], g, O.Params(tuple(cfg["alphas"]), 0.5), None,
                         singlet_only=not cfg["doublet"], want_grid=False)

    def execute(plan, repeats=1):
        t0 = time.perf_counter()
        for _ in range(repeats):
            plan.execute()
        return time.perf_counter() - t0, plan.n_pairs * repeats

    def run(ncells):
        return execute(prepare(0, ncells))

    # size the sample from the oracle's measured cost on this class of host (~6 ns per singlet term, ~19 ns per doublet
    # pair-evaluation) so that ONE run lands near target_s
    A = len(cfg["alphas"])
    ns_per_pair = 6.0 * (V + 1) + (19.0 * (V * V * A + A) if cfg["doublet"] else 0.0)
    pairs_per_cell = max(1.0, dp.n_pairs / max(dp.n_cells, 1))
    n = int(max(1, min(dp.n_cells, round(target_s / (1e-9 * ns_per_pair * pairs_per_cell)))))
    t, pairs = run(n)
    out = dict(value=pairs * V / t, unit="cell-SNP-sample triples/s", cores=1, kind="port",
               sample=f"first {n} barcodes of the same workload ({pairs} covered pairs), oracle/dmx_oracle.c "
                      f"(gcc -O2 -ffp-contract=off), {t:.1f} s wall", seconds=t)
    # the same code on all host cores at once (the reference itself is single-threaded, cmd_cram_demuxlet.cpp has no
    # parallelism; this is what `--group-list` sharding over cores would buy): one thread per core, each thread its own
    # barcodes of the same workload, host memory bounded to ~6 GB
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                    # a container's CPU quota is the real core count (cgroup v2: "quota period")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    if cores > 1:
        import threading
        bytes_per_cell = pairs_per_cell * 24.0
        n_mt = int(max(1, min(n, dp.n_cells // cores, 32e9 / (cores * bytes_per_cell))))
        reps = int(max(1, round(0.5 * n / n_mt)))                # each thread: about half the serial leg's work
        prep = [prepare(i * n_mt, n_mt) for i in range(cores)]
        res = [None] * cores
        gate = threading.Barrier(cores + 1)

        def work(i):
            gate.wait()
            res[i] = execute(prep[i], reps)

        th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        for x in th: x.start()
        gate.wait()
        t0 = time.perf_counter()
        for x in th: x.join()
        wall = time.perf_counter() - t0
        tot = sum(r[1] for r in res)
        out["all_cores"] = dict(value=tot * V / wall, unit="cell-SNP-sample triples/s", cores=cores,
                                sample=f"{cores} threads x {n_mt} barcodes x {reps} passes ({tot} covered pairs), {wall:.1f} s wall")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--cells", type=int, default=0, help="override barcodes per GPU (smaller = quicker run; not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--samples", type=int, default=0, help="override the number of samples (experiments; not the headline)")
    ap.add_argument("--field", default="", help="override the genotype field GT|GP|PL (experiments; not the headline)")
    ap.add_argument("--fast", action="store_true", help="DMX_MODE_FAST (bilinear doublet terms; opt-in, not the headline)")
    ap.add_argument("--alphas", default="", help="override the alpha grid, comma separated (experiments; not the headline)")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON record.  Libraries that print banners from C (RCCL's version block is written
    # to fd 1 and flushed at exit, i.e. AFTER anything Python printed) are sent to stderr: fd 1 is re-pointed at fd 2 for the
    # whole run and the record goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from demuxlet_amd import build, engine, synth, synth_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # DMX_BENCH_FORCE_DIST=1 runs the collective code path with a 1-rank RCCL group (1-GPU boxes: exercises the gather)
    use_dist = world > 1 or bool(os.environ.get("DMX_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
    if rank == 0:
        build.build()
    if use_dist:
        dist.barrier()

    cfg = dict(CONFIGS[args.config])
    if args.cells:
        cfg["B"] = args.cells
    if args.samples:
        cfg["V"] = args.samples
    if args.field:
        cfg["field"] = args.field
    if args.alphas:
        cfg["alphas"] = tuple(float(x) for x in args.alphas.split(","))
    if args.samples or args.field or args.alphas:
        cfg["name"] += f" [override: V={cfg['V']}, field={cfg['field']}, alphas={list(cfg['alphas'])}]"
    B, S, V, A = cfg["B"], cfg["S"], cfg["V"], len(cfg["alphas"])
    rng = np.random.default_rng(0xD3A00000 + args.config)       # the panel is shared by all ranks
    raw, g = genotype_matrix(engine, synth, rng, S, V, cfg["field"])
    dosage = torch.from_numpy(np.clip(raw.alleles, 0, 1).sum(axis=2).astype(np.float32)).to(dev)
    dp = synth_torch.make_device_pileup(dosage, B, cfg["delta"], cfg["rbar"], seed=0xD3A0 + 1000 * args.config + rank,
                                        device=dev, cell_id_base=rank * B)
    torch.cuda.synchronize()

    # one explicit HIP stream for everything timed: the engine launches on it, torch events are recorded on it and RCCL
    # orders against it (the legacy NULL stream would not do: dmx_engine_set_stream(NULL) means "the engine's own")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = engine.Engine(V, cfg["alphas"], 0.5, device=local, mode=engine.capi.DMX_MODE_FAST if args.fast else engine.capi.DMX_MODE_STRICT)
    if args.fast:
        cfg["name"] += " [DMX_MODE_FAST]"
    assert stream.cuda_stream != 0
    eng.set_stream(stream.cuda_stream)
    eng.set_genotypes(g)
    eng.set_pileup_struct(dp.as_struct(), keep=dp)
    nbytes = eng.algorithmic_bytes()

    # device views of the per-cell records, for the end-of-step gather (same row layout as demuxlet_amd/dist.py)
    def record_matrix():
        v = eng.device_view()
        cols = [synth_torch.tensor_from_ptr(v.llks, (B, V), torch.float64, dev),
                synth_torch.tensor_from_ptr(v.llk0s, (B, 1), torch.float64, dev)]
        if cfg["doublet"]:
            cols += [synth_torch.tensor_from_ptr(v.sing, (B, V), torch.float64, dev),
                     synth_torch.tensor_from_ptr(v.llks00, (B, A), torch.float64, dev),
                     synth_torch.tensor_from_ptr(v.summary, (B, engine.capi.SUMMARY_DTYPE.itemsize // 8), torch.float64, dev)]
        return torch.cat(cols, dim=1)

    gathered = None

    def step(ev=None):
        nonlocal gathered
        if ev: ev[0].record()
        eng.run_singlet()
        if ev: ev[1].record()
        if cfg["doublet"]:
            eng.run_doublet()
            if ev: ev[2].record()
        if use_dist:
            # THE collective of the job: one fixed-size record per barcode -> rank 0 (RCCL gather over xGMI)
            rec = record_matrix()
            outs = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
            dist.gather(rec, outs, dst=0)
            gathered = outs

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(evs[i])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        counts = torch.tensor([dp.n_pairs], dtype=torch.float64, device=dev)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        total_pairs = float(counts.item())
    else:
        total_pairs = float(dp.n_pairs)

    k1_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    k2_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs])) if cfg["doublet"] else 0.0
    if rank == 0:
        triples = total_pairs * V
        ms_per_step = 1e3 * elapsed / args.steps
        dom_ms, dom_bytes, dom_name = (k2_ms, nbytes.doublet_bytes, "k_doublet") if cfg["doublet"] else (k1_ms, nbytes.singlet_bytes, "k_singlet")
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic and VALU instruction counts of one launch of the dominant kernel are properties of the workload; they
        # come from the committed rocprofv3 PMC passes of the same workload (profiles/, tools/profile_round.sh)
        traffic, valu = None, None
        prof = ROOT / "profiles" / f"pmc_cfg{args.config}.json"
        if prof.exists():
            pj = json.loads(prof.read_text())
            if pj.get("barcodes_per_gpu") == B:
                traffic = pj.get("hbm_bytes_per_launch")
                if pj.get("valu_wave_insts_per_launch"):
                    rate = pj["valu_wave_insts_per_launch"] / (dom_ms * 1e-3)
                    valu = {"bound": "fp64_valu", "achieved": rate, "peak": VALU_PEAK_WAVE_INSTS, "unit": "wave-instructions/s",
                            "frac": rate / VALU_PEAK_WAVE_INSTS, "kernel": pj.get("kernel"),
                            "note": "the binding roofline of this path: FP64 VALU issue (1024 SIMDs x 2.4 GHz / 4 cycles per wave64 FP64 "
                                    "instruction); instruction count per launch from profiles/ PMC (SQ_INSTS_VALU), time live"}
        logs = dp.n_pairs * ((V + 1) + (V * V * A + A if cfg["doublet"] else 0))
        out = {
            "metric": "cell-SNP-sample triples/sec (singlet+doublet llk); HBM GB/s vs roofline",
            "value": triples * args.steps / elapsed, "unit": "triples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"], "barcodes_per_gpu": B, "snps": S, "samples": V, "alphas": list(cfg["alphas"]),
                       "covered_pairs_per_gpu": dp.n_pairs, "reads_per_gpu": dp.n_reads, "mode": "fast" if args.fast else "strict",
                       "sharding": f"barcodes x{world}, one RCCL gather per step" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes,
                         "kernel_ms": dom_ms,
                         "note": "FP64-VALU/log-bound path (SURVEY §8d): HBM fraction is reported as the metric asks; see roofline_valu. "
                                 "traffic = PMC FETCH_SIZE x2 + WRITE_SIZE of one launch (profiles/); above the algorithmic bytes it "
                                 "contains the L2 misses of the GL seed-table gathers (DESIGN.md §6), not re-reads of the inputs"},
            "roofline_valu": valu,
            "fp64_valu": {"logical_log_terms_per_s": logs / ((k1_ms + k2_ms) * 1e-3), "kernel_ms": {"k_singlet": k1_ms, "k_doublet+k_reduce": k2_ms},
                          "peak_tflops": FP64_VALU_PEAK_TFLOPS},
        }
        if world == 1:
            # the device's log() ceiling from a register-resident microkernel, same run: what fraction of it the
            # path's LOGICAL log terms amount to (the genotype-class kernels execute fewer logs than the reference's count,
            # so this fraction can exceed 1 for GT inputs; it cannot for GP/PL inputs)
            import ctypes
            rates = []
            for which in (0, 1):
                r = ctypes.c_double(0.0)
                engine.check(engine.capi.load().dmx_debug_log_rate(which, 4096, local, ctypes.byref(r)))
                rates.append(r.value)
            out["log_microkernel"] = {"dmx_log_per_s": rates[0], "ocml_log_per_s": rates[1],
                                      "path_logical_logs_over_dmx_log_ceiling": out["fp64_valu"]["logical_log_terms_per_s"] / rates[0]}
            # ALGORITHMIC FP64 rate against the 78.6 TF vector peak, log() costed as 1 op and as C_log ops, where
            # C_log = (peak wave-instructions/s x 64 lanes) / measured dmx_log/s issue slots, 2 flop each
            rbar = dp.n_reads / max(dp.n_pairs, 1)
            ops1 = dp.n_pairs * ((rbar * 11 + 8 + V * 7 + 7) + ((rbar * A * 54 + A * 18 + V * V * A * 20 + A * 20) if cfg["doublet"] else 0))
            c_log = VALU_PEAK_WAVE_INSTS * 64 / rates[0] * 2
            secs = (k1_ms + k2_ms) * 1e-3
            out["fp64_valu"].update({"algorithmic_tflops_log_as_1_op": ops1 / secs / 1e12, "c_log_flops": c_log,
                                     "algorithmic_tflops_log_as_c_log": (ops1 + logs * (c_log - 1)) / secs / 1e12,
                                     "frac_of_peak_log_as_c_log": (ops1 + logs * (c_log - 1)) / secs / 1e12 / FP64_VALU_PEAK_TFLOPS})
        if cfg["doublet"]:
            out["pair_evals_per_s"] = total_pairs * V * V * A * args.steps / elapsed
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is a rank-0, N=1 leg only
            out["cpu_baseline"] = cpu_baseline(dp, g, cfg)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        if rank == 0 and gathered is not None:
            assert len(gathered) == world and tuple(gathered[0].shape) == tuple(record_matrix().shape)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
