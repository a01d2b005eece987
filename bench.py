#!/usr/bin/env python3
"""bench.py — throughput of the likelihood engine on synthetic pileups of the BASELINE.json configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5|6] [--cells B] [--fast] [--only]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one resident batch: K1 singlet accumulation (cmd_cram_demuxlet.cpp:412-461), then —
for the doublet configs — K2 doublet grid (:576-710) and K3 per-cell reduction (:713-734,:746-758,:799-828).  Inputs are
generated in HBM before the timed region.

N = 1 (the driver's headline line): BASELINE.json configs[2] = cfg3, the heaviest single-GPU configuration and the one that
exercises the whole metric (singlet + doublet), in STRICT mode (the reference's operation order).  The same run appends, as
nested records under "also", cfg3 in FAST mode, cfg2 (singlet-only), cfg5 (sparse PL) and — the strong-scaling base of the
N > 1 line — ALL of cfg4 (100k barcodes x 100k SNPs x 64 samples, 1e10 covered pairs, 22.5 GB of pileup) on this one GPU in
both modes, and cfg4's 12 500-barcode shard (one GPU's share of the 8-GPU run), each timed the same way with fewer steps.  `--only` skips them.

Output contract: stdout carries exactly ONE compact JSON line (< 6 KB: the driver's keys, `config`, `roofline`, `roofline_valu`,
`cpu_baseline` and one short object per nested configuration under `also`; no prose).  The full record (every per-kernel time, the
executed-flop figures, end-to-end stage seconds, sample descriptions) is written to `bench_full.json` beside this file (path in
DMX_BENCH_FULL) and to stderr.

`--gpus N` is what decides the number of ranks.  Launched plainly (`python bench.py --gpus N`, no WORLD_SIZE in the environment) with
N > 1, this script checks that N GPUs are visible and re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1`; launched by torchrun, `--gpus` must equal WORLD_SIZE.  Every rank then checks that RCCL's
world size is N and that the N ranks sit on N distinct devices; anything else exits non-zero with a message.

N > 1: BASELINE.json configs[3] = cfg4 (100k barcodes x 100k SNPs x 64 samples, GT), STRONG scaling: the 100k barcodes are cut
into N contiguous equal ranges (barcodes are independent, cmd_cram_demuxlet.cpp:576; dense pileup = equal work), rank r
generates and owns range r, and every step ends with THE one collective of the job: the RCCL gather of the fixed-size
per-barcode records to rank 0.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import gc
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
    4: dict(B=100_000, S=100_000, V=64, field="GT", alphas=(0.0, 0.5), delta=1.0, rbar=1.25, doublet=True,
            name="cfg4: 100k barcodes x 100k SNPs x 64 samples, GT field, doublet grid alpha 0,0.5, dense"),
    5: dict(B=20_000, S=200_000, V=16, field="PL", alphas=(0.0, 0.5), delta=0.05, rbar=2.0, doublet=True,
            name="cfg5: 20k barcodes x 200k SNPs x 16 samples, PL field, doublet grid, sparse (delta=0.05, rbar=2)"),
    6: dict(B=20_000, S=100_000, V=16, field="GT", alphas=(0.0, 0.5), delta=0.02, rbar=1.25, doublet=True,
            name="cfg6 (not in BASELINE.json; SURVEY 8d's realistic 10x shape): 20k barcodes x 100k SNPs x 16 samples, GT, doublet "
                 "grid, sparse (delta=0.02 -> ~2000 covered SNPs per barcode, rbar=1.25)"),
}
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6  # FP64 vector peak (spec; half the 157.3 TF FP32 vector rate), FMA = 2 flop
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9
VALU_PEAK_WAVE_INSTS = SIMDS * CLOCK_HZ / 4   # FP64 wave64 instructions/s: 16 FP64 lanes/clk per SIMD (= 78.6 TF / 128)
LOG_FP64_INSTS = 10          # FP64 instructions of one evaluation of the doublet kernels' log (dmx_log2, csrc/dmx_log.hpp; the singlet
                             # kernels' 128-bin dmx_log has 11)
METRIC = "cell-SNP-sample triples/sec (singlet+doublet llk); HBM GB/s vs roofline"


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


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                    # a container's CPU quota is the real core count (cgroup v2: "quota period")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_sample_cells(dp, cfg, target_s):
    """How many of the workload's first barcodes the oracle gets: sized from its measured cost on this class of host (~6 ns per
    singlet term, ~19 ns per doublet pair-evaluation) so that ONE run lands near target_s."""
    V, A = cfg["V"], len(cfg["alphas"])
    ns_per_pair = 6.0 * (V + 1) + (19.0 * (V * V * A + A) if cfg["doublet"] else 0.0)
    pairs_per_cell = max(1.0, dp.n_pairs / max(dp.n_cells, 1))
    return int(max(1, min(dp.n_cells, round(target_s / (1e-9 * ns_per_pair * pairs_per_cell)))))


def engine_rows(cx, eng, cfg, n):
    """What the engine holds for its first n barcodes — called right after the LAST TIMED step, before anything else runs on the engine,
    so these ARE the timed path's results (dmx_engine_run: K1 beside / after K2, K3, K3b) at the timed size."""
    torch, st = cx.torch, cx.synth_torch
    B, V, A = eng.B, cfg["V"], len(cfg["alphas"])
    v = eng.device_view()
    rows = dict(llks=st.tensor_from_ptr(v.llks, (B, V), torch.float64, cx.dev)[:n].cpu().numpy(),
                llk0s=st.tensor_from_ptr(v.llk0s, (B,), torch.float64, cx.dev)[:n].cpu().numpy())
    if cfg["doublet"]:
        words = cx.engine.capi.SUMMARY_DTYPE.itemsize // 8
        rows["l00"] = st.tensor_from_ptr(v.llks00, (B, A), torch.float64, cx.dev)[:n].cpu().numpy()
        sm = st.tensor_from_ptr(v.summary, (B, words), torch.float64, cx.dev)[:n].cpu().numpy()
        rows["summ"] = np.ascontiguousarray(sm).view(cx.engine.capi.SUMMARY_DTYPE).reshape(-1)
        rows["grid"] = eng.get_cell_grids(np.arange(n, dtype=np.int32))
    return rows


def calls_of_grid(grid):
    """The calls cmd_cram_demuxlet.cpp:746-758 (best / next singlet: strict <, first maximum) and :799-814 (best doublet over j != k, alpha
    index >= 1) make from one barcode's llksAB[V][V][A]: (best singlet, next singlet, {j, k} of the best doublet, its alpha index)."""
    V, _, A = grid.shape
    i1 = i2 = -1; m1 = m2 = -1e300
    for j in range(V):
        x = grid[j, 0, 0]
        if m1 < x: m2, i2, i1, m1 = m1, i1, j, x
        elif m2 < x: i2, m2 = j, x
    jb = kb = nb = -1; mab = -1e300
    for j in range(V):
        for k in range(V):
            if j == k: continue
            for a in range(1, A):
                if mab < grid[j, k, a]: jb, kb, nb, mab = j, k, a, grid[j, k, a]
    return i1, i2, frozenset((jb, kb)), nb


def parity_check(rows, want, cfg, fast):
    """The engine's rows of the timed steps against the oracle's values for the same barcodes (the CPU leg evaluates them anyway): the
    largest absolute difference over llks, llk0s, llksAB (FAST: the entries demuxlet prints or decides on — the singlet column and every
    alpha >= 1 entry; the rest FAST does not compute) and llks00, and whether K3's records make the oracle's calls."""
    n = len(want.llk0s)
    d = max(float(np.abs(rows["llks"][:n] - want.llks).max()), float(np.abs(rows["llk0s"][:n] - want.llk0s).max()))
    if cfg["doublet"]:
        V, A = cfg["V"], len(cfg["alphas"])
        dg = np.abs(rows["grid"][:n] - want.llksAB)
        if fast:
            mask = np.zeros((V, V, A), dtype=bool); mask[:, 0, 0] = True; mask[:, :, 1:] = True
            dg = dg[:, mask]
        d = max(d, float(dg.max()), float(np.abs(rows["l00"][:n] - want.llks00).max()))
        from demuxlet_amd import capi
        near = capi.DMX_CELL_NEAR_DOUBLET | capi.DMX_CELL_NEAR_SINGLET
        same, flagged = True, 0
        for c in range(n):
            sm = rows["summ"][c]
            if int(sm["n_pairs"]) == 0:
                continue
            i1, i2, jk, nb = calls_of_grid(want.llksAB[c])
            same = same and (int(sm["i_sing1"]), int(sm["i_sing2"]), frozenset((int(sm["j_best"]), int(sm["k_best"]))), int(sm["n_best"])) == (i1, i2, jk, nb)
            flagged += int((int(sm["flags"]) & near) != 0)                # near-tie barcodes: the writers decide those from the grid (host arbiter)
        out = {"barcodes": n, "max_abs_delta": d, "calls_identical": bool(same), "near_tie_flagged": flagged}
    else:
        out = {"barcodes": n, "max_abs_delta": d, "calls_identical": bool(np.array_equal(np.argmax(rows["llks"][:n], axis=1), np.argmax(want.llks, axis=1)))}
    out.update(tolerance=1e-9, ok=bool(d <= 1e-9 and out["calls_identical"]),
               what="engine rows after the last timed step vs oracle/dmx_oracle.c on the same barcodes at the timed size"
                    + (" (FAST: printed entries)" if fast and cfg["doublet"] else ""))
    return out


def cpu_baseline(dp, g, cfg, target_s=12.0, rows=None, fast=False, legs=True):
    """The oracle (CPU restatement of the reference, 1 thread) on the first cells of the SAME workload.  With `rows` (engine_rows of
    the same barcodes) the oracle's values are compared with the engine's: `parity_check`.  legs=False: that comparison only."""
    from oracle import oracle_py as O
    V = cfg["V"]

    def prepare(c0, ncells, want_grid=False):
        h = dp.host_slice(c0, ncells)
        words = ((h["reads"] >> 7).astype(np.uint32) << 24) | ((h["reads"] & 0x7F).astype(np.uint32) << 16) | 1
        pair_snp = h["pair_snp"] if h["pair_snp"] is not None else np.tile(np.arange(dp.n_snps, dtype=np.int32), ncells)
        csr = O.Csr([f"c{i:07d}" for i in range(ncells)], h["cell_pair_off"], pair_snp,
                    np.concatenate([[0], np.cumsum(h["pair_nrd"].astype(np.int64))]), words,
                    np.zeros(ncells, np.int32), np.zeros(ncells, np.int32), np.zeros(ncells, np.int32))
        return O.CsrPlan(csr, [f"s{j}" for j in range(V)], g, O.Params(tuple(cfg["alphas"]), 0.5), None,
                         singlet_only=not cfg["doublet"], want_grid=want_grid)

    def execute(plan, repeats=1):
        t0 = time.perf_counter()
        for _ in range(repeats):
            plan.execute()
        return time.perf_counter() - t0, plan.n_pairs * repeats

    A = len(cfg["alphas"])
    pairs_per_cell = max(1.0, dp.n_pairs / max(dp.n_cells, 1))
    n = len(rows["llk0s"]) if rows is not None else cpu_sample_cells(dp, cfg, target_s)
    plan = prepare(0, n, want_grid=rows is not None and cfg["doublet"])
    t, pairs = execute(plan)
    # oracle/dmx_oracle.c, gcc -O2 -ffp-contract=off: a CSR walk of the reference's arithmetic — the reference's own std::map walk is
    # slower (see reference_slice below), so this baseline is conservative
    out = dict(value=pairs * V / t, unit="cell-SNP-sample triples/s", cores=1, kind="port",
               sample=f"first {n} barcodes of the same workload ({pairs} covered pairs), oracle/dmx_oracle.c on 1 thread, {t:.1f} s", seconds=t)
    if rows is not None:
        out["parity_check"] = parity_check(rows, plan.out, cfg, fast)
    del plan
    if not legs:
        return out
    if cfg["doublet"]:
        out["pair_evals_per_s"] = pairs * V * V * A / t
    ref = reference_slice_leg(dp, g, cfg)
    if ref:
        out["reference_slice"] = ref
    # the same code on all host cores at once (the reference itself is single-threaded, cmd_cram_demuxlet.cpp has no
    # parallelism; this is what `--group-list` sharding over cores would buy): one thread per core, each thread its own
    # barcodes of the same workload, host memory bounded to ~6 GB
    cores = host_cores()
    if cores > 1:
        import threading
        bytes_per_cell = pairs_per_cell * 24.0
        n_mt = int(max(1, min(n, dp.n_cells // cores, 32e9 / (cores * bytes_per_cell))))
        cores = max(1, min(cores, dp.n_cells // n_mt))           # (a workload of fewer barcodes than cores: one barcode per thread, fewer threads)
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


def reference_slice_leg(dp, g, cfg, n_cells=24, n_snps=2000):
    """the reference's OWN lines (cmd_cram_demuxlet.cpp:390-881 compiled verbatim into oracle/_ref/ref_slice_harness,
    std::map store, materialised pair tables, text writers) timed on a corner of the same workload: the first n_cells barcodes
    restricted to the first n_snps SNPs.  Only where the prebuilt harness travelled with the repo; it reads nothing else."""
    import subprocess
    import tempfile
    from oracle import oracle_py as O
    if not O.have_ref() or not cfg["doublet"]:
        return None
    V = cfg["V"]
    n_cells = min(n_cells, dp.n_cells)
    n_snps = min(n_snps, dp.n_snps)
    h = dp.host_slice(0, n_cells)
    po, ro = h["cell_pair_off"], h["cell_read_off"]
    bc, snp, umi, al, bq, nr = [], [], [], [], [], []
    pairs = 0
    for c in range(n_cells):
        r = int(ro[c])
        for p in range(int(po[c]), int(po[c + 1])):
            s = int(h["pair_snp"][p]) if h["pair_snp"] is not None else p - int(po[c])
            n = int(h["pair_nrd"][p])
            if s < n_snps:
                pairs += 1
                for q in range(n):
                    b = int(h["reads"][r + q])
                    bc.append(f"BC{c:05d}"); snp.append(s); umi.append(f"U{q:03d}"); al.append(b >> 7); bq.append(b & 127); nr.append(1)
            r += n
    ev = O.Events(bc, np.array(snp, dtype=np.int32), umi, np.array(al, dtype=np.uint8), np.array(bq, dtype=np.uint8), np.array(nr, dtype=np.uint8))
    pb = O.Problem([f"s{j}" for j in range(V)], g[:n_snps], ev, O.Params(tuple(cfg["alphas"]), 0.5))
    with tempfile.TemporaryDirectory() as td:
        spec = os.path.join(td, "spec.txt")
        O.write_spec(pb, spec)
        r = subprocess.run([str(O.REF_HARNESS), spec, os.path.join(td, "ref"), "--no-raw"], capture_output=True, text=True)
    secs = [float(l.split()[1]) for l in r.stderr.splitlines() if l.startswith("SLICE_SECONDS")]
    if r.returncode != 0 or not secs:
        return None
    A = len(cfg["alphas"])
    return dict(value=pairs * V / secs[0], unit="cell-SNP-sample triples/s", pair_evals_per_s=pairs * V * V * A / secs[0], cores=1, seconds=secs[0],
                kind="reference lines 390-881, verbatim, behind I/O stand-ins (oracle/_ref/ref_slice_harness; see DESIGN.md §3)",
                sample=f"first {n_cells} barcodes x first {n_snps} SNPs of the same workload ({pairs} covered pairs), incl. the reference's "
                       f"S*V^2*9 pair-table precompute and its four text files")


def write_pair_leg(cx, cfg, dp, g, B, S, V):
    """BASELINE config 4's defining option (`--write-pair`, cmd_cram_demuxlet.cpp:55 "(HUGE)", rows :772-797) on this workload through
    dmx_demuxlet_run from the device-resident pileup, STRICT and FAST: stage seconds, the `.pair` file's rows and bytes, rows per second of the
    writer stage.  Outside the timed region."""
    import tempfile
    engine = cx.engine
    nreads = np.diff(dp.cell_read_off.cpu().numpy()).astype(np.int32)
    bcs = [f"BC{i:07d}-1" for i in range(B)]
    sms = [f"SM{j:02d}" for j in range(V)]
    A = len(cfg["alphas"])
    n_half = sum(1 for a in cfg["alphas"][1:] if a == 0.5)
    rows = B * (V + V * (V - 1) * (A - 1 - n_half) + V * (V - 1) // 2 * n_half)
    out = {"what": f"dmx_demuxlet_run with write_pair on {B} barcodes x {S} SNPs x {V} samples from a device-resident pileup (1 GPU, tie arbiter on): "
                   "wall seconds per stage; write_s = the writer thread (arbiter + .single/.sing2/.best/.pair text + fwrite), beside the GPU",
           "pair_rows": rows, "host_cores": host_cores()}
    with tempfile.TemporaryDirectory() as td:
        out["tmp_fs"] = os.popen(f"stat -f -c %T {td}").read().strip()
        for name, md in (("strict", engine.capi.DMX_MODE_STRICT), ("fast", engine.capi.DMX_MODE_FAST)):
            ds = dp.as_struct()
            ds.rd_totl = ds.rd_pass = ds.rd_uniq = nreads.ctypes.data
            tm = engine.demuxlet_run(ds, g, sms, cfg["alphas"], os.path.join(td, name), barcodes=bcs, timing=True, mode=md, write_pair=True)
            tm.pop("reserved", None)
            nbytes = os.path.getsize(os.path.join(td, name + ".pair"))
            tm.update(pair_bytes=nbytes, pair_rows_per_s_of_write_s=rows / max(tm["write_s"], 1e-9), pair_GB_per_s_of_write_s=nbytes / max(tm["write_s"], 1e-9) / 1e9,
                      triples_per_s=dp.n_pairs * V / tm["total_s"])
            if name == "strict":                      # the file holds what :772-797 prints: one header + `rows` lines
                with open(os.path.join(td, name + ".pair"), "rb") as f:
                    head = f.readline()
                    tm["header_ok"] = head == b"BARCODE\tSM1.ID\tSM2.ID\tLLK12\tPOSTPRB\n"
            out[name] = tm
            os.remove(os.path.join(td, name + ".pair"))
    return out


def cli_leg(n_reads=2_000_000, n_snps=60_000, n_samples=16, n_barcodes=3_000):
    """BAM + VCF in, four files out through the `demuxlet` binary (rows f1-f3 + the engine): the scan's reads/s on this box's host cores
    (windowed, all cores; and read by read on one thread) and the stage seconds of the dmx_demuxlet_run call behind it.  The inputs are
    tools/make_cli_bench.py's synthetic coordinate-sorted BAM and GT VCF, made here in a temporary directory (no reference data)."""
    import re
    import subprocess
    import tempfile
    cli = ROOT / "demuxlet_amd" / "demuxlet"
    if not cli.exists():
        return None
    out = {"what": f"demuxlet --sam bench.bam --vcf bench.vcf --field GT: {n_reads} reads x {n_snps} SNPs x {n_samples} samples x {n_barcodes} barcodes "
                   f"(tools/make_cli_bench.py), STRICT, tie arbiter on", "host_cores": host_cores()}
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([sys.executable, str(ROOT / "tools" / "make_cli_bench.py"), td, str(n_reads), str(n_snps), str(n_samples), str(n_barcodes)],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for name, env_extra, reps in (("all_cores", {}, 3), ("one_thread", {"DMX_THREADS": "1"}, 1)):
            env = dict(os.environ, DMX_CLI_TIMING="1", DMX_E2E_TIMING="1", **env_extra)
            runs = []
            for _ in range(reps):                                 # the all-cores scan three times: the record is the median run
                t0 = time.perf_counter()
                r = subprocess.run([str(cli), "--sam", f"{td}/bench.bam", "--vcf", f"{td}/bench.vcf", "--field", "GT", "--out", f"{td}/o_{name}"],
                                   capture_output=True, text=True, env=env)
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    return None
                rec = {"wall_s": wall}
                m = re.search(r"scan timing \((\d+) threads, ([a-z ]+)\): total ([0-9.]+) s = ([0-9.e+]+) reads/s", r.stderr)
                if m:
                    rec.update(scan_threads=int(m.group(1)), scan_mode=m.group(2), scan_s=float(m.group(3)), scan_reads_per_s=float(m.group(4)))
                m = re.search(r'\{"dmx_demuxlet_run": (\{.*?\})\}', r.stderr)
                if m:
                    rec["dmx_demuxlet_run"] = json.loads(m.group(1))
                runs.append(rec)
            runs.sort(key=lambda x: x.get("scan_s", 1e9))
            out[name] = dict(runs[len(runs) // 2], scan_s_all_runs=[x.get("scan_s") for x in runs])
        same = all(open(f"{td}/o_all_cores.{suf}", "rb").read() == open(f"{td}/o_one_thread.{suf}", "rb").read() for suf in ("single", "sing2", "best"))
        out["files_identical_between_the_two_scans"] = same
    return out


def pmc_profile(cfgno, B, mode, dense):
    """HBM traffic and VALU instruction counts of one launch of the dominant kernel are properties of the workload; they come
    from the committed rocprofv3 PMC passes of the SAME workload (profiles/, tools/profile_round.sh).  A dense configuration does
    exactly the same work for every barcode, so counts measured on another number of its barcodes scale linearly (said in the
    record: `scaled_from_barcodes`); sparse ones must match the size."""
    for name in (f"pmc_cfg{cfgno}_{mode}.json", f"pmc_cfg{cfgno}.json"):
        p = ROOT / "profiles" / name
        if p.exists():
            pj = json.loads(p.read_text())
            if pj.get("mode", "strict") != mode or not pj.get("barcodes_per_gpu"):
                continue
            pj["counts_file"] = f"profiles/{name}"
            if pj["barcodes_per_gpu"] == B:
                return pj
            if dense:
                f = B / pj["barcodes_per_gpu"]
                out = {k: (v * f if isinstance(v, (int, float)) and k.endswith(("_per_launch", "_raw", "_x2", "_bytes")) else v) for k, v in pj.items()}
                out["scaled_from_barcodes"] = pj["barcodes_per_gpu"]
                out["barcodes_per_gpu"] = B
                return out
    return None


def shard_range(B_total, world, rank):
    """Strong scaling: rank r owns barcodes [lo, hi) of the B_total (contiguous, equal counts: equal work on a dense pileup)."""
    return (B_total * rank) // world, (B_total * (rank + 1)) // world


def gather_records(torch, dist, rec, counts, rank, world):
    """THE collective of the job: one fixed-size record per barcode -> rank 0 (RCCL gather over xGMI; gloo in the CPU test).  Ranks may
    own B/N or B/N + 1 barcodes, so the records are padded to the largest count (gather wants equal shapes).  Returns the list of
    per-rank matrices on rank 0 (still padded; counts[r] rows of matrix r are real), None elsewhere."""
    pad = max(counts) - rec.shape[0]
    if pad:
        rec = torch.cat([rec, rec.new_zeros((pad, rec.shape[1]))], dim=0)
    outs = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
    dist.gather(rec, outs, dst=0)
    return outs


def collect_timing(torch, dist, dev, elapsed, own_elapsed, n_pairs, steps, world):
    """max-over-ranks wall time of the timed region, the job's covered pairs and every rank's own ms per step (all ranks call this)."""
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([float(n_pairs)], dtype=torch.float64, device=dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    mine = torch.tensor([1e3 * own_elapsed / steps], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return float(tmax.item()), float(cnt.item()), [float(x.item()) for x in allr]


class Ctx:
    pass


def run_config(cx, cfgno, cfg, mode, steps, warmup, with_cpu, with_log, with_e2e=False, parity_s=0.0, defer_cpu=False):
    """Generate the workload of one configuration on this rank's GPU, time `steps` passes, return the record (rank 0)."""
    torch, dist, engine, synth, synth_torch = cx.torch, cx.dist, cx.engine, cx.synth, cx.synth_torch
    dev, world, rank, local = cx.dev, cx.world, cx.rank, cx.local
    fast = mode == "fast"
    B_total, S, V, A = cfg["B"], cfg["S"], cfg["V"], len(cfg["alphas"])
    # strong scaling: rank r owns barcodes [lo, hi) of the B_total (contiguous, equal counts: equal work on a dense pileup)
    lo, hi = shard_range(B_total, world, rank)
    B = hi - lo
    # the workload of the last call is kept (one entry): the same configuration in the other mode does not generate it again
    key = (cfgno, B_total, S, V, cfg["field"], cfg["delta"], cfg["rbar"], world, rank)
    if getattr(cx, "inputs", None) is not None and cx.inputs[0] == key:
        raw, g, dp = cx.inputs[1:]
    else:
        cx.inputs = None
        gc.collect()
        torch.cuda.empty_cache()
        rng = np.random.default_rng(0xD3A00000 + cfgno)             # the panel is shared by all ranks
        raw, g = genotype_matrix(engine, synth, rng, S, V, cfg["field"])
        dosage = torch.from_numpy(np.clip(raw.alleles, 0, 1).sum(axis=2).astype(np.float32)).to(dev)
        dp = synth_torch.make_device_pileup(dosage, B, cfg["delta"], cfg["rbar"], seed=0xD3A0 + 1000 * cfgno + rank,
                                            device=dev, cell_id_base=lo)
        del dosage
        cx.inputs = (key, raw, g, dp)
    torch.cuda.synchronize()

    # one explicit HIP stream for everything timed: the engine launches on it, torch events are recorded on it and RCCL
    # orders against it (the legacy NULL stream would not do: dmx_engine_set_stream(NULL) means "the engine's own")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = engine.Engine(V, cfg["alphas"], 0.5, device=local, mode=engine.capi.DMX_MODE_FAST if fast else engine.capi.DMX_MODE_STRICT)
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

    counts = [shard_range(B_total, world, r)[1] - shard_range(B_total, world, r)[0] for r in range(world)]
    gathered = None
    # dmx_engine_run (what dmx_demuxlet_run does too): K1 beside K2 -> K3 -> K3b in one call.  tools/profile_round.sh wants the kernels one after
    # the other (DMX_EXPERIMENTS=1 DMX_NO_OVERLAP=1: the engine honours experiment switches only behind that fence, and so does this script)
    beside = bool(cfg["doublet"]) and not (os.environ.get("DMX_EXPERIMENTS") == "1" and os.environ.get("DMX_NO_OVERLAP"))

    def step(ev=None):
        nonlocal gathered
        if ev: ev[0].record()
        if beside:                                  # dmx_engine_run: K1 beside K2 -> K3 -> K3b (no boundary between them on this stream)
            if ev: ev[1].record()
            eng.run()
        else:
            eng.run_singlet()
            if ev: ev[1].record()
            if cfg["doublet"]:
                eng.run_doublet()
        if ev: ev[2].record()
        if cx.use_dist:
            gathered = gather_records(torch, dist, record_matrix(), counts, rank, world)
        if ev: ev[3].record()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    eng.reset_kernel_times()                        # per-kernel HIP events of the timed launches only
    if cx.use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        step(evs[i])
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0          # this rank's own time (before waiting for the others)
    if cx.use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [1e3 * own_elapsed / steps]
    if cx.use_dist:
        elapsed, total_pairs, per_rank_ms = collect_timing(torch, dist, dev, elapsed, own_elapsed, dp.n_pairs, steps, world)
    else:
        total_pairs = float(dp.n_pairs)

    # the timed path's own results for the barcodes the CPU leg evaluates: taken NOW, after the last timed step and
    # before the untimed one-after-the-other steps below overwrite them
    rows = None
    if rank == 0 and (with_cpu or parity_s > 0):
        rows = engine_rows(cx, eng, cfg, cpu_sample_cells(dp, cfg, 12.0 if with_cpu else parity_s))

    # per-kernel times: the engine brackets every launch of K1, K2, K3 and K3b with HIP events on the stream it launches on
    # (dmx_engine_mean_kernel_times: mean over the TIMED launches, at most the last 16); torch's events on the same stream give
    # the K1 and K2 + K3 + K3b spans as a cross-check.  The roofline of the dominant kernel uses these timed-launch means.
    km = eng.mean_kernel_times()
    launched = eng.kernel_names()                   # which kernels the timed steps launched, and where K1 ran (dmx_engine_kernel_names)
    k1_late = launched["k1_placement"] == 2         # K1 beside K3 + K3b, after K2 (where K2 leaves K1 no room: k_doublet_clsp)
    k1_ms = float(km.singlet_ms)
    k2_only_ms = float(km.doublet_ms) if cfg["doublet"] else 0.0
    k3_ms, k3b_ms = (float(km.reduce_ms), float(km.certify_ms)) if cfg["doublet"] else (0.0, 0.0)
    alone = None
    if beside:
        # K1 ran beside K2 in the timed steps (the product's way, dmx_engine_run): K1's own event span is mostly waiting and K2's
        # contains what K1 took from it.  Two extra, untimed steps with the kernels one after the other give each kernel ALONE
        # (full record only: `kernel_ms_alone`).
        eng.reset_kernel_times()
        for _ in range(2):
            eng.run_singlet(); eng.run_doublet()
        torch.cuda.synchronize()
        ka = eng.mean_kernel_times()
        alone = {"k_singlet": float(ka.singlet_ms), "k_doublet": float(ka.doublet_ms), "k_reduce": float(ka.reduce_ms), "k_certify": float(ka.certify_ms)}
        k1_ms = alone["k_singlet"]                      # K1's timed-launch span is not a kernel time when it runs beside K2
    k1_span_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    k2_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs])) if cfg["doublet"] else 0.0      # K2 + K3 + K3b (beside: K1 as well)
    gather_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs])) if cx.use_dist else 0.0
    # seconds of K1 + K2 + K3 + K3b per step: the torch span of the timed steps (beside: one span covers all four)
    kernels_s = ((k1_span_ms if beside else k1_ms) + k2_ms) * 1e-3
    out = None
    if rank == 0:
        if cx.use_dist and gathered is not None:
            assert len(gathered) == world and gathered[0].shape[0] == max(counts)
        triples = total_pairs * V
        ms_per_step = 1e3 * elapsed / steps
        dom_ms, dom_bytes, dom_name = (k2_only_ms, nbytes.doublet_bytes, "k_doublet") if cfg["doublet"] else (k1_ms, nbytes.singlet_bytes, "k_singlet")
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, valu, counts_file = None, None, None
        pj = pmc_profile(cfgno, B, mode, cfg["delta"] >= 1.0)
        dom_launched = launched["doublet" if cfg["doublet"] else "singlet"]
        counts_stale = None
        if pj and dom_launched and pj.get("kernel") and pj["kernel"] != dom_launched:
            # the committed counters belong to another kernel than the one this run launched (a kernel change without a refreshed
            # profiles/pmc_*.json): quote no counter-derived fraction rather than a stale one
            counts_stale = {"counts": pj.get("counts_file"), "counted_kernel": pj["kernel"], "launched_kernel": dom_launched}
            pj = None
        if pj:
            traffic, counts_file = pj.get("hbm_bytes_per_launch"), pj.get("counts_file")
            if pj.get("issue_cycles_per_launch"):
                # issue-cycle roofline (the binding one, SURVEY 8d / DESIGN 6): every wave64 VALU instruction holds its SIMD's issue
                # port for 4 cycles (FP64, transcendental) or 2 (the rest) -> `frac`; `frac_measured_costs` prices the non-FP64
                # instructions at their measured average cost in this kernel's loops (tools/micro/valu_rate.hip, tools/isa_mix.py);
                # 1024 SIMDs x 2.4 GHz cycles are available per second.  Counts per launch from profiles/ PMC, time live.
                rate = pj["issue_cycles_per_launch"] / (dom_ms * 1e-3)
                valu = {"bound": "valu_issue", "achieved": rate, "peak": SIMDS * CLOCK_HZ, "unit": "SIMD issue cycles/s",
                        "frac": rate / (SIMDS * CLOCK_HZ), "kernel": pj.get("kernel"), "counts": counts_file,
                        "counts_scaled_from_barcodes": pj.get("scaled_from_barcodes"),
                        "fp64_insts_per_launch": pj.get("fp64_insts_per_launch"), "other_valu_insts_per_launch": pj.get("other_valu_insts_per_launch"),
                        "upper_bound_all_insts_at_4_cycles": pj.get("valu_wave_insts_per_launch", 0) / (dom_ms * 1e-3) / VALU_PEAK_WAVE_INSTS,
                        "frac_measured_costs": (pj["issue_cycles_measured_costs_per_launch"] / (dom_ms * 1e-3) / (SIMDS * CLOCK_HZ)
                                                if pj.get("issue_cycles_measured_costs_per_launch") else None),
                        "cycles_per_other_valu_inst": pj.get("cycles_per_other_valu_inst")}
            elif pj.get("valu_wave_insts_per_launch"):
                rate = pj["valu_wave_insts_per_launch"] / (dom_ms * 1e-3)       # every VALU instruction charged 4 cycles: an upper bound
                valu = {"bound": "valu_issue", "achieved": rate, "peak": VALU_PEAK_WAVE_INSTS, "unit": "wave-instructions/s",
                        "frac": rate / VALU_PEAK_WAVE_INSTS, "kernel": pj.get("kernel"), "counts": counts_file}
        logs = total_pairs * ((V + 1) + (V * V * A + A if cfg["doublet"] else 0))
        # GT inputs run the genotype-class kernels (log once per distinct class pair) and FAST evaluates the printed entries only:
        # both EXECUTE fewer logs / flops than the reference's count.  For them only executed figures are put against the machine
        # (VERDICT r2 weak 9b); the logical count stays, named as such.
        reduced = fast or cfg["field"] == "GT"
        executed = None
        if pj and pj.get("fp64_fma_per_launch") is not None:
            # EXECUTED FP64 flop of the dominant kernel: PMC per-type wave-instruction counts x 64 lanes, FMA = 2, over its live event time
            ex_flop = 64.0 * (2 * pj["fp64_fma_per_launch"] + pj["fp64_mul_per_launch"] + pj["fp64_add_per_launch"] + pj.get("fp64_trans_per_launch", 0.0))
            executed = {"kernel": pj.get("kernel"), "fp64_flop_per_launch": ex_flop, "tflops": ex_flop / (dom_ms * 1e-3) / 1e12,
                        "frac_of_peak": ex_flop / (dom_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}
        out = {
            "metric": METRIC,
            "value": triples * steps / elapsed, "unit": "triples/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            # N > 1 = cfg4's fixed 100k barcodes cut N ways (strong).  The N = 1 line is cfg3, another workload: it is no point of that curve and
            # claims neither (the curve's N = 1 point is this line's also[cfg4/strict])
            "scaling": "strong" if world > 1 else "none",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"] + (" [DMX_MODE_FAST]" if fast else ""), "tag": f"cfg{cfgno}/{mode}", "barcodes_total": B_total,
                       "barcodes_per_gpu": B, "snps": S, "samples": V, "alphas": list(cfg["alphas"]),
                       "covered_pairs_per_gpu": dp.n_pairs, "reads_per_gpu": dp.n_reads, "mode": mode,
                       "sharding": (f"{B_total} barcodes cut into {world} contiguous ranges, one per GPU (strong scaling); "
                                    f"one RCCL gather of {max(counts)} x {record_matrix().shape[1] * 8} B records per rank per step")
                       if world > 1 else "single GPU"},
            # HBM fraction as the metric asks (the path is VALU-issue bound, SURVEY 8d: see roofline_valu).  kernel_ms = mean HIP-event
            # time of the dominant kernel over the TIMED launches (the engine's own events on the launch stream); traffic = PMC
            # FETCH_SIZE x2 + WRITE_SIZE of one launch of this workload, from the committed pass named in `counts`
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes,
                         "kernel_ms": dom_ms, "counts": counts_file, "kernel_launched": dom_launched, "counts_stale": counts_stale},
            "roofline_valu": valu,
            "kernels_launched": launched,
            # logical = the REFERENCE's count of log() evaluations for this workload (P*(V+1) + P*(V*V*A+A)) over K1 + K2 + K3 + K3b time;
            # GT inputs and FAST execute fewer (genotype classes / printed-entry set)
            "fp64_valu": {"logical_log_terms_per_s": logs / world / kernels_s,
                          "executes_fewer_logs_than_logical": reduced,
                          "executed": executed,
                          "kernel_ms": {"k_singlet": k1_ms, "k_doublet": k2_only_ms, "k_reduce": k3_ms, "k_certify": k3b_ms,
                                        "torch_events_k_singlet": k1_span_ms, "torch_events_k_doublet+k_reduce+k_certify": k2_ms,
                                        "k1_beside_k2": beside and not k1_late, "k1_beside_k3b_in_one_call": k1_late},
                          "kernel_ms_alone": alone,
                          "peak_tflops": FP64_VALU_PEAK_TFLOPS},
        }
        if world > 1 or cx.use_dist:
            out["ranks_seen"] = dist.get_world_size()
            out["per_rank_ms_per_step"] = per_rank_ms
            out["gather_ms"] = gather_ms
            out["rank_devices"] = getattr(cx, "devices_txt", None)
        if cfg["doublet"]:
            out["pair_evals_per_s"] = total_pairs * V * V * A * steps / elapsed
        if with_log:
            # the device's log() ceiling from a register-resident microkernel, same run: what fraction of it the
            # path's LOGICAL log terms amount to (the genotype-class kernels and FAST mode execute fewer logs than the
            # reference's count, so this fraction can exceed 1 there; it cannot for GP/PL inputs in STRICT mode)
            import ctypes
            rates = []
            for which in (0, 1):
                r = ctypes.c_double(0.0)
                engine.check(engine.capi.load().dmx_debug_log_rate(which, 4096, local, ctypes.byref(r)))
                rates.append(r.value)
            out["log_microkernel"] = {"dmx_log_per_s": rates[0], "ocml_log_per_s": rates[1]}
            # ALGORITHMIC FP64 rate against the 78.6 TF vector peak, log() costed as 1 op and as C_log flop.  C_log is
            # what dmx_log executes per evaluation (csrc/dmx_log.hpp: LOG_FP64_INSTS FP64 instructions, each priced as an FMA = 2
            # flop; the integer argument reduction is not FP64 work).  The microkernel's rate stays in the record as the device's
            # log()/s ceiling, but it is latency-bound and is NOT used to price a log.  Only where the kernels
            # execute the reference's count (soft fields, STRICT): elsewhere the fractions would exceed 1 without meaning it.
            rbar = dp.n_reads / max(dp.n_pairs, 1)
            ops1 = dp.n_pairs * ((rbar * 11 + 8 + V * 7 + 7) + ((rbar * A * 54 + A * 18 + V * V * A * 20 + A * 20) if cfg["doublet"] else 0))
            c_log = 2.0 * LOG_FP64_INSTS
            out["fp64_valu"]["c_log_flops"] = c_log
            if not reduced:
                out["log_microkernel"]["path_logs_over_dmx_log_ceiling"] = out["fp64_valu"]["logical_log_terms_per_s"] / rates[0]
                out["fp64_valu"].update({"algorithmic_tflops_log_as_1_op": ops1 / kernels_s / 1e12,
                                         "algorithmic_tflops_log_as_c_log": (ops1 + logs / world * (c_log - 1)) / kernels_s / 1e12,
                                         "frac_of_peak_log_as_c_log": (ops1 + logs / world * (c_log - 1)) / kernels_s / 1e12 / FP64_VALU_PEAK_TFLOPS})
            else:
                out["fp64_valu"]["logical_tflops_log_as_1_op"] = ops1 / kernels_s / 1e12
        if with_e2e == "e2e4wp" and cfg["doublet"]:
            out["end_to_end_write_pair"] = write_pair_leg(cx, cfg, dp, g, B, S, V)
        elif with_e2e and cfg["doublet"]:
            # The same workload through the one-call C-ABI entry (dmx_demuxlet_run: frozen host pileup -> H2D -> K1/K2/K3(+K3b) ->
            # tie arbiter -> .single/.sing2/.best), stage seconds from dmx_job_timing.  Outside the timed region; host -> device
            # copies included (this is the PCIe-inclusive picture DESIGN.md asks for).
            import tempfile
            h = dp.host_slice(0, B)
            nreads = np.diff(h["cell_read_off"]).astype(np.int32)
            hp = engine.HostPileup(B, S, h["cell_pair_off"], h["cell_read_off"], h["pair_snp"], h["pair_nrd"], h["reads"], nreads, nreads, nreads)
            bcs = [f"BC{i:07d}-1" for i in range(B)]
            sms = [f"SM{j:02d}" for j in range(V)]
            e2e = {"what": "dmx_demuxlet_run on this workload from a frozen HOST pileup (no --write-pair, tie arbiter on, 1 GPU): wall seconds per stage"}
            light = with_e2e == "e2e6"               # the realistic-coverage record: both modes, host and device pileup, no BAM/VCF leg
            covered = int((np.diff(h["cell_pair_off"]) > 0).sum())
            with tempfile.TemporaryDirectory() as td_dir:
                for name, md in (("strict", engine.capi.DMX_MODE_STRICT), ("fast", engine.capi.DMX_MODE_FAST)):
                    tm = engine.demuxlet_run(hp, g, sms, cfg["alphas"], os.path.join(td_dir, name), barcodes=bcs, timing=True, mode=md)
                    tm.pop("reserved", None)
                    tm["arbiter_format_write_frac"] = tm["write_s"] / tm["total_s"]
                    tm["triples_per_s"] = dp.n_pairs * V / tm["total_s"]
                    # barcodes whose grid the arbiter fetched (K3's near-tie flags): at ~2 000 covered SNPs per droplet no longer a corner case
                    tm["grid_fetched_frac"] = tm["n_cells_grid_fetched"] / max(covered, 1)
                    e2e[name] = tm
                    # the same job from the DEVICE-resident pileup (dmx_pileup.memory = DMX_MEM_DEVICE: no host slicing, no H2D of the CSR;
                    # only the per-cell counters and the barcodes are host memory); same files, byte for byte
                    ds = dp.as_struct()
                    ds.rd_totl = ds.rd_pass = ds.rd_uniq = nreads.ctypes.data
                    td = engine.demuxlet_run(ds, g, sms, cfg["alphas"], os.path.join(td_dir, name + "_dev"), barcodes=bcs, timing=True, mode=md)
                    td.pop("reserved", None)
                    td["triples_per_s"] = dp.n_pairs * V / td["total_s"]
                    td["files_identical_to_the_host_run"] = all(
                        open(os.path.join(td_dir, f"{name}.{suf}"), "rb").read() == open(os.path.join(td_dir, f"{name}_dev.{suf}"), "rb").read()
                        for suf in ("single", "sing2", "best"))
                    e2e[name + "_device_pileup"] = td
            if not light:
                try:
                    e2e["from_bam_and_vcf"] = cli_leg()
                except Exception as ex:                  # the leg is a side record: never let it take the bench line down
                    e2e["from_bam_and_vcf"] = {"error": repr(ex)}
            out["end_to_end"] = e2e
            del hp, h
        if with_cpu and defer_cpu:
            # N > 1: the CPU leg runs on rank 0 after the LAST barrier and the group's shutdown (main), so that no rank waits in a collective
            # while one host thread works for half a minute
            cx.deferred_cpu = (dp, g, cfg, rows, fast)
        elif with_cpu:                                   # the CPU baseline: rank 0, after the timed region, on the first barcodes of ITS range
            attach_cpu_baseline(out, cpu_baseline(dp, g, cfg, rows=rows, fast=fast))
        elif rows is not None:                           # nested records: the comparison only, on a few barcodes
            out["parity_check"] = cpu_baseline(dp, g, cfg, rows=rows, fast=fast, legs=False)["parity_check"]
    eng.close()
    del eng, dp, gathered, raw, g
    gc.collect()
    torch.cuda.empty_cache()
    return out


def attach_cpu_baseline(out, cb):
    out["parity_check"] = cb.pop("parity_check", None)
    out["cpu_baseline"] = cb
    cb["gpu_over_cpu"] = out["value"] / cb["value"]


def sig(x, n=6):
    """floats to n significant digits (the stdout line is compact; the full record keeps every digit)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    return x


def pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def compact_line(full):
    """The ONE stdout line: the driver's keys + config + roofline + roofline_valu + cpu_baseline + a ~200-byte object per nested
    configuration.  No prose.  Everything else is in the full record (bench_full.json, stderr)."""
    line = pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data", "pair_evals_per_s", "ranks_seen", "per_rank_ms_per_step", "gather_ms", "rank_devices"))
    line["vs_baseline"] = None
    line["config"] = pick(full["config"], ("workload", "tag", "barcodes_total", "barcodes_per_gpu", "snps", "samples", "alphas",
                                            "covered_pairs_per_gpu", "reads_per_gpu", "mode"))
    line["config"]["sharding"] = f"{full['n_gpus']} contiguous barcode ranges + one RCCL gather per step" if full["n_gpus"] > 1 else "single GPU"
    line["roofline"] = pick(full["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                "kernel_ms", "counts", "kernel_launched", "counts_stale"))
    line["roofline"].setdefault("traffic", None)
    alone = (full.get("fp64_valu") or {}).get("kernel_ms_alone")
    if alone:                                   # the dominant kernel with nothing beside it (two extra untimed steps)
        line["roofline"]["kernel_ms_alone"] = alone.get(full["roofline"]["kernel"])
    if full.get("roofline_valu"):
        line["roofline_valu"] = pick(full["roofline_valu"], ("bound", "frac", "frac_measured_costs", "kernel"))
    if full.get("parity_check"):                # the timed path's rows against the oracle, same barcodes, timed size
        line["parity_check"] = pick(full["parity_check"], ("barcodes", "max_abs_delta", "calls_identical", "near_tie_flagged", "tolerance", "ok"))
    cb = full.get("cpu_baseline")
    if cb:
        c = pick(cb, ("value", "unit", "cores", "kind", "sample", "seconds", "gpu_over_cpu"))
        if cb.get("reference_slice"):
            c["reference_slice"] = pick(cb["reference_slice"], ("value", "cores", "seconds"))
        if cb.get("all_cores"):
            c["all_cores"] = pick(cb["all_cores"], ("value", "cores"))
        line["cpu_baseline"] = c
    if full.get("also"):
        line["also"] = [dict(workload=a["config"]["tag"], barcodes=a["config"]["barcodes_total"], ms_per_step=a["ms_per_step"], value=a["value"],
                             steps=a["steps"], kernel_ms=a["roofline"]["kernel_ms"], roofline_frac=a["roofline"]["frac"],
                             roofline_valu_frac=(a.get("roofline_valu") or {}).get("frac"),
                             roofline_valu_frac_measured=(a.get("roofline_valu") or {}).get("frac_measured_costs"),
                             **pick(a, ("ranks_seen", "gather_ms")),
                             **({"parity": [a["parity_check"]["barcodes"], a["parity_check"]["max_abs_delta"], a["parity_check"]["calls_identical"]]}
                                if a.get("parity_check") else {}))
                        for a in full["also"]]
    e2e = full.get("end_to_end")
    if e2e:
        line["end_to_end"] = {m: pick(e2e[m], ("total_s", "stage_s", "wait_s", "write_s", "files_identical_to_the_host_run"))
                              for m in ("strict", "fast", "strict_device_pileup", "fast_device_pileup") if m in e2e}
        if e2e.get("cfg6"):                         # the realistic-coverage job: seconds, writer-thread seconds, share of barcodes whose grid K3's flags fetched
            line["end_to_end"]["cfg6"] = {m: pick(e2e["cfg6"][m], ("total_s", "write_s", "grid_fetched_frac"))
                                          for m in ("strict", "fast", "strict_device_pileup", "fast_device_pileup") if m in e2e["cfg6"]}
        if e2e.get("cfg4_shard_write_pair"):        # cfg4's 12 500-barcode shard with --write-pair: seconds, and the writer's .pair rate
            wp = e2e["cfg4_shard_write_pair"]
            line["end_to_end"]["cfg4_shard_write_pair"] = dict(pair_rows=wp["pair_rows"], **pick(wp, ("slowest_rank_total_s",)),
                                                               **{m: pick(wp[m], ("total_s", "wait_s", "write_s", "pair_bytes", "pair_rows_per_s_of_write_s"))
                                                                  for m in ("strict", "fast") if m in wp})
        cli = e2e.get("from_bam_and_vcf") or {}
        if "all_cores" in cli:
            line["end_to_end"]["bam_vcf_scan_reads_per_s"] = cli["all_cores"].get("scan_reads_per_s")
    line["full_record"] = full.get("full_record")
    return sig(line)


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: check the devices, then become the launcher.  One process per GPU,
    rendezvous on 127.0.0.1 (barcodes shard over the GPUs of ONE node, cmd_cram_demuxlet.cpp:576 — independent barcodes)."""
    inject = bool(os.environ.get("DMX_BENCH_INJECT"))       # CPU test hook: gloo ranks with an injected compute, no devices needed
    if not inject:
        n = visible_gpus()
        if args.gpus > n:
            sys.exit(f"bench.py: {args.gpus} GPUs requested, {n} visible")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def run_injected(cx, cfg, steps, warmup):
    """DMX_BENCH_INJECT=1 (tests/test_bench_dist_cpu.py): the N-rank bookkeeping of this script — launch, world-size checks, ranges,
    padded gather, max-over-ranks timing, the compact line — on CPU tensors over gloo with a made-up record matrix instead of the
    engine.  The line says so in `data`; it is never a measurement."""
    torch, dist = cx.torch, cx.dist
    world, rank, V, A = cx.world, cx.rank, cfg["V"], len(cfg["alphas"])
    lo, hi = shard_range(cfg["B"], world, rank)
    counts = [shard_range(cfg["B"], world, r)[1] - shard_range(cfg["B"], world, r)[0] for r in range(world)]
    ncols = 2 * V + 1 + A + 24
    rec = (torch.arange(lo, hi, dtype=torch.float64)[:, None] + 0.5 * torch.arange(ncols, dtype=torch.float64)[None, :]).contiguous()
    gathered = None
    for _ in range(warmup):
        gathered = gather_records(torch, dist, rec, counts, rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        gathered = gather_records(torch, dist, rec, counts, rank, world)
    own = time.perf_counter() - t0
    dist.barrier()
    elapsed = time.perf_counter() - t0
    n_pairs = (hi - lo) * cfg["S"]
    elapsed, total_pairs, per_rank = collect_timing(torch, dist, cx.dev, elapsed, own, n_pairs, steps, world)
    if rank != 0:
        return None
    b = 0
    for r in range(world):                                   # rank order == barcode order, padding rows zero
        assert torch.equal(gathered[r][:counts[r], 0], torch.arange(b, b + counts[r], dtype=torch.float64))
        assert (gathered[r][counts[r]:] == 0).all()
        b += counts[r]
    assert b == cfg["B"]
    ms = 1e3 * elapsed / steps
    return {"metric": METRIC, "value": total_pairs * V * steps / elapsed, "unit": "triples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "INJECTED records over gloo (DMX_BENCH_INJECT test hook: no GPU, no engine; not a measurement)",
            "config": {"workload": cfg["name"], "tag": "cfg4/injected", "barcodes_total": cfg["B"], "barcodes_per_gpu": hi - lo, "snps": cfg["S"],
                       "samples": V, "alphas": list(cfg["alphas"]), "mode": "injected"},
            "roofline": {"bound": "hbm", "kernel": "none (injected)", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0,
                         "traffic": None, "kernel_ms": ms},
            "ranks_seen": dist.get_world_size(), "per_rank_ms_per_step": per_rank, "gather_ms": ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0, choices=[0] + sorted(CONFIGS),
                    help="0 = the driver's default: cfg3 (+ nested cfg3-fast/cfg2/cfg5/cfg4-whole records) at N=1, cfg4 sharded at N>1")
    ap.add_argument("--cells", type=int, default=0, help="override the TOTAL barcode count (smaller = quicker run; not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="N=1: skip the nested records of the other configurations")
    ap.add_argument("--samples", type=int, default=0, help="override the number of samples (experiments; not the headline)")
    ap.add_argument("--field", default="", help="override the genotype field GT|GP|PL (experiments; not the headline)")
    ap.add_argument("--fast", action="store_true", help="DMX_MODE_FAST for the main record (opt-in, not the headline)")
    ap.add_argument("--alphas", default="", help="override the alpha grid, comma separated (experiments; not the headline)")
    ap.add_argument("--e2e-write-pair", action="store_true", help="with --config/--cells: add the dmx_demuxlet_run --write-pair leg of this workload to the full record")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")

    # --gpus decides the number of ranks.  No launcher in the environment and N > 1: become the launcher (never returns).
    # Under a launcher: its WORLD_SIZE must be what --gpus says.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            self_launch(args)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    # stdout carries exactly one line, the JSON record.  Libraries that print banners from C (RCCL's version block is written
    # to fd 1 and flushed at exit, i.e. AFTER anything Python printed) are sent to stderr: fd 1 is re-pointed at fd 2 for the
    # whole run and the record goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    cx = Ctx()
    cx.torch, cx.dist = torch, dist
    cx.world = int(os.environ.get("WORLD_SIZE", "1"))
    cx.rank = int(os.environ.get("RANK", "0"))
    cx.local = int(os.environ.get("LOCAL_RANK", "0"))
    inject = bool(os.environ.get("DMX_BENCH_INJECT"))
    cx.inputs = None
    # DMX_BENCH_FORCE_DIST=1 runs the collective code path with a 1-rank RCCL group (1-GPU boxes: exercises the gather)
    cx.use_dist = cx.world > 1 or bool(os.environ.get("DMX_BENCH_FORCE_DIST"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if inject:
        cx.dev = torch.device("cpu")
        dist.init_process_group(backend="gloo", rank=cx.rank, world_size=cx.world)
    else:
        ndev = visible_gpus()
        if ndev == 0:
            sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
        if cx.world > ndev or cx.local >= ndev:
            sys.exit(f"bench.py: {cx.world} GPUs requested, {ndev} visible")
        torch.cuda.set_device(cx.local)
        cx.dev = torch.device("cuda", cx.local)
        if cx.use_dist:
            dist.init_process_group(backend="nccl", device_id=cx.dev, rank=cx.rank, world_size=cx.world)
    if cx.use_dist or inject:
        # the group must be what --gpus asked for, one rank per DISTINCT device
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        if not inject:
            pr = torch.cuda.get_device_properties(cx.local)
            ids = [None] * cx.world
            dist.all_gather_object(ids, (cx.local, getattr(pr, "pci_domain_id", None), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)))
            cx.devices = ids
            cx.devices_txt = [f"{i[0]}" + (f"@{i[1]:04x}:{i[2]:02x}:{i[3]:02x}" if None not in i[1:] else "") for i in ids]   # local ordinal @ PCI domain:bus:device
            if len({i[0] for i in ids}) != cx.world:
                sys.exit(f"bench.py: {cx.world} ranks on {len({i[0] for i in ids})} distinct local devices")
            if all(i[2] is not None for i in ids) and len({i[1:] for i in ids}) != cx.world and cx.rank == 0:
                sys.stderr.write(f"bench.py: WARNING: ranks report non-distinct PCI ids: {ids}\n")

    default_run = args.config == 0
    cfgno = args.config or (3 if cx.world == 1 else 4)
    cfg = dict(CONFIGS[cfgno])
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
    if args.cells:
        cfg["name"] += f" [override: {args.cells} barcodes]"

    if inject:
        out = run_injected(cx, cfg, args.steps, args.warmup)
    else:
        from demuxlet_amd import build, engine, synth, synth_torch
        cx.engine, cx.synth, cx.synth_torch = engine, synth, synth_torch
        if cx.rank == 0:
            build.build()
        if cx.use_dist:
            dist.barrier()
        single = cx.world == 1 and not cx.use_dist
        # (the CPU baseline runs on rank 0 at every N: a SCALE line is judged by the same rule as the N = 1 line.  Its engine rows are taken
        #  right after the timed steps; with a process group the oracle itself runs after the group's last barrier — see the end of main)
        cx.deferred_cpu = None
        out = run_config(cx, cfgno, cfg, "fast" if args.fast else "strict", args.steps, args.warmup,
                         with_cpu=not args.no_cpu_baseline, with_log=single,
                         with_e2e=(single and default_run and not args.only) or (single and args.e2e_write_pair and "e2e4wp"),
                         defer_cpu=cx.use_dist)
        keys = ("value", "unit", "n_gpus", "ms_per_step", "steps", "warmup", "config", "roofline", "roofline_valu", "fp64_valu", "pair_evals_per_s",
                "ranks_seen", "per_rank_ms_per_step", "gather_ms", "parity_check")
        if single and default_run and not args.only:
            # nested records of the same run: the other single-GPU BASELINE configurations and the opt-in mode, fewer steps each
            also = []
            k, w = max(2, min(args.steps, 5)), min(args.warmup, 1)
            # cfg4 WHOLE on this one GPU (2 steps each: a STRICT pass is ~7 s) is the N = 1 point of the strong-scaling curve whose
            # N > 1 points the driver measures with `--gpus N` (same workload, same code path minus the gather)
            # ... and its 12 500-barcode shard, one GPU's share of the 8-GPU run: what a perfectly scaling N = 8 step costs (plus the gather)
            # cfg6 = SURVEY 8d's "realistic run reported alongside": 2 000 covered SNPs per barcode of 100 k (what a 10x droplet looks like), GT
            for no, mode, kk, shard in ((3, "fast", k, 0), (2, "strict", k, 0), (5, "strict", k, 0), (5, "fast", k, 0), (6, "strict", k, 0), (6, "fast", k, 0),
                                        (4, "strict", 2, 0), (4, "fast", 2, 0), (4, "strict", 3, 12_500)):
                c = dict(CONFIGS[no])
                if shard:
                    c["B"] = shard
                    c["name"] += f" [one of 8 shards: {shard} barcodes]"
                if args.cells:
                    c["B"] = min(c["B"], args.cells)
                    c["name"] += f" [override: {c['B']} barcodes]"
                # every nested record is oracle-checked on its first barcode(s) too (~2 s of one host thread each; one cfg4 barcode is ~15 s, so
                # cfg4 gets one per kernel: the shard's STRICT k_doublet_clsp and the whole job's FAST one)
                par = 0.0 if args.no_cpu_baseline or (no == 4 and mode == "strict" and not shard) else 2.0
                r = run_config(cx, no, c, mode, kk, w, with_cpu=False, with_log=False,
                               with_e2e=((no == 6 and mode == "fast") and "e2e6") or (bool(shard) and "e2e4wp"), parity_s=par)
                if r.get("end_to_end"):
                    out["end_to_end"]["cfg6"] = r.pop("end_to_end")
                if r.get("end_to_end_write_pair"):            # cfg4's shard with --write-pair: the job, not only the kernels
                    out["end_to_end"]["cfg4_shard_write_pair"] = r.pop("end_to_end_write_pair")
                if shard:
                    r["config"]["tag"] = f"cfg{no}-shard/{mode}"
                also.append({key: r[key] for key in keys if key in r})
            out["also"] = also
        if cx.world > 1 and default_run and not args.only:
            # the sharded job once more in the opt-in FAST mode (every rank takes part: it ends with the same gather)
            r = run_config(cx, cfgno, cfg, "fast", max(2, min(args.steps, 5)), min(args.warmup, 1), with_cpu=False, with_log=False)
            if cx.rank == 0:
                out["also"] = [{key: r[key] for key in keys if key in r}]
        if cx.use_dist and ((default_run and not args.only) or args.e2e_write_pair) and cx.inputs is not None:
            # ... and the JOB, not only its kernels: BASELINE config 4 is "full doublet + --write-pair".  Every rank runs
            # dmx_demuxlet_run with write_pair on ITS range at the same time (its own files; the `.pair` rows are formatted on its GPU and go to
            # write(2)), so the host cores and the file system are shared as they would be in the real job; the record is rank 0's stage seconds
            # plus the slowest rank's total.
            _, _, g_r, dp_r = cx.inputs
            dist.barrier()
            wp = write_pair_leg(cx, cfg, dp_r, g_r, dp_r.n_cells, cfg["S"], cfg["V"])
            tot = torch.tensor([wp["strict"]["total_s"], wp["fast"]["total_s"]], dtype=torch.float64, device=cx.dev)
            dist.all_reduce(tot, op=dist.ReduceOp.MAX)
            if cx.rank == 0:
                wp["slowest_rank_total_s"] = {"strict": float(tot[0].item()), "fast": float(tot[1].item())}
                wp["pair_rows_all_ranks"] = wp["pair_rows"] * cx.world if cfg["B"] % cx.world == 0 else None
                out["end_to_end"] = {"cfg4_shard_write_pair": wp}
        cx.inputs = None
    if cx.use_dist or inject:
        dist.barrier()
        dist.destroy_process_group()
    if cx.rank == 0 and getattr(cx, "deferred_cpu", None):
        dp, g, c, rows, fast = cx.deferred_cpu
        attach_cpu_baseline(out, cpu_baseline(dp, g, c, rows=rows, fast=fast))
        cx.deferred_cpu = None
    if cx.rank == 0:
        full_path = Path(os.environ.get("DMX_BENCH_FULL", str(ROOT / "bench_full.json")))
        out["full_record"] = full_path.name
        try:
            full_path.write_text(json.dumps(out, indent=1) + "\n")
        except OSError as ex:
            out["full_record"] = f"not written: {ex!r}"
        sys.stderr.write("bench.py full record: " + json.dumps(out) + "\n")
        sys.stderr.flush()
        line = json.dumps(compact_line(out), separators=(",", ":"))
        assert len(line) < 6144, len(line)
        os.write(json_fd, (line + "\n").encode())


if __name__ == "__main__":
    main()
