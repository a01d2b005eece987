#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs of tools/profile_round.sh into profiles/<tag>_cfg<N>[_fast]_{kernel_stats.csv,pmc_summary.json,
bench.json} and writes profiles/pmc_cfg<N>_<mode>.json (read by bench.py for roofline.traffic / roofline_valu).

    python tools/summarize_profile.py gpurun_out/prof_r02_cfg3 r02 3

Issue-cycle accounting: a wave64 VALU instruction holds its SIMD's issue port for 4 cycles when it is FP64
(16 lanes/clk: add/mul/fma/transcendental incl. v_rcp_f64) and for 2 cycles otherwise (32 lanes/clk: int32, FP32, moves,
compares, converts — MI355X_MICROARCH.md "v_fma_f32 (wave64) 2 cyc").  issue_cycles = 4 x FP64 + 2 x (SQ_INSTS_VALU - FP64).
v_cvt_f64_f32 is priced at 2 here (conservative: if it runs at the FP64 rate the true utilisation is higher; `issue_cycles_cvt4`
gives that variant)."""
import collections
import csv
import gzip
import json
import sys
from pathlib import Path

src = Path(sys.argv[1])
tag = sys.argv[2]
cfg = sys.argv[3]
out = Path(__file__).resolve().parents[1] / "profiles"
out.mkdir(exist_ok=True)
bench = json.load(open(src / "bench.json"))
mode = bench["config"].get("mode", "strict")
sfx = "" if mode == "strict" else f"_{mode}"
rows = [r for r in csv.DictReader(open(src / "kernel_stats.csv")) if "(anonymous namespace)::k_" in r["Name"]]
with open(out / f"{tag}_cfg{cfg}{sfx}_kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        w.writerow(r)
summ = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for p in sorted(list(src.glob("pmc_*.csv")) + list(src.glob("pmc_*.csv.gz"))):
    for r in csv.DictReader(gzip.open(p, "rt") if p.suffix == ".gz" else open(p)):
        if "(anonymous namespace)::k_" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        summ[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = dict(vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"])
# average issue cost of a NON-FP64 VALU instruction per kernel, from its ISA and the measured per-instruction costs (tools/isa_mix.py,
# tools/micro/valu_rate.hip); 3.0 where the kernel has no entry
try:
    isa_mix = json.load(open(out / f"{tag}_isa_mix.json"))
except OSError:
    isa_mix = {}
res = {}
for k, cs in summ.items():
    d = res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    d["_dispatch"] = meta[k]
    if "FETCH_SIZE" in d:
        # rocprofv3 units: KiB.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
        # coalesced stream (x2 correction); narrower accesses are uncalibrated, so both figures are kept.
        d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
        d["hbm_read_bytes_x2"] = d["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    f64 = [d.get(f"SQ_INSTS_VALU_{x}_F64") for x in ("ADD", "MUL", "FMA", "TRANS")]
    if all(v is not None for v in f64) and "SQ_INSTS_VALU" in d:
        d["fp64_insts"] = sum(f64)
        d["other_valu_insts"] = d["SQ_INSTS_VALU"] - d["fp64_insts"]
        d["issue_cycles"] = 4 * d["fp64_insts"] + 2 * d["other_valu_insts"]
        d["issue_cycles_cvt4"] = d["issue_cycles"] + 2 * d.get("SQ_INSTS_VALU_CVT", 0.0)
        w = isa_mix.get(k, {}).get("cycles_per_other_valu_inst", 3.0)
        d["cycles_per_other_valu_inst"] = w
        d["issue_cycles_measured_costs"] = 4 * d["fp64_insts"] + w * d["other_valu_insts"]
json.dump(res, open(out / f"{tag}_cfg{cfg}{sfx}_pmc_summary.json", "w"), indent=1)
json.dump(bench, open(out / f"{tag}_cfg{cfg}{sfx}_bench.json", "w"))
dom = [k for k in res if "k_doublet" in k] or [k for k in res if "k_singlet" in k]
dom = max(dom, key=lambda k: res[k].get("SQ_WAVE_CYCLES", 0))
d = dict(res[dom])
# k_doublet_a2u (round 6) leaves a barcode's 64 diagonal accumulators to k_doublet_diag (k_doublet_a2s<.., 0> in its first form), launched right behind it inside the same K2 event pair: the
# counts of one K2 launch are the two kernels' together (named in "kernel_group"; "kernel" stays the one dmx_engine_kernel_names reports)
group = [dom] + ([k for k in res if k.startswith("k_doublet_diag<") or (k.startswith("k_doublet_a2s<") and k.rstrip(">").endswith(", 0"))] if dom.startswith("k_doublet_a2u") else [])
for k in group[1:]:
    for c, v in res[k].items():
        if isinstance(v, (int, float)) and c != "cycles_per_other_valu_inst" and isinstance(d.get(c), (int, float)):
            d[c] = d[c] + v
json.dump({"kernel": dom, "kernel_group": group, "barcodes_per_gpu": bench["config"]["barcodes_per_gpu"], "mode": mode,
           "valu_wave_insts_per_launch": d.get("SQ_INSTS_VALU"), "valu_busy_quadcycles_per_launch": d.get("SQ_ACTIVE_INST_VALU"),
           "fp64_insts_per_launch": d.get("fp64_insts"), "other_valu_insts_per_launch": d.get("other_valu_insts"),
           "fp64_add_per_launch": d.get("SQ_INSTS_VALU_ADD_F64"), "fp64_mul_per_launch": d.get("SQ_INSTS_VALU_MUL_F64"),
           "fp64_fma_per_launch": d.get("SQ_INSTS_VALU_FMA_F64"), "fp64_trans_per_launch": d.get("SQ_INSTS_VALU_TRANS_F64"),
           "issue_cycles_per_launch": d.get("issue_cycles"), "issue_cycles_cvt4_per_launch": d.get("issue_cycles_cvt4"),
           "issue_cycles_measured_costs_per_launch": d.get("issue_cycles_measured_costs"), "cycles_per_other_valu_inst": d.get("cycles_per_other_valu_inst"),
           "lds_wait_inst_quadcycles_per_launch": d.get("SQ_WAIT_INST_LDS"), "wave_quadcycles_per_launch": d.get("SQ_WAVE_CYCLES"),
           "wait_any_quadcycles_per_launch": d.get("SQ_WAIT_ANY"), "wait_inst_any_quadcycles_per_launch": d.get("SQ_WAIT_INST_ANY"),
           "sq_inst_cycles_valu_per_launch": d.get("SQ_INST_CYCLES_VALU"),
           "lds_insts_per_launch": d.get("SQ_INSTS_LDS"), "lds_active_quadcycles_per_launch": d.get("SQ_ACTIVE_INST_LDS"),
           "lds_idx_active_per_launch": d.get("SQ_LDS_IDX_ACTIVE"), "lds_bank_conflict_per_launch": d.get("SQ_LDS_BANK_CONFLICT"),
           "grbm_gui_active_per_launch": d.get("GRBM_GUI_ACTIVE"),
           "hbm_bytes_per_launch": d.get("hbm_read_bytes_x2", 0) + d.get("hbm_write_bytes", 0),
           "hbm_read_bytes_raw": d.get("hbm_read_bytes_raw"), "hbm_read_bytes_x2": d.get("hbm_read_bytes_x2"),
           "hbm_write_bytes": d.get("hbm_write_bytes"), "source": f"profiles/{tag}_cfg{cfg}{sfx}_pmc_summary.json"},
          open(out / f"pmc_cfg{cfg}_{mode}.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:4000])
