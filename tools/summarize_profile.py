#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs of tools/profile_round.sh into profiles/<tag>_cfg<N>_{kernel_stats.csv,pmc_summary.json}
and writes profiles/pmc_cfg<N>.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import json
import sys
from pathlib import Path

src = Path(sys.argv[1])
tag = sys.argv[2]
cfg = sys.argv[3]
out = Path(__file__).resolve().parents[1] / "profiles"
out.mkdir(exist_ok=True)
rows = [r for r in csv.DictReader(open(src / "kernel_stats.csv")) if "(anonymous namespace)::k_" in r["Name"]]
with open(out / f"{tag}_cfg{cfg}_kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        w.writerow(r)
summ = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for p in sorted(src.glob("pmc_*.csv")):
    for r in csv.DictReader(open(p)):
        if "(anonymous namespace)::k_" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        summ[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = dict(vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"])
res = {}
for k, cs in summ.items():
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["_dispatch"] = meta[k]
    if "FETCH_SIZE" in res[k]:
        # rocprofv3 units: KiB.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
        # coalesced stream (x2 correction); narrower accesses are uncalibrated, so both figures are kept.
        res[k]["hbm_read_bytes_raw"] = res[k]["FETCH_SIZE"] * 1024
        res[k]["hbm_read_bytes_x2"] = res[k]["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in res[k]:
        res[k]["hbm_write_bytes"] = res[k]["WRITE_SIZE"] * 1024
json.dump(res, open(out / f"{tag}_cfg{cfg}_pmc_summary.json", "w"), indent=1)
dom = [k for k in res if "k_doublet" in k] or [k for k in res if "k_singlet" in k]
dom = max(dom, key=lambda k: res[k].get("SQ_WAVE_CYCLES", 0))
d = res[dom]
bench = json.load(open(src / "bench.json"))
json.dump({"kernel": dom, "barcodes_per_gpu": bench["config"]["barcodes_per_gpu"],
           "valu_wave_insts_per_launch": d.get("SQ_INSTS_VALU"), "valu_busy_quadcycles_per_launch": d.get("SQ_ACTIVE_INST_VALU"),
           "hbm_bytes_per_launch": d.get("hbm_read_bytes_x2", 0) + d.get("hbm_write_bytes", 0),
           "hbm_read_bytes_raw": d.get("hbm_read_bytes_raw"), "hbm_read_bytes_x2": d.get("hbm_read_bytes_x2"),
           "hbm_write_bytes": d.get("hbm_write_bytes"), "source": f"profiles/{tag}_cfg{cfg}_pmc_summary.json"},
          open(out / f"pmc_cfg{cfg}.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
