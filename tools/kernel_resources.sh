#!/bin/bash
# usage: tools/kernel_resources.sh [regex]   -> registers, spills, LDS and scratch of every kernel whose name matches (device-only
# compile of dmx_engine.hip to assembly under /tmp/dmx_asm; the .s file stays there for reading the ISA)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/dmx_asm
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Idemuxlet_amd/csrc --cuda-device-only -S \
  -x hip demuxlet_amd/csrc/dmx_engine.hip -o /tmp/dmx_asm/dmx_engine.s "${@:2}"
python3 - "$1" <<'PY'
import re, sys
rx = re.compile(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else ".")
txt = open("/tmp/dmx_asm/dmx_engine.s").read()
md = txt[txt.index("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if not rx.search(dem): continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk)
    print(f"{dem:48s} vgpr {g('vgpr_count').group(1):>4} spill {g('vgpr_spill_count').group(1):>3} sgpr {g('sgpr_count').group(1):>4} "
          f"lds {g('group_segment_fixed_size').group(1):>6} scratch {g('private_segment_fixed_size').group(1):>5}")
PY
