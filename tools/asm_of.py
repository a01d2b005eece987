#!/usr/bin/env python3
"""usage: tools/asm_of.py 'k_doublet_sym<64, 32, 4, true, 3, 0>' [out]  -> the ISA of one kernel from /tmp/dmx_asm/dmx_engine.s
(tools/kernel_resources.sh writes that file), plus an instruction-class histogram of its longest loop body."""
import re, subprocess, sys
name = sys.argv[1]
txt = open("/tmp/dmx_asm/dmx_engine.s").read().split("\n")
labels = [(i, l[:-1].split(":")[0]) for i, l in enumerate(txt) if l.startswith("_ZN") and ":" in l]
dem = subprocess.run(["c++filt"], input="\n".join(l for _, l in labels), capture_output=True, text=True).stdout.split("\n")
for (i, l), d in zip(labels, dem):
    d = d.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if d == name:
        j = next(k for k in range(i, len(txt)) if txt[k].startswith(".Lfunc_end"))
        body = txt[i:j]
        out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/dmx_asm/kernel.s"
        open(out, "w").write("\n".join(body))
        print(f"{d}: {len(body)} lines -> {out}")
        break
else:
    sys.exit("not found: " + name)
