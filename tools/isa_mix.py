#!/usr/bin/env python3
"""Issue-cycle weights of the engine's kernels from their ISA (written by tools/kernel_resources.sh to /tmp/dmx_asm/dmx_engine.s) and the
per-instruction issue costs measured by tools/micro/valu_rate.hip (profiles/r03_valu_rate.json): a wave64 VALU instruction holds its
SIMD's issue port for 4 cycles when it is FP64, a conversion to/from FP64, v_cmp_class_f64, v_mov_b64, any three-source integer
operation (v_and_or_b32, v_lshl_add_u32, v_add3_u32, v_bfe_u32, v_alignbit_b32, v_perm_b32, v_mad_*), a VOPC compare, or carries a DPP /
SDWA modifier; for 2 cycles when it is a plain one- or two-source 32-bit operation (add, and, or, shifts, v_mov_b32, FP32 add/mul/fma).
PMC counters give exact dynamic counts of the FP64 classes but lump everything else together, so the average cost of a NON-FP64 VALU
instruction is taken from the kernel's own loop bodies: every basic block weighted by 16^(loop depth).

    python tools/isa_mix.py 'k_doublet_a2<256, 4, 4, true, false>' ... > profiles/r03_isa_mix.json"""
import json
import re
import subprocess
import sys

FOUR = re.compile(r"(_f64|^v_cvt_.*f64|^v_mov_b64|^v_and_or_b32|^v_lshl_add_u32|^v_add3_u32|^v_lshl_or_b32|^v_or3_b32|^v_bfe_|^v_bfi_b32|^v_alignbit_b32|"
                  r"^v_perm_b32|^v_mad_|^v_mul_lo_u32|^v_mul_hi_u32|^v_cmp_|^v_cmpx_|_dpp$|_sdwa$|^v_lshl_add_u64|^v_readlane|^v_readfirstlane|^v_writelane|"
                  r"^v_cndmask_b32_e64|^v_med3|^v_min3|^v_max3|^v_xad_u32|^v_sad_)")
FP64 = re.compile(r"^v_(add|mul|fma|fmac|max|min|rcp|rsq|sqrt|ldexp|frexp_mant|trunc|floor|ceil|rndne|fract|div_scale|div_fmas|div_fixup)_f64")


def kernels(path="/tmp/dmx_asm/dmx_engine.s"):
    txt = open(path).read().split("\n")
    labels = [(i, l.split(":")[0]) for i, l in enumerate(txt) if l.startswith("_ZN") and ":" in l]
    dem = subprocess.run(["c++filt"], input="\n".join(l for _, l in labels), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for (i, _), d in zip(labels, dem):
        d = d.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        j = next(k for k in range(i, len(txt)) if txt[k].startswith(".Lfunc_end"))
        out[d] = txt[i:j]
    return out


def mix(body):
    depth = 0
    acc = dict(fp64=0.0, other=0.0, other_cycles=0.0, lds=0.0, vmem=0.0, salu=0.0)
    hot = dict(acc)
    for ln in body:
        m = re.match(r"^\.LBB\d+_\d+:\s*;.*Depth=(\d+)", ln)
        if re.match(r"^\.LBB\d+_\d+:", ln):
            m2 = re.search(r"Depth=(\d+)", ln)
            depth = int(m2.group(1)) if m2 else 0
            continue
        m3 = re.match(r"^\s*;\s*(=>)?\s*(This |Parent )?.*Depth=(\d+)", ln)
        if m3 and "This" in ln:
            depth = max(depth, int(m3.group(3)))
            continue
        t = ln.strip()
        if not t or t[0] in ";." or t.startswith("s_nop"):
            continue
        op = t.split()[0]
        w = 16.0 ** depth
        if op.startswith("v_"):
            if FP64.match(op):
                acc["fp64"] += w
            else:
                acc["other"] += w
                acc["other_cycles"] += w * (4 if FOUR.search(op) else 2)
        elif op.startswith("ds_"):
            acc["lds"] += w
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            acc["vmem"] += w
        elif op.startswith("s_"):
            acc["salu"] += w
    return acc


if __name__ == "__main__":
    ks = kernels()
    res = {}
    for name in sys.argv[1:]:
        if name not in ks:
            sys.exit("no such kernel in the assembly: " + name)
        a = mix(ks[name])
        res[name] = {"cycles_per_other_valu_inst": a["other_cycles"] / max(a["other"], 1e-9), "fp64_per_other_static": a["fp64"] / max(a["other"], 1e-9),
                     "lds_per_valu_static": a["lds"] / max(a["fp64"] + a["other"], 1e-9)}
    json.dump(res, sys.stdout, indent=1)
    print()
