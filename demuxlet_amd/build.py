"""Builds demuxlet_amd/libdmx.so (HIP kernels + C-ABI + host finaliser) in-tree with hipcc for gfx950.

    python -m demuxlet_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract (STRICT mode reproduces the
reference's separate multiply/add roundings) — do not remove it."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libdmx.so"
SOURCES = [CSRC / "dmx_host.cpp", CSRC / "dmx_engine.hip"]
HEADERS = [ROOT / "include" / "dmx.h", CSRC / "dmx_internal.hpp", CSRC / "dmx_log.hpp"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=default",
         "-Wall", "-Wno-unused-function", f"-I{ROOT / 'include'}", f"-I{CSRC}"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


CLI = PKG / "demuxlet"
CLI_SRC = CSRC / "dmx_cli.cpp"


def build_cli(force: bool = False, verbose: bool = False) -> Path:
    """The `demuxlet` command-line front end (host C++ only; links libdmx.so and zlib)."""
    deps = [CLI_SRC, CSRC / "dmx_inflate.hpp", ROOT / "include" / "dmx.h", LIB]
    if not force and CLI.exists() and CLI.stat().st_mtime >= max(p.stat().st_mtime for p in deps):
        return CLI
    cmd = [hipcc(), "-O2", "-std=c++17", "-Wall", "-x", "c++", f"-I{ROOT / 'include'}", str(CLI_SRC), "-o", str(CLI),
           f"-L{PKG}", "-ldmx", "-lz", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI


def build(force: bool = False, verbose: bool = False) -> Path:
    if force or needs_build():
        cmd = [hipcc(), *FLAGS, "-x", "hip", *map(str, SOURCES), "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    build_cli(force, verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
