"""CPU: libdmx.so builds/loads and exports exactly the symbols include/dmx.h declares; no GPU compute is called."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from demuxlet_amd import build, capi
    build.build()
    return capi.load()


def declared_functions():
    text = (ROOT / "include" / "dmx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from demuxlet_amd import capi
    assert declared_functions() == sorted(capi.SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(ROOT / "demuxlet_amd" / "libdmx.so")], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name in declared_functions():
        assert name in exported, name
        getattr(lib, name)


def test_abi_version(lib):
    assert lib.dmx_abi_version() == 8


def test_by_pointer_input_structs_do_not_grow():
    """dmx_final_input is passed by pointer and has no size member, so a member added at its end is read past the end of
    an older caller's object.  ABI 6's trailing `cell_grid` was withdrawn in ABI 7 (the grids are an argument of
    dmx_write_doublet_summary_grids): the struct has its ABI 5 layout — it ends with tie_tol — in the header, the C compiler and the binding."""
    from demuxlet_amd import capi
    text = (ROOT / "include" / "dmx.h").read_text()
    body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} dmx_final_input;", text, flags=re.S).group(1)
    members = re.findall(r"(\w+)\s*;", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert members[-1] == "tie_tol" and "cell_grid" not in members
    assert capi.FinalInput._fields_[-1][0] == "tie_tol"
    src = '#include "dmx.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){printf("%zu %zu\\n", sizeof(dmx_final_input), offsetof(dmx_final_input, tie_tol));return 0;}\n'
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "s.c").write_text(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(Path(d) / "s.c"), "-o", str(Path(d) / "s")])
        size, off = map(int, subprocess.check_output([str(Path(d) / "s")], text=True).split())
    assert size == C.sizeof(capi.FinalInput) == off + 8 and off == capi.FinalInput.tie_tol.offset


def test_code_object_is_gfx950_only():
    so = ROOT / "demuxlet_amd" / "libdmx.so"
    data = so.read_bytes()
    assert b"gfx950" in data
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"sm_90"):
        assert other not in data


def test_engine_fails_loudly_without_gpu(lib):
    """There is no CPU fallback: on a box without a gfx950 device the engine refuses to exist."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from demuxlet_amd import capi
    from demuxlet_amd.engine import Engine
    with pytest.raises(capi.DmxError) as ei:
        Engine(4)
    assert ei.value.code in (-5, -2)
    assert b"no" in lib.dmx_last_error().lower() or b"fail" in lib.dmx_last_error().lower()


def test_product_never_touches_the_oracle():
    """Layout rule: nothing under demuxlet_amd/ or include/ may import, include, link or name oracle/."""
    for p in list((ROOT / "demuxlet_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.is_file() and p.suffix in (".py", ".cpp", ".hip", ".hpp", ".h"):
            txt = p.read_text()
            assert "oracle_py" not in txt and "dmx_oracle" not in txt and "liboracle" not in txt, p
            assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), p


def test_device_warm_up_without_a_gpu_reports_and_does_not_crash():
    """dmx_device_warm_up (ABI 5) on a box without a HIP device: a negative status and a message, like every other entry point that
    needs the GPU (this library has no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from demuxlet_amd import capi
    lib = capi.load()
    rc = lib.dmx_device_warm_up(0, 1)
    assert rc < 0
    assert b"no HIP device" in lib.dmx_last_error() or b"HIP" in lib.dmx_last_error()
