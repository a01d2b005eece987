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
    assert lib.dmx_abi_version() == 6


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
