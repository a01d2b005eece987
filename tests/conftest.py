import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


# the tests force launch geometries and kernel variants through DMX_* experiment switches; the library honours those only when
# DMX_EXPERIMENTS=1 (read once per engine, at dmx_engine_create) — tests/test_gpu_parity.py::test_experiment_switches_are_fenced checks the fence
os.environ.setdefault("DMX_EXPERIMENTS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py
