"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs golden vectors / vs oracle/_ref, host logic, C-ABI symbol checks (CPU only).
`-m gpu`       : parity tests proper; they call the HIP path through the C-ABI on cuda:0.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
PKG = ROOT / "llama.cpp.dsp_amd"
if str(PKG) not in sys.path:
    sys.path.insert(0, str(PKG))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
