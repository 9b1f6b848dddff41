import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: build it on demand (gcc only, a few seconds)
    so = ROOT / "oracle" / "librt3_oracle.so"
    srcs = [ROOT / "oracle" / n for n in ("rt3_oracle.c", "rt3_oracle_probes.c", "rt3_oracle.h", "Makefile")]
    if not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in srcs):
        subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def has_gpu():
    import torch

    return torch.cuda.is_available()
