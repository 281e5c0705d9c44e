"""pytest configuration: the ``gpu`` marker + import path.

``-m "not gpu"`` runs on the CPU-only build container (oracle vs goldens, host
logic, C-ABI symbol table, gloo sharding); ``-m gpu`` runs on an MI355X box and
goes through the C-ABI of the HIP library.
"""

import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o
