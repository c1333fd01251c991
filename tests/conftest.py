"""Shared fixtures.  `-m "not gpu"` runs everywhere; `-m gpu` needs one MI355X."""
import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name is not a Python identifier).  A native library that is missing or older
    than its sources is built first (what __graft_entry__.build() does): a fresh checkout must not fail on that."""
    mod = importlib.import_module("2048_q-learning_amd")
    mod._native.build_host()       # the CPU twin (g++ only): a no-op when csrc/libq2048_host.so is newer than its sources
    try:
        mod._native.hipcc_path()
    except FileNotFoundError:
        # a host with neither hipcc nor a GPU runs the CPU-twin and oracle suites all the same; anything that
        # needs libq2048_hip.so fails loudly when it loads it (`_native.lib()` raises, nothing substitutes for it)
        return mod
    mod._native.build()            # likewise for csrc/libq2048_hip.so
    return mod


def load_npz(name):
    # materialise once: NpzFile re-reads the archive on every [] access
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    return load_npz


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
