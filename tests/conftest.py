import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("lio-slam_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("lio-slam_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle, build
    so = os.environ.get("LIO_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liblio_oracle.so")   # e.g. an ASAN/UBSAN build
    if not os.path.exists(so):
        build(fast=False)
    return Oracle(so)


@pytest.fixture(scope="session")
def small_case(synth):
    """VLP-16 scan vs a 6-keyframe map (CPU ray casting, ~2 s)."""
    return synth.make_case("vlp16", n_keyframes=6, seed=11, device="cpu", n_queries=3)
