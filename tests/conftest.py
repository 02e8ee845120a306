import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The tests that go through RCCL run last: a communicator set-up hang on a box (tests/rccl_record.py keeps its
    record) must not stand between `-x` and the rest of the suite."""
    late = [it for it in items if "rccl" in it.name.lower()]
    if late:
        items[:] = [it for it in items if "rccl" not in it.name.lower()] + late


@pytest.fixture(scope="session")
def c1_problem():
    from ceres_slam_amd import synth
    return synth.make_config("C1")


@pytest.fixture(scope="session")
def tiny_problem():
    from ceres_slam_amd import synth
    return synth.make_problem(8, 60, track_len=5, seed=7)
