import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# development aid: AGT_TEST_LIB=libagt_hip_knobs.so runs the suite on the knobs build (whose environment knobs select experimental
# kernels, e.g. AGT_PYR4=1); the product library is the default and what the driver tests
if os.environ.get("AGT_TEST_LIB"):
    from accurate_aprilgroup_tracking_amd import hiplib as _hiplib
    _hiplib.LIB_PATH = os.path.join(os.path.dirname(_hiplib.LIB_PATH), os.environ["AGT_TEST_LIB"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure only)."""
    from oracle import cvoracle
    cvoracle.build()
    return cvoracle


@pytest.fixture(scope="session")
def seq640():
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    return syn.Sequence(640, 480, n_tags=12, n_frames=6, seed=0)


@pytest.fixture(scope="session")
def seq720():
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    return syn.Sequence(1280, 720, n_tags=12, n_frames=4, seed=1)


@pytest.fixture(scope="session")
def seq640_dist():
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    return syn.Sequence(640, 480, n_tags=12, n_frames=4, seed=2, dist=syn.MILD_DIST)


@pytest.fixture(scope="session")
def seq1080():
    """BASELINE.json configs[3] geometry: one 1920x1080 stream (the 8-GPU config shards eight of them)."""
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    return syn.Sequence(1920, 1080, n_tags=12, n_frames=5, seed=4, supersample=2)


@pytest.fixture(scope="session")
def seq720_long():
    """BASELINE.json configs[1] geometry, 60 tracked frames: one 1280x720 stream, 12 tags / 48 corners"""
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    return syn.Sequence(1280, 720, n_tags=12, n_frames=61, seed=1)
