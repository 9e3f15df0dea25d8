import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU; run on the GPU box with -m gpu")


@pytest.fixture(scope="session")
def scene_small():
    """Scene S0 with the 8,712-triangle dragon (9,680 triangles in all): the geometry of the reference's committed
    80kScene dump."""
    import harness
    return harness.rt.build_scene_s0(subdivide=False)


@pytest.fixture(scope="session")
def scene_80k():
    """Scene S0 with the dragon subdivided to 78,408 triangles (the "80kScene*" stand-in)."""
    import harness
    return harness.rt.build_scene_s0(subdivide=True)


@pytest.fixture(scope="session")
def gpu_ctx():
    import harness
    ctx = harness.rt.RtContext(0)
    yield ctx
    ctx.close()
