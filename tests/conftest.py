import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle as po
    po.build()
    return po


@pytest.fixture(scope="session")
def rt():
    import ray_tracer_amd as rt
    rt.load_library()
    return rt


@pytest.fixture(scope="session")
def scene_cache():
    """Finalised (BVH + lights) scenes shared by the tests of one session."""
    from ray_tracer_amd import scenes
    cache = {}

    def get(name, spatial=False, **kw):
        key = (name, spatial, tuple(sorted(kw.items())))
        if key not in cache:
            cache[key] = scenes.GENERATORS[name](**kw).finalize(spatial=spatial)
        return cache[key]
    return get
