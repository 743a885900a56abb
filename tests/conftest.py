import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ra():
    import raytracer_amd

    raytracer_amd.lib()  # fails loudly if librt_amd.so was not built
    return raytracer_amd


@pytest.fixture(scope="session")
def ref():
    from oracle import ref as _ref

    _ref.lib()
    return _ref


def scene_path(name):
    return os.path.join(SCENES, f"{name}.yaml")
