"""raytracer_amd — MI355X-native drop-in for the per-pixel render path of cwbriones/raytracer.

The product is the shared library ``librt_amd.so`` (C ABI in include/rt.h: scene loader, BVH build,
SoA flattening, HIP render kernels) and the ``raytracer`` CLI; this package is the ctypes binding used by
the tests and by bench.py."""
from .api import *  # noqa: F401,F403
from .api import lib, load_scene, example_scene, Scene, SceneDesc, tonemap, write_png  # noqa: F401
