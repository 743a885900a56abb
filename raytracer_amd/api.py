"""ctypes binding of librt_amd.so (include/rt.h) and a thin host-side mirror of the reference's
render-path interface: ``load_scene(path, aspect_ratio, use_bvh)`` (load.rs:39), a ``Scene`` whose
``render`` replaces the pixel loop main.rs:150-185, ``tonemap`` (main.rs:178-180) and ``write_png``
(main.rs:192-200).  Python here is plumbing only: every computation happens in the shared library.
The library is required — there is no fallback path."""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")  # override: A/B of kernel builds

RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = 0, 1, 2, 3, 4
RT_SURF_SPHERE, RT_SURF_TRIANGLE, RT_SURF_QUAD, RT_SURF_MEDIUM = 0, 1, 2, 3
RT_KERNEL_PERSISTENT, RT_KERNEL_WAVEFRONT, RT_KERNEL_POOL = 0, 1, 2
RT_DEVICE_CURRENT, RT_DEVICE_NONE = -1, -2
RT_SAMPLER_REFERENCE, RT_SAMPLER_DIRECT = 0, 1
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_UV_CHECKER, RT_TEX_MARBLE, RT_TEX_IMAGE = 0, 1, 2, 3, 4

# every symbol include/rt.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "rt_load_scene", "rt_example_scene", "rt_scene_desc_free", "rt_scene_create", "rt_scene_destroy", "rt_scene_get_info",
    "rt_scene_leaf_order", "rt_render_device", "rt_render_device_noalloc", "rt_render", "rt_primary_hits", "rt_tonemap_rgb8",
    "rt_tonemap_rgb8_device", "rt_group_create", "rt_group_reduce_sum", "rt_group_destroy", "rt_write_png", "rt_set_verbose", "rt_last_error", "rt_version",
]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", C.c_double * 3), ("param", C.c_double),
                ("albedo2", C.c_double * 3), ("tex_scale", C.c_double), ("tex_index", C.c_int32), ("_pad", C.c_int32)]


class RtSphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("material", C.c_int32), ("moving", C.c_int32),
                ("center1", C.c_double * 3)]


class RtPerlin(C.Structure):
    _fields_ = [("ranvec", (C.c_double * 3) * 256), ("perm", (C.c_int32 * 256) * 3)]


class RtMedium(C.Structure):
    _fields_ = [("boundary", C.c_int32), ("first", C.c_int32), ("count", C.c_int32), ("material", C.c_int32),
                ("density", C.c_double)]


class RtBvhGroup(C.Structure):
    _fields_ = [("parent", C.c_int32), ("leaf", C.c_int32), ("box", C.c_double * 6)]


class RtImage(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("width", C.c_uint32), ("height", C.c_uint32)]


class RtTriangle(C.Structure):
    _fields_ = [("v", (C.c_double * 3) * 3), ("n", (C.c_double * 3) * 3), ("material", C.c_int32),
                ("_pad", C.c_int32)]


class RtQuad(C.Structure):
    _fields_ = [("q", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("normal", C.c_double * 3),
                ("d", C.c_double), ("w", C.c_double * 3), ("material", C.c_int32), ("_pad", C.c_int32)]


class RtSurfaceRef(C.Structure):
    _fields_ = [("type", C.c_int32), ("index", C.c_int32)]


class RtCamera(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w")] + [
        ("lens_radius", C.c_double)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("materials", C.POINTER(RtMaterial)), ("n_materials", C.c_size_t),
        ("spheres", C.POINTER(RtSphere)), ("n_spheres", C.c_size_t),
        ("triangles", C.POINTER(RtTriangle)), ("n_triangles", C.c_size_t),
        ("quads", C.POINTER(RtQuad)), ("n_quads", C.c_size_t),
        ("surfaces", C.POINTER(RtSurfaceRef)), ("n_surfaces", C.c_size_t),
        ("background", C.c_double * 3), ("camera", RtCamera), ("use_bvh", C.c_int32), ("_pad", C.c_int32),
        ("perlin", C.POINTER(RtPerlin)), ("n_perlin", C.c_size_t),
        ("images", C.POINTER(RtImage)), ("n_images", C.c_size_t),
        ("media", C.POINTER(RtMedium)), ("n_media", C.c_size_t),
        ("groups", C.POINTER(RtBvhGroup)), ("n_groups", C.c_size_t),
        ("surface_group", C.POINTER(C.c_int32)), ("surface_build_box", C.POINTER(C.c_double)),
    ]


class RtSceneInfo(C.Structure):
    _fields_ = [("n_surfaces", C.c_uint64), ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64),
                ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32), ("device_bytes", C.c_uint64),
                ("build_ms", C.c_double), ("render_nodes", C.c_uint64), ("render_depth", C.c_uint32),
                ("render_max_leaf", C.c_uint32), ("n_loose", C.c_uint32), ("device_build", C.c_uint32),
                ("render_build_ms", C.c_double)]


class RtRenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sn", C.c_uint32), ("max_depth", C.c_uint32),
                ("seed", C.c_uint64), ("stratum_begin", C.c_uint32), ("stratum_step", C.c_uint32),
                ("kernel", C.c_int32), ("collect_stats", C.c_int32), ("sampler", C.c_int32), ("_pad", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("quad_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("n_launches", C.c_uint32), ("_pad", C.c_uint32)]


# numpy record layouts of the flat scene arrays
MATERIAL_DTYPE = np.dtype([("kind", "<i4"), ("texture", "<i4"), ("albedo", "<f8", 3), ("param", "<f8"),
                           ("albedo2", "<f8", 3), ("tex_scale", "<f8"), ("tex_index", "<i4"), ("_pad", "<i4")])
SPHERE_DTYPE = np.dtype([("center", "<f8", 3), ("radius", "<f8"), ("material", "<i4"), ("moving", "<i4"),
                         ("center1", "<f8", 3)])
MEDIUM_DTYPE = np.dtype([("boundary", "<i4"), ("first", "<i4"), ("count", "<i4"), ("material", "<i4"), ("density", "<f8")])
PERLIN_DTYPE = np.dtype([("ranvec", "<f8", (256, 3)), ("perm", "<i4", (3, 256))])
TRIANGLE_DTYPE = np.dtype([("v", "<f8", (3, 3)), ("n", "<f8", (3, 3)), ("material", "<i4"), ("_pad", "<i4")])
QUAD_DTYPE = np.dtype([("q", "<f8", 3), ("u", "<f8", 3), ("v", "<f8", 3), ("normal", "<f8", 3), ("d", "<f8"),
                       ("w", "<f8", 3), ("material", "<i4"), ("_pad", "<i4")])
SURFACE_DTYPE = np.dtype([("type", "<i4"), ("index", "<i4")])
GROUP_DTYPE = np.dtype([("parent", "<i4"), ("leaf", "<i4"), ("box", "<f8", 6)])


class RtError(RuntimeError):
    pass


_lib = None


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so, and a process that first loads
    /opt/rocm's copy (through librt_amd.so) and then imports torch finds no GPU in torch.  When torch is
    installed but not imported yet, map its runtime first so that librt_amd.so binds to the same one."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for root in (spec.submodule_search_locations or []) if spec else []:
        path = os.path.join(root, "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return


def lib() -> C.CDLL:
    """Load librt_amd.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(or `make -C raytracer_amd/csrc`). There is no fallback path.")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
    L.rt_load_scene.argtypes = [C.c_char_p, dbl, i32, C.POINTER(C.POINTER(RtSceneDesc))]
    L.rt_load_scene.restype = i32
    L.rt_example_scene.argtypes = [C.c_char_p, dbl, C.c_uint64, i32, C.POINTER(C.POINTER(RtSceneDesc))]
    L.rt_example_scene.restype = i32
    L.rt_scene_desc_free.argtypes = [C.POINTER(RtSceneDesc)]
    L.rt_scene_desc_free.restype = None
    L.rt_scene_create.argtypes = [C.POINTER(RtSceneDesc), i32, C.POINTER(vp)]
    L.rt_scene_create.restype = i32
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_destroy.restype = None
    L.rt_scene_get_info.argtypes = [vp, C.POINTER(RtSceneInfo)]
    L.rt_scene_get_info.restype = i32
    L.rt_scene_leaf_order.argtypes = [vp, vp]
    L.rt_scene_leaf_order.restype = i32
    L.rt_render_device.argtypes = [vp, C.POINTER(RtRenderParams), vp, vp, C.POINTER(RtStats)]
    L.rt_render_device.restype = i32
    L.rt_render_device_noalloc.argtypes = [vp, C.POINTER(RtRenderParams), vp, vp]
    L.rt_render_device_noalloc.restype = i32
    L.rt_render.argtypes = [vp, C.POINTER(RtRenderParams), vp, C.POINTER(RtStats)]
    L.rt_render.restype = i32
    L.rt_primary_hits.argtypes = [vp, u32, u32, i32, vp, vp]
    L.rt_primary_hits.restype = i32
    L.rt_tonemap_rgb8.argtypes = [vp, u32, u32, u64, vp]
    L.rt_tonemap_rgb8.restype = i32
    L.rt_tonemap_rgb8_device.argtypes = [vp, u32, u32, u64, vp, vp]
    L.rt_tonemap_rgb8_device.restype = i32
    L.rt_group_create.argtypes = [C.POINTER(C.c_int), i32, C.POINTER(vp)]
    L.rt_group_create.restype = i32
    L.rt_group_reduce_sum.argtypes = [vp, C.POINTER(vp), C.c_size_t, i32, C.POINTER(vp)]
    L.rt_group_reduce_sum.restype = i32
    L.rt_group_destroy.argtypes = [vp]
    L.rt_group_destroy.restype = None
    L.rt_write_png.argtypes = [C.c_char_p, vp, u32, u32]
    L.rt_write_png.restype = i32
    L.rt_set_verbose.argtypes = [i32]
    L.rt_set_verbose.restype = None
    L.rt_last_error.restype = C.c_char_p
    L.rt_version.restype = C.c_char_p
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise RtError(lib().rt_last_error().decode("utf-8", "replace"))


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=dtype, count=n).copy()  # a copy: the library frees the source with the desc


class SceneDesc:
    """Flat host scene: the output of the loader, the input of Scene (and of the oracle in tests)."""

    def __init__(self, ptr=None, keepalive=None):
        self._ptr = ptr          # POINTER(RtSceneDesc) owned by the library, or None
        self._keep = keepalive   # arrays backing a Python-built description
        self._struct = ptr.contents if ptr is not None else None

    # -- construction from numpy arrays (synthetic scenes in tests)
    @classmethod
    def from_arrays(cls, materials, spheres, triangles, quads, surfaces, background, camera: RtCamera,
                    use_bvh=True, perlin=None, images=(), media=None, groups=None, surface_group=None,
                    surface_build_box=None) -> "SceneDesc":
        """`perlin`: array of PERLIN_DTYPE; `images`: sequence of uint8 arrays of shape (h, w, 3); `groups`
        (GROUP_DTYPE), `surface_group` (int32 per surface), `surface_build_box` (6 doubles per surface or None):
        nesting of the reference-topology tree, rt_bvh_group in include/rt.h."""
        arrs = dict(
            materials=np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE),
            spheres=np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE),
            triangles=np.ascontiguousarray(triangles, dtype=TRIANGLE_DTYPE),
            quads=np.ascontiguousarray(quads, dtype=QUAD_DTYPE),
            surfaces=np.ascontiguousarray(surfaces, dtype=SURFACE_DTYPE),
        )
        d = RtSceneDesc()
        for name, typ in (("materials", RtMaterial), ("spheres", RtSphere), ("triangles", RtTriangle),
                          ("quads", RtQuad), ("surfaces", RtSurfaceRef)):
            a = arrs[name]
            setattr(d, name, C.cast(a.ctypes.data, C.POINTER(typ)))
            setattr(d, "n_" + name, len(a))
        d.background = (C.c_double * 3)(*background)
        d.camera = camera
        d.use_bvh = _bvh_mode(use_bvh)
        arrs["perlin"] = np.ascontiguousarray(perlin if perlin is not None else np.zeros(0, PERLIN_DTYPE), dtype=PERLIN_DTYPE)
        d.perlin = C.cast(arrs["perlin"].ctypes.data, C.POINTER(RtPerlin))
        d.n_perlin = len(arrs["perlin"])
        arrs["image_pixels"] = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        arrs["images"] = (RtImage * max(len(images), 1))()
        for i, im in enumerate(arrs["image_pixels"]):
            arrs["images"][i].rgb = C.cast(im.ctypes.data, C.POINTER(C.c_uint8)) if im.size else None
            arrs["images"][i].height, arrs["images"][i].width = (im.shape[0], im.shape[1]) if im.size else (0, 0)
        d.images = C.cast(arrs["images"], C.POINTER(RtImage))
        d.n_images = len(images)
        arrs["media"] = np.ascontiguousarray(media if media is not None else np.zeros(0, MEDIUM_DTYPE), dtype=MEDIUM_DTYPE)
        d.media = C.cast(arrs["media"].ctypes.data, C.POINTER(RtMedium))
        d.n_media = len(arrs["media"])
        if groups is not None and len(groups):
            arrs["groups"] = np.ascontiguousarray(groups, dtype=GROUP_DTYPE)
            arrs["surface_group"] = np.ascontiguousarray(surface_group, dtype=np.int32)
            assert len(arrs["surface_group"]) == len(arrs["surfaces"])
            d.groups = C.cast(arrs["groups"].ctypes.data, C.POINTER(RtBvhGroup))
            d.n_groups = len(arrs["groups"])
            d.surface_group = C.cast(arrs["surface_group"].ctypes.data, C.POINTER(C.c_int32))
            if surface_build_box is not None:
                arrs["surface_build_box"] = np.ascontiguousarray(surface_build_box, dtype=np.float64).reshape(-1, 6)
                assert len(arrs["surface_build_box"]) == len(arrs["surfaces"])
                d.surface_build_box = C.cast(arrs["surface_build_box"].ctypes.data, C.POINTER(C.c_double))
        self = cls(None, arrs)
        self._struct = d
        return self

    @property
    def struct(self) -> RtSceneDesc:
        return self._struct

    materials = property(lambda s: _view(s._struct.materials, s._struct.n_materials, MATERIAL_DTYPE))
    spheres = property(lambda s: _view(s._struct.spheres, s._struct.n_spheres, SPHERE_DTYPE))
    triangles = property(lambda s: _view(s._struct.triangles, s._struct.n_triangles, TRIANGLE_DTYPE))
    quads = property(lambda s: _view(s._struct.quads, s._struct.n_quads, QUAD_DTYPE))
    surfaces = property(lambda s: _view(s._struct.surfaces, s._struct.n_surfaces, SURFACE_DTYPE))
    perlin = property(lambda s: _view(s._struct.perlin, s._struct.n_perlin, PERLIN_DTYPE))
    media = property(lambda s: _view(s._struct.media, s._struct.n_media, MEDIUM_DTYPE))
    groups = property(lambda s: _view(s._struct.groups, s._struct.n_groups, GROUP_DTYPE))

    @property
    def surface_group(self):
        """int32 per surface (-1: the scene's own list), or None when the scene does not nest."""
        st = self._struct
        if not st.n_groups or not st.surface_group:
            return None
        return np.ctypeslib.as_array(st.surface_group, shape=(st.n_surfaces,)).copy()

    @property
    def surface_build_box(self):
        st = self._struct
        if not st.n_groups or not st.surface_build_box:
            return None
        return np.ctypeslib.as_array(st.surface_build_box, shape=(st.n_surfaces, 6)).copy()

    @property
    def images(self):
        out = []
        for i in range(self._struct.n_images):
            im = self._struct.images[i]
            n = 3 * im.width * im.height
            out.append(np.ctypeslib.as_array(im.rgb, shape=(n,)).reshape(im.height, im.width, 3).copy() if n
                       else np.zeros((0, 0, 3), np.uint8))
        return out
    background = property(lambda s: np.array(list(s._struct.background)))
    # (a copy: the struct lives in the library's memory and goes with the description)
    camera = property(lambda s: RtCamera.from_buffer_copy(s._struct.camera))
    n_surfaces = property(lambda s: int(s._struct.n_surfaces))
    use_bvh = property(lambda s: bool(s._struct.use_bvh))

    def close(self):
        if self._ptr is not None:
            lib().rt_scene_desc_free(self._ptr)
            self._ptr = None
            self._struct = None

    def __del__(self):
        # at interpreter shutdown the process's memory goes back anyway; do not call into a library whose HIP
        # runtime may already be unloading
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def _bvh_mode(use_bvh) -> int:
    """False / True as the reference's flag; "device" (or 2) additionally builds the render tree on the GPU."""
    if use_bvh == "device" or use_bvh == 2:
        return 2
    return 1 if use_bvh else 0


def load_scene(path: str, aspect_ratio: float, use_bvh=True) -> SceneDesc:
    """scene::load_scene(path, aspect_ratio, use_bvh) — load.rs:39-160."""
    out = C.POINTER(RtSceneDesc)()
    _check(lib().rt_load_scene(os.fsencode(path), float(aspect_ratio), _bvh_mode(use_bvh), C.byref(out)))
    return SceneDesc(out)


def example_scene(name: str, aspect_ratio: float, seed: int = 0x5EED5EED, use_bvh: bool = True) -> SceneDesc:
    """Example::scene — scene/example.rs:53-79 (all eight example names)."""
    out = C.POINTER(RtSceneDesc)()
    _check(lib().rt_example_scene(name.encode(), float(aspect_ratio), int(seed), _bvh_mode(use_bvh), C.byref(out)))
    return SceneDesc(out)


@dataclass
class Stats:
    samples: int
    segments: int
    node_visits: int
    tri_tests: int
    sphere_tests: int
    quad_tests: int
    kernel_ms: float
    n_launches: int


class Scene:
    """Device-resident scene (BVH built, flattened, uploaded once)."""

    def __init__(self, desc: SceneDesc, device: int = RT_DEVICE_CURRENT):
        self._h = C.c_void_p()
        _check(lib().rt_scene_create(C.byref(desc.struct), int(device), C.byref(self._h)))
        self.n_surfaces = desc.n_surfaces

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if sys.is_finalizing():  # see SceneDesc.__del__
            return
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> RtSceneInfo:
        i = RtSceneInfo()
        _check(lib().rt_scene_get_info(self._h, C.byref(i)))
        return i

    def leaf_order(self) -> np.ndarray:
        out = np.empty(self.n_surfaces, dtype=np.int32)
        _check(lib().rt_scene_leaf_order(self._h, out.ctypes.data))
        return out

    @staticmethod
    def _params(width, height, sn, max_depth, seed, stratum_begin, stratum_step, kernel, collect_stats, sampler):
        return RtRenderParams(width, height, sn, max_depth, seed, stratum_begin, stratum_step, kernel,
                              1 if collect_stats else 0, sampler, 0)

    def render(self, width, height, sn, *, max_depth=40, seed=1, stratum_begin=0, stratum_step=1,
               kernel=RT_KERNEL_PERSISTENT, collect_stats=False, sampler=RT_SAMPLER_DIRECT):
        """Pixel loop main.rs:150-185 for strata begin, begin+step, ...: returns (rgb_sum[H,W,3] f32, Stats)."""
        p = self._params(width, height, sn, max_depth, seed, stratum_begin, stratum_step, kernel, collect_stats,
                         sampler)
        out = np.empty((height, width, 3), dtype=np.float32)
        st = RtStats()
        _check(lib().rt_render(self._h, C.byref(p), out.ctypes.data, C.byref(st)))
        return out, _stats(st)

    def render_device(self, d_ptr: int, width, height, sn, *, max_depth=40, seed=1, stratum_begin=0,
                      stratum_step=1, kernel=RT_KERNEL_PERSISTENT, collect_stats=False, stream: int = 0,
                      want_stats=True, sampler=RT_SAMPLER_DIRECT):
        """Same, into device memory at ``d_ptr`` (e.g. a torch tensor's data_ptr()) on HIP stream ``stream``."""
        p = self._params(width, height, sn, max_depth, seed, stratum_begin, stratum_step, kernel, collect_stats,
                         sampler)
        st = RtStats()
        _check(lib().rt_render_device(self._h, C.byref(p), C.c_void_p(d_ptr), C.c_void_p(stream),
                                      C.byref(st) if want_stats else None))
        return _stats(st) if want_stats else None

    def render_device_noalloc(self, d_ptr: int, width, height, sn, *, max_depth=40, seed=1, stratum_begin=0,
                              stratum_step=1, stream: int = 0, sampler=RT_SAMPLER_DIRECT, kernel=RT_KERNEL_PERSISTENT):
        """render_device through the entry point that allocates nothing (static device scratch; persistent kernel,
        frames up to 32 768 tiles); enqueues on ``stream`` and returns."""
        p = self._params(width, height, sn, max_depth, seed, stratum_begin, stratum_step, kernel, False, sampler)
        _check(lib().rt_render_device_noalloc(self._h, C.byref(p), C.c_void_p(d_ptr), C.c_void_p(stream)))

    def primary_hits(self, width, height, precision=64):
        ids = np.empty((height, width), dtype=np.int32)
        t = np.empty((height, width), dtype=np.float64)
        _check(lib().rt_primary_hits(self._h, width, height, precision, ids.ctypes.data, t.ctypes.data))
        return ids, t


def _stats(st: RtStats) -> Stats:
    return Stats(st.samples, st.segments, st.node_visits, st.tri_tests, st.sphere_tests, st.quad_tests,
                 st.kernel_ms, st.n_launches)


def tonemap(rgb_sum: np.ndarray, n_samples: int) -> np.ndarray:
    """average + sqrt + clamp + quantise, main.rs:177-180."""
    a = np.ascontiguousarray(rgb_sum, dtype=np.float32)
    h, w, _ = a.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    _check(lib().rt_tonemap_rgb8(a.ctypes.data, w, h, int(n_samples), out.ctypes.data))
    return out


def write_png(path: str, rgb8: np.ndarray) -> None:
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = a.shape
    _check(lib().rt_write_png(os.fsencode(path), a.ctypes.data, w, h))


def set_verbose(v: bool) -> None:
    lib().rt_set_verbose(1 if v else 0)
