"""ctypes wrapper of oracle/librt_ref.so — TEST INFRASTRUCTURE (tests/, smoke(), bench.py's cpu_baseline
leg only).  ``RefScene.from_yaml`` goes through oracle/loader_ref.py, i.e. nothing here touches product code."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import loader_ref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_ref.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        vp, dp = C.c_void_p, C.c_void_p
        L.ref_scene_create.restype = vp
        L.ref_scene_create.argtypes = [dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp,
                                       C.c_size_t, dp, dp, C.c_int]
        L.ref_scene_create_ext.restype = vp
        L.ref_scene_create_ext.argtypes = [dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp, C.c_size_t, dp,
                                           C.c_size_t, dp, dp, C.c_int, dp, dp, dp, dp, C.c_size_t, dp, dp, C.c_size_t,
                                           dp, C.c_size_t]
        L.ref_scene_create_nested.restype = vp
        L.ref_scene_create_nested.argtypes = L.ref_scene_create_ext.argtypes + [dp, C.c_size_t, dp, dp]
        L.ref_scene_destroy.argtypes = [vp]
        L.ref_leaf_order.argtypes = [vp, dp, dp]
        L.ref_camera.argtypes = [vp, dp]
        L.ref_primary_hits.argtypes = [vp, C.c_uint32, C.c_uint32, dp, dp, C.c_int]
        L.ref_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                 C.c_uint32, C.c_int, dp, dp, C.c_int]
        L.ref_sample.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, dp]
        L.ref_tonemap.argtypes = [dp, C.c_size_t, C.c_double, dp]
        L.ref_aabb_surface_area.restype = C.c_double
        L.ref_aabb_surface_area.argtypes = [dp, dp]
        L.ref_aabb_centroid.argtypes = [dp, dp, dp]
        L.ref_sphere_bbox.argtypes = [dp, C.c_double, dp]
        L.ref_aabb_hit.restype = C.c_int
        L.ref_aabb_hit.argtypes = [dp, dp, dp, dp, C.c_double, C.c_double]
        L.ref_trace.restype = C.c_int
        L.ref_trace.argtypes = [vp, dp, dp, C.c_double, C.c_double, dp]
        L.ref_scatter.restype = C.c_int
        L.ref_scatter.argtypes = [C.c_int, dp, C.c_double, dp, dp, C.c_int, C.c_uint64, dp]
        L.ref_reflectance.restype = C.c_double
        L.ref_reflectance.argtypes = [C.c_double, C.c_double]
        L.ref_sphere_uv.argtypes = [dp, dp]
        L.ref_rng_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, dp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data


def _d(x):
    return np.ascontiguousarray(x, dtype=np.float64)


class RefScene:
    def __init__(self, flat: "loader_ref.FlatScene", use_bvh=True):
        self.flat = flat
        m, s, t, q = _d(flat.materials), _d(flat.spheres), _d(flat.triangles), _d(flat.quads)
        sf = np.ascontiguousarray(flat.surfaces, dtype=np.int32)
        bg, cam = _d(flat.background), _d(flat.camera)
        self.n_surfaces = len(sf)
        ext = getattr(flat, "extras", None)
        if ext is None:
            self._h = lib().ref_scene_create(_p(m), len(m), _p(s), len(s), _p(t), len(t), _p(q), len(q), _p(sf), len(sf),
                                             _p(bg), _p(cam), 1 if use_bvh else 0)
        else:
            me, se = _d(ext["mat_ext"]), _d(ext["sphere_end"])
            pv = _d(ext["perlin_vec"])
            pp = np.ascontiguousarray(ext["perlin_perm"], dtype=np.int32)
            ib = np.ascontiguousarray(ext["image_bytes"], dtype=np.uint8)
            idm = np.ascontiguousarray(ext["image_dims"], dtype=np.int64)
            md = _d(ext["media"])
            # nesting of the tree (rt_bvh_group): groups n x 8 doubles, a group per surface, build boxes n x 6 or None
            gr = _d(ext["groups"]).reshape(-1, 8) if ext.get("groups") is not None else np.zeros((0, 8))
            sg = np.ascontiguousarray(ext["surface_group"], dtype=np.int32) if len(gr) else np.zeros(0, np.int32)
            bb = _d(ext["surface_build_box"]).reshape(-1, 6) if len(gr) and ext.get("surface_build_box") is not None else None
            assert not len(gr) or len(sg) == len(sf)
            self._keep = (me, se, pv, pp, ib, idm, md, gr, sg, bb)
            self._h = lib().ref_scene_create_nested(_p(m), len(m), _p(s), len(s), _p(t), len(t), _p(q), len(q), _p(sf),
                                                    len(sf), _p(bg), _p(cam), 1 if use_bvh else 0, _p(me), _p(se), _p(pv),
                                                    _p(pp), len(pp), _p(ib), _p(idm), len(idm), _p(md), len(md),
                                                    _p(gr) if len(gr) else None, len(gr), _p(sg) if len(gr) else None,
                                                    _p(bb) if bb is not None else None)
        if not self._h:
            raise RuntimeError("ref_scene_create failed (bad surface index)")

    @classmethod
    def from_yaml(cls, path, aspect_ratio, use_bvh=True):
        return cls(loader_ref.load_scene(path, aspect_ratio), use_bvh)

    @classmethod
    def from_desc(cls, desc, camera_params, use_bvh=True):
        """Oracle over the arrays of a product SceneDesc (procedural example scenes have no file to re-load).
        `camera_params` = [vfov_deg, aspect, from xyz, towards xyz, aperture, has_focus, focus_dist]: the oracle
        builds its own camera from them (camera.rs:43-74), so the product's camera is checked, not trusted."""
        flat = loader_ref.FlatScene()
        m, s, t, q = desc.materials, desc.spheres, desc.triangles, desc.quads
        flat.materials = np.column_stack([m["kind"].astype(np.float64), m["albedo"], m["param"]]).reshape(-1, 5)
        flat.spheres = np.column_stack([s["center"], s["radius"], s["material"].astype(np.float64)]).reshape(-1, 5)
        flat.triangles = (np.column_stack([t["v"].reshape(len(t), 9), t["n"].reshape(len(t), 9),
                                           t["material"].astype(np.float64)]).reshape(-1, 19)
                          if len(t) else np.zeros((0, 19)))
        flat.quads = (np.column_stack([q["q"], q["u"], q["v"], q["material"].astype(np.float64)]).reshape(-1, 10)
                      if len(q) else np.zeros((0, 10)))
        sf = desc.surfaces
        flat.surfaces = np.column_stack([sf["type"], sf["index"]]).astype(np.int32)
        flat.background = np.array(desc.background, dtype=np.float64)
        flat.camera = np.array(camera_params, dtype=np.float64)
        images = desc.images
        offs, dims = 0, []
        for im in images:
            dims.append([offs, im.shape[1] if im.size else 0, im.shape[0] if im.size else 0])
            offs += im.size
        pl = desc.perlin
        flat.extras = dict(
            mat_ext=np.column_stack([m["texture"].astype(np.float64), m["albedo2"], m["tex_scale"],
                                     m["tex_index"].astype(np.float64)]).reshape(-1, 6),
            sphere_end=np.array(s["center1"], dtype=np.float64).reshape(-1, 3),
            perlin_vec=np.array(pl["ranvec"], dtype=np.float64).reshape(-1, 256, 3),
            perlin_perm=np.array(pl["perm"], dtype=np.int32).reshape(-1, 3, 256),
            image_bytes=(np.concatenate([im.reshape(-1) for im in images]) if offs else np.zeros(1, np.uint8)),
            image_dims=np.array(dims, dtype=np.int64).reshape(-1, 3),
            media=np.column_stack([desc.media[k].astype(np.float64) for k in ("boundary", "first", "count", "material",
                                                                               "density")]).reshape(-1, 5))
        g = desc.groups
        if len(g):
            flat.extras["groups"] = np.column_stack([g["parent"].astype(np.float64), g["leaf"].astype(np.float64), g["box"]])
            flat.extras["surface_group"] = desc.surface_group
            flat.extras["surface_build_box"] = desc.surface_build_box
        return cls(flat, use_bvh)

    def close(self):
        if getattr(self, "_h", None):
            lib().ref_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def leaf_order(self):
        out = np.empty(self.n_surfaces, dtype=np.int32)
        st = np.zeros(4, dtype=np.uint32)
        lib().ref_leaf_order(self._h, _p(out), _p(st))
        return out, dict(nodes=int(st[0]), leaves=int(st[1]), depth=int(st[2]), max_leaf=int(st[3]))

    def camera(self):
        out = np.zeros(22)
        lib().ref_camera(self._h, _p(out))
        names = ["origin", "lower_left", "horizontal", "vertical", "u", "v", "w"]
        d = {n: out[3 * i:3 * i + 3].copy() for i, n in enumerate(names)}
        d["lens_radius"] = float(out[21])
        return d

    def primary_hits(self, width, height, threads=8):
        ids = np.empty((height, width), dtype=np.int32)
        t = np.empty((height, width), dtype=np.float64)
        lib().ref_primary_hits(self._h, width, height, _p(ids), _p(t), threads)
        return ids, t

    def render(self, width, height, sn, max_depth=40, seed=1, stratum_begin=0, stratum_step=1, threads=8,
               sampler=0):
        """sampler 0 = the reference's rejection loops (util.rs:15-35), 1 = the equal-distribution direct maps."""
        out = np.empty((height, width, 3), dtype=np.float64)
        cnt = np.zeros(5, dtype=np.uint64)
        lib().ref_render(self._h, width, height, sn, max_depth, seed, stratum_begin, stratum_step, threads, _p(out),
                         _p(cnt), sampler)
        names = ["box_tests", "tri_tests", "sphere_tests", "quad_tests", "segments"]
        return out, {n: int(c) for n, c in zip(names, cnt)}

    def trace(self, o, d, tmin=0.001, tmax=np.inf):
        o, d, out = _d(o), _d(d), np.zeros(8)
        sid = lib().ref_trace(self._h, _p(o), _p(d), tmin, tmax, _p(out))
        if sid < 0:
            return None
        return dict(surface=sid, t=out[0], point=out[1:4].copy(), normal=out[4:7].copy(), front_face=bool(out[7]))


def tonemap(rgb_sum, n_samples):
    a = _d(rgb_sum)
    out = np.empty(a.shape, dtype=np.uint8)
    lib().ref_tonemap(_p(a), a.size, float(n_samples), _p(out))
    return out


def scatter(kind, albedo, param, direction, normal, front_face, rng_state):
    a, d, n, out = _d(albedo), _d(direction), _d(normal), np.zeros(6)
    ok = lib().ref_scatter(kind, _p(a), param, _p(d), _p(n), 1 if front_face else 0, rng_state, _p(out))
    return (out[:3].copy(), out[3:].copy()) if ok else None


def sample(which, sampler, state, n):
    """which: 0 in_unit_sphere, 1 unit(in_unit_sphere), 2 in_unit_disk."""
    out = np.zeros((n, 3))
    lib().ref_sample(which, sampler, state, n, _p(out))
    return out


def sphere_uv(n):
    """get_sphere_uv (surfaces.rs:108-118) of a unit outward normal."""
    v, out = _d(n), np.zeros(2)
    lib().ref_sphere_uv(_p(v), _p(out))
    return float(out[0]), float(out[1])


def rng_stream(seed, pixel, stratum, n):
    out = np.zeros(n)
    lib().ref_rng_stream(seed, pixel, stratum, n, _p(out))
    return out
