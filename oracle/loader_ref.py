"""numpy restatement of the reference's scene loader (src/scene/load.rs) and mesh normal generation
(src/surfaces.rs:146-169) — TEST INFRASTRUCTURE, not product code (see oracle/README in DESIGN.md §Oracle).

It is independent of the product's C++ loader: YAML via PyYAML ``safe_load`` (standing in for serde_yaml),
OBJ text parsed as f32 then widened (tobj yields f32 positions, load.rs:258-261).  All geometry arithmetic is
IEEE f64 in the reference's operation order, so baked vertices and normals are comparable bit for bit with
the product loader's output.  Rotation uses the reference's own quaternion product (geom.rs:454-469), which
the reference's tests pin (geom.rs:493-562; restated in tests/test_oracle_kat.py).
"""
from __future__ import annotations

import gzip
import math
import os

import numpy as np
import yaml

LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT = 0, 1, 2, 3
SPHERE, TRIANGLE, QUAD = 0, 1, 2


# ---------------------------------------------------------------- geom.rs:351-470
def quat_new(x, y, z, w):
    inv_mag = 1.0 / math.sqrt(x * x + y * y + z * z + w * w)
    return (inv_mag * x, inv_mag * y, inv_mag * z, inv_mag * w)


def quat_rotation(axis, angle):
    c, s = math.cos(angle / 2.0), math.sin(angle / 2.0)
    return quat_new(s * axis[0], s * axis[1], s * axis[2], c)


def quat_mul(a, b):
    """Works on scalars or numpy arrays per component; term-for-term geom.rs:454-469."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    ww = (az + ax) * (bx + by)
    yy = (aw - ay) * (bw + bz)
    zz = (aw + ay) * (bw - bz)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (az - ax) * (bx - by))
    w = qq - ww + (az - ay) * (by - bz)
    x = qq - xx + (ax + aw) * (bx + bw)
    y = qq - yy + (aw - ax) * (by + bz)
    z = qq - zz + (az + ay) * (bw - bx)
    return (x, y, z, w)


def quat_rotate_vec(q, v):
    """rotate_vec, geom.rs:424-434; v is (N,3) or (3,)."""
    v = np.asarray(v, dtype=np.float64)
    conj = (-q[0], -q[1], -q[2], q[3])
    zeros = np.zeros_like(v[..., 0])
    lhs = quat_mul((v[..., 0], v[..., 1], v[..., 2], zeros), conj)
    out = quat_mul(q, lhs)
    return np.stack(np.broadcast_arrays(out[0], out[1], out[2]), axis=-1)


def quat_rotate_point(q, origin, p):
    return origin + quat_rotate_vec(q, p - origin)  # geom.rs:437-439


# ---------------------------------------------------------------- load.rs:332-344
def compute_mesh_center(pts):
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    max_v = np.maximum(np.zeros(3), pts.max(axis=0))  # max starts at (0,0,0), not -inf (quirk Q6)
    min_v = pts.min(axis=0)
    return min_v + 0.5 * (max_v - min_v)


def _num(x):
    if isinstance(x, bool):
        raise ValueError("expected a number")
    if isinstance(x, (int, float)):
        return float(x)
    if isinstance(x, str):  # PyYAML's YAML-1.1 resolver leaves e.g. "-.45" a string; serde_yaml reads a float
        return float(x)
    raise ValueError(f"expected a number, got {x!r}")


def _vec(x):
    if not isinstance(x, (list, tuple)) or len(x) != 3:
        raise ValueError(f"expected 3 numbers, got {x!r}")
    return np.array([_num(c) for c in x], dtype=np.float64)


def parse_albedo(a):
    """load.rs:471-522: 6 hex digits (each byte / 256) or a float triple."""
    if isinstance(a, (list, tuple)):
        return _vec(a)
    if not isinstance(a, str):
        raise ValueError("albedo: expected string or array of floats")
    if len(a) != 6:
        raise ValueError(f"expected hex color format aabbcc, got {a}")
    v = int(a, 16)
    return np.array([((v >> 16) & 255) / 256.0, ((v >> 8) & 255) / 256.0, (v & 255) / 256.0])


def parse_material(m):
    t = m["type"]
    if t == "lambertian":
        return (LAMBERTIAN, parse_albedo(m["albedo"]), 0.0)
    if t == "dielectric":
        return (DIELECTRIC, np.ones(3), _num(m["index"]))
    if t == "metal":
        return (METAL, parse_albedo(m["albedo"]), _num(m["fuzz"]))
    if t == "diffuseLight":
        return (DIFFUSE_LIGHT, parse_albedo(m["albedo"]), 0.0)
    raise ValueError(f"unknown material {t}")


def _axis(a):
    if a == "x":
        return np.array([1.0, 0.0, 0.0])
    if a == "y":
        return np.array([0.0, 1.0, 0.0])
    if a == "z":
        return np.array([0.0, 0.0, 1.0])
    return _vec(a)


def read_obj_first_model(path):
    """tobj::load_obj_buf, default options, first model: f32 positions re-indexed in order of first use."""
    opener = gzip.open if path.endswith(".gz") else open
    pos, nrm, faces, nfaces = [], [], [], []
    with opener(path, "rt") as f:
        for line in f:
            tok = line.split()  # tobj: the keyword is the first whitespace-separated token of the line
            key = tok[0] if tok else ""
            if key == "v":
                pos.append(tok[1:4])
            elif key == "vn":
                nrm.append(tok[1:4])
            elif key == "f":
                toks = [t.split("/") for t in tok[1:]]
                if len(toks) != 3:
                    raise ValueError("only triangles are supported")
                # 1-based, or negative = relative to the statements read SO FAR (tobj resolves an index where it parses the face)
                faces.append([int(t[0]) - 1 if int(t[0]) > 0 else len(pos) + int(t[0]) for t in toks])
                if len(toks[0]) == 3 and toks[0][2]:
                    nfaces.append([int(t[2]) - 1 if int(t[2]) > 0 else len(nrm) + int(t[2]) for t in toks])
            elif key in ("o", "g") and faces:  # a bare `g` / `o` line too
                break  # only the first model is used (load.rs:225-229)
    pos32 = np.array(pos, dtype=np.float32)  # numpy parses decimal -> f32 with correct rounding

    def first_use(ids, n):
        ids = np.array(ids, dtype=np.int64)
        flat = ids.reshape(-1)
        _, first = np.unique(flat, return_index=True)
        order = flat[np.sort(first)]  # ids in order of first use (export_faces_multi_index)
        remap = np.full(n, -1, dtype=np.int64)
        remap[order] = np.arange(len(order))
        return order, remap[ids]

    order, faces = first_use(faces, len(pos32))
    normals = None
    if nfaces:  # normals are exported in order of THEIR first use, by their own index map
        nrm32 = np.array(nrm, dtype=np.float32)
        norder, _ = first_use(nfaces, len(nrm32))
        normals = nrm32[norder].astype(np.float64)
    return pos32[order].astype(np.float64), faces, normals


def load_mesh(path, transforms):
    """load.rs:202-313 + Mesh::new surfaces.rs:146-169.  Returns (v[T,3,3], n[T,3,3])."""
    verts, faces, given_normals = read_obj_first_model(path)
    mesh_origin = compute_mesh_center(verts)
    verts = verts - mesh_origin
    for t in transforms or []:
        k = t["type"]
        if k == "scale":
            verts = _num(t["factor"]) * verts
            mesh_origin = compute_mesh_center(verts)
        elif k == "rotate":
            q = quat_rotation(_axis(t["axis"]), _num(t["angle"]) * (math.pi / 180.0))
            verts = quat_rotate_point(q, mesh_origin, verts)  # about the PRE-centring centre (Q6)
        elif k == "translate":
            d = _vec(t["dir"])
            verts = verts + d
            mesh_origin = mesh_origin + d
        elif k == "translateTo":
            d = _vec(t["dest"]) - mesh_origin
            verts = verts + d
            mesh_origin = mesh_origin + d
        else:
            raise ValueError(f"unknown transform {k}")
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    if given_normals is not None:
        # Mesh::new_with_normals (load.rs:303-311): the file's normals as they are — no transform turns them — and
        # looked up by the POSITION index (surfaces.rs:231-243)
        if len(given_normals) < len(verts):
            raise ValueError("mesh has fewer normals than vertices")
        tri_v = np.stack([v0, v1, v2], axis=1)
        tri_n = np.stack([given_normals[faces[:, 0]], given_normals[faces[:, 1]], given_normals[faces[:, 2]]], axis=1)
        return tri_v, tri_n, mesh_origin
    e1, e2 = v1 - v0, v0 - v2
    # cross(e2, e1), component formulas of geom.rs:182-188
    fn = np.stack([e2[:, 1] * e1[:, 2] - e2[:, 2] * e1[:, 1],
                   e2[:, 2] * e1[:, 0] - e2[:, 0] * e1[:, 2],
                   e2[:, 0] * e1[:, 1] - e2[:, 1] * e1[:, 0]], axis=1)
    normals = np.zeros_like(verts)
    # sequential accumulation in face order, vertex slot order (np.add.at is unbuffered and in order)
    np.add.at(normals, faces.reshape(-1), np.repeat(fn, 3, axis=0))
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = normals[:, 0] * normals[:, 0] + normals[:, 1] * normals[:, 1] + normals[:, 2] * normals[:, 2]
        normals = 1.0 * (normals / np.sqrt(sq)[:, None])
    tri_v = np.stack([v0, v1, v2], axis=1)
    tri_n = np.stack([normals[faces[:, 0]], normals[faces[:, 1]], normals[faces[:, 2]]], axis=1)
    return tri_v, tri_n, mesh_origin


def _quad_points_transform(q, u, v, transforms, origin):
    """transform_quad, load.rs:162-200.  Returns the new (q, u, v)."""
    p = np.stack([q, q + u, q + v])
    origin = origin.copy()
    for t in transforms or []:
        k = t["type"]
        if k == "scale":
            p = origin + _num(t["factor"]) * (p - origin)
        elif k == "rotate":
            qt = quat_rotation(_axis(t["axis"]), _num(t["angle"]) * (math.pi / 180.0))
            p = quat_rotate_point(qt, origin, p)
        elif k == "translate":
            d = _vec(t["dir"])
            p = p + d
            origin = origin + d
        elif k == "translateTo":
            d = _vec(t["dest"]) - origin
            p = p + d
            origin = origin + d
    return p[0], p[1] - p[0], p[2] - p[0]


class FlatScene:
    """Arrays in the layout oracle/rt_ref.cpp's ref_scene_create takes."""

    def __init__(self):
        self.materials, self.spheres, self.triangles, self.quads, self.surfaces = [], [], [], [], []
        self.background = np.zeros(3)
        self.camera = np.zeros(11)
        self.mesh_origins = []

    def finish(self):
        def arr(rows, width):
            return (np.array(rows, dtype=np.float64).reshape(-1, width) if len(rows)
                    else np.zeros((0, width), dtype=np.float64))
        self.materials = arr(self.materials, 5)
        self.spheres = arr(self.spheres, 5)
        self.triangles = (np.concatenate(self.triangles, axis=0) if len(self.triangles)
                          else np.zeros((0, 19), dtype=np.float64))
        self.quads = arr(self.quads, 10)
        self.surfaces = np.array(self.surfaces, dtype=np.int32).reshape(-1, 2)
        return self


def load_scene(path, aspect_ratio):
    """load_scene, load.rs:39-160."""
    with open(path) as f:
        cfg = yaml.safe_load(f)
    scene = cfg["scene"]
    if "background" not in scene:
        raise ValueError("missing field `background`")  # load.rs:379
    if not scene.get("surfaces"):
        raise ValueError("scene is empty")
    out = FlatScene()
    out.background = _vec(scene["background"])
    base = os.path.dirname(path)

    def add_material(m):
        kind, alb, param = parse_material(m)
        out.materials.append([kind, alb[0], alb[1], alb[2], param])
        return len(out.materials) - 1

    n_tri = 0
    for s in scene["surfaces"]:
        t = s["type"]
        if t == "mesh":
            mat = add_material(s["material"])
            tv, tn, origin = load_mesh(os.path.join(base, s["path"]), s.get("transform"))
            out.mesh_origins.append(origin)
            block = np.concatenate([tv.reshape(-1, 9), tn.reshape(-1, 9), np.full((len(tv), 1), float(mat))], axis=1)
            out.triangles.append(block)
            out.surfaces.extend([[TRIANGLE, n_tri + i] for i in range(len(tv))])
            n_tri += len(tv)
        elif t == "sphere":
            c = _vec(s["position"])
            r = _num(s["radius"])
            mat = add_material(s["material"])
            out.surfaces.append([SPHERE, len(out.spheres)])
            out.spheres.append([c[0], c[1], c[2], r, mat])
        elif t == "quad":
            q = _vec(s["position"])
            u, v = _vec(s["sides"][0]), _vec(s["sides"][1])
            mat = add_material(s["material"])
            out.surfaces.append([QUAD, len(out.quads)])
            out.quads.append(list(q) + list(u) + list(v) + [mat])
        elif t == "box":
            a, b = _vec(s["corners"][0]), _vec(s["corners"][1])
            mat = add_material(s["material"])
            lo, hi = np.minimum(a, b), np.maximum(a, b)
            dx = np.array([hi[0] - lo[0], 0.0, 0.0])
            dy = np.array([0.0, hi[1] - lo[1], 0.0])
            dz = np.array([0.0, 0.0, hi[2] - lo[2]])
            quads = [
                (np.array([lo[0], lo[1], hi[2]]), dx, dy), (np.array([hi[0], lo[1], hi[2]]), -dz, dy),
                (np.array([hi[0], lo[1], lo[2]]), -dx, dy), (np.array([lo[0], lo[1], lo[2]]), dz, dy),
                (np.array([lo[0], hi[1], hi[2]]), dx, -dz), (np.array([lo[0], lo[1], lo[2]]), dx, dz),
            ]
            pts = []
            for (q, u, v) in quads:
                pts += [q, q + u + v]
            origin = compute_mesh_center(np.array(pts))
            for (q, u, v) in quads:
                q2, u2, v2 = _quad_points_transform(q, u, v, s.get("transform"), origin)
                out.surfaces.append([QUAD, len(out.quads)])
                out.quads.append(list(q2) + list(u2) + list(v2) + [mat])
        else:
            raise ValueError(f"unknown surface {t}")
    cam = cfg["camera"]
    fd = cam.get("focus_distance")
    out.camera = np.array([_num(cam["fov"]), aspect_ratio, *_vec(cam["from"]), *_vec(cam["towards"]),
                           _num(cam.get("aperture") or 0.0), 1.0 if fd is not None else 0.0,
                           _num(fd) if fd is not None else 0.0], dtype=np.float64)
    return out.finish()


# ---------------------------------------------------------------- scene/example.rs: the examples that draw no
# random numbers — two-spheres (:208-234), cornell (:262-329), cornell-smoke (:331-403), earth (:405-438) —
# restated independently of the product's generator (raytracer_amd/csrc/host/examples.cpp), so that
# tests/test_examples.py can compare rt_example_scene's arrays with these bit for bit.
def _make_box(a, b):
    """make_box, surfaces.rs:542-590: the six quads (q, u, v) front, right, back, left, top, bottom."""
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    dx = np.array([hi[0] - lo[0], 0.0, 0.0])
    dy = np.array([0.0, hi[1] - lo[1], 0.0])
    dz = np.array([0.0, 0.0, hi[2] - lo[2]])
    P = lambda x, y, z: np.array([x, y, z], dtype=np.float64)
    return [(P(lo[0], lo[1], hi[2]), dx, dy), (P(hi[0], lo[1], hi[2]), -dz, dy), (P(hi[0], lo[1], lo[2]), -dx, dy),
            (P(lo[0], lo[1], lo[2]), dz, dy), (P(lo[0], hi[1], hi[2]), dx, -dz), (P(lo[0], lo[1], lo[2]), dx, dz)]


def _pad_box(lo, hi, delta=0.0001):
    """Aabb::pad, trace.rs: an axis narrower than delta is widened to delta about its middle."""
    lo, hi = lo.copy(), hi.copy()
    for a in range(3):
        if hi[a] - lo[a] < delta:
            lo[a], hi[a] = lo[a] - delta / 2.0, hi[a] + delta / 2.0
    return lo, hi


def _turned_box(a, b, degrees, offset):
    """Translated(Rotated(make_box(a, b), +y, degrees), offset) as world-space quads: Rotated turns about the
    centroid of the box's bounding box — the merge of its quads' padded boxes (surfaces.rs:356-359, 507-517) —
    and the hit point / normal are turned back (surfaces.rs:520-534), which is the same as turning the quads;
    Translated then shifts them (surfaces.rs:480-491)."""
    quads = _make_box(np.asarray(a, float), np.asarray(b, float))
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for q, u, v in quads:
        far = q + u + v
        blo, bhi = _pad_box(np.minimum(q, far), np.maximum(q, far))
        lo, hi = np.minimum(lo, blo), np.maximum(hi, bhi)
    centre = lo + 0.5 * (hi - lo)  # Aabb::centroid, trace.rs
    rot = quat_rotation(np.array([0.0, 1.0, 0.0]), math.radians(degrees))
    out = []
    for q, u, v in quads:
        p0 = quat_rotate_point(rot, centre, q) + offset
        p1 = quat_rotate_point(rot, centre, q + u) + offset
        p2 = quat_rotate_point(rot, centre, q + v) + offset
        out.append((p0, p1 - p0, p2 - p0))
    return out


def _turned_box_bounds(a, b, degrees, offset):
    """Bounding box of Translated::new(Rotated::new(make_box(a, b), +y, degrees), offset) — what the scene's own BVH
    build sees of the box: the leaf's bound (the merge of the quads' padded boxes), its eight corners turned about its
    midpoint (Aabb::rotate, trace.rs:191-211), shifted (Aabb::translate, trace.rs:185-189)."""
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for q, u, v in _make_box(np.asarray(a, float), np.asarray(b, float)):
        far = q + u + v
        blo, bhi = _pad_box(np.minimum(q, far), np.maximum(q, far))
        lo, hi = np.minimum(lo, blo), np.maximum(hi, bhi)
    delta = hi - lo
    mid = lo + 0.5 * delta
    rot = quat_rotation(np.array([0.0, 1.0, 0.0]), math.radians(degrees))
    mn, mx = np.full(3, np.inf), np.full(3, -np.inf)
    for ii in range(8):
        corner = lo + np.array([float(ii & 1), float((ii >> 1) & 1), float((ii >> 2) & 1)]) * delta
        r = quat_rotate_point(rot, mid, corner)
        mn, mx = np.minimum(mn, r), np.maximum(mx, r)
    return np.concatenate([mn + offset, mx + offset])


def example_scene(name):
    """-> dict(camera=[fov, from, towards, aperture, focus or None], background, surfaces=[...], media=[...]).
    A surface is ("sphere", centre, radius, material) or ("quad", q, u, v, material); a medium is
    (boundary quads [(q, u, v)...], density, albedo); a material is (kind, albedo, param, texture, albedo2, scale)
    with texture in {"solid", "uv_checker", "image"}."""
    solid = lambda kind, albedo, param=0.0: (kind, tuple(albedo), param, "solid", (0.0, 0.0, 0.0), 0.0)
    V = lambda *c: np.array(c, dtype=np.float64)
    if name == "two-spheres":
        checker = (LAMBERTIAN, (0.2, 0.3, 0.1), 0.0, "uv_checker", (0.9, 0.9, 0.9), 30.0)
        return dict(camera=[20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 0.0, None], background=(0.7, 0.8, 1.0),
                    surfaces=[("sphere", V(0, -10, 0), 10.0, checker), ("sphere", V(0, 10, 0), 10.0, checker)], media=[])
    if name == "earth":
        # ./scenes/res/earth.png is not in the reference's repository: earth_texture() falls back to the solid
        # colour (0, 1, 1) (example.rs:430-438); the two Rotated wrappers turn a sphere about its own centre
        globe = (LAMBERTIAN, (0.0, 1.0, 1.0), 0.0, "image", (0.0, 0.0, 0.0), 1.0)
        return dict(camera=[20.0, (0.0, 0.0, 12.0), (0.0, 0.0, 0.0), 0.0, None], background=(0.7, 0.8, 1.0),
                    surfaces=[("sphere", V(0, 0, 0), 2.0, globe)], media=[])
    if name in ("cornell", "cornell-smoke"):
        smoke = name == "cornell-smoke"
        red, white, green = solid(LAMBERTIAN, (0.65, 0.05, 0.05)), solid(LAMBERTIAN, (0.73, 0.73, 0.73)), solid(LAMBERTIAN, (0.12, 0.45, 0.15))
        light = solid(DIFFUSE_LIGHT, (7.0, 7.0, 7.0) if smoke else (15.0, 15.0, 15.0))
        lq = (V(113, 554, 127), V(330, 0, 0), V(0, 0, 305)) if smoke else (V(343, 554, 332), V(-130, 0, 0), V(0, 0, -105))
        surfaces = [("quad", V(555, 0, 0), V(0, 555, 0), V(0, 0, 555), green),
                    ("quad", V(0, 0, 0), V(0, 555, 0), V(0, 0, 555), red),
                    ("quad", *lq, light),
                    ("quad", V(0, 555, 0), V(555, 0, 0), V(0, 0, 555), white),
                    ("quad", V(0, 0, 0), V(555, 0, 0), V(0, 0, 555), white),
                    ("quad", V(0, 0, 555), V(555, 0, 0), V(0, 555, 0), white)]
        box1 = _turned_box((0.0, 0.001, 0.0), (165.0, 333.0, 165.0), 15.0, V(265, 0, 295))
        box2 = _turned_box((0.0, 0.001, 0.0), (165.0, 165.0, 165.0), -18.0, V(130, 0, 65))
        media, groups, surface_group = [], [], [-1] * 6
        if smoke:  # ConstantMedium::new(box, 0.01, colour), example.rs:387-398
            media = [(box1, 0.01, (0.0, 0.0, 0.0)), (box2, 0.01, (1.0, 1.0, 1.0))]
        else:
            surfaces += [("quad", q, u, v, white) for q, u, v in box1 + box2]
            # each box is a Surface::Bvh of ONE leaf (make_box -> Bvh::new_leaf) wrapped and added to the scene's list
            # (example.rs:310-326): groups (parent, is one leaf, box in the scene's build), a group per surface
            groups = [(-1, 1, _turned_box_bounds((0.0, 0.001, 0.0), (165.0, 333.0, 165.0), 15.0, V(265, 0, 295))),
                      (-1, 1, _turned_box_bounds((0.0, 0.001, 0.0), (165.0, 165.0, 165.0), -18.0, V(130, 0, 65)))]
            surface_group += [0] * 6 + [1] * 6
        return dict(camera=[40.0, (278.0, 278.0, -800.0), (278.0, 278.0, 0.0), 0.0, None], background=(0.0, 0.0, 0.0),
                    surfaces=surfaces, media=media, groups=groups, surface_group=surface_group)
    raise ValueError(f"{name}: not one of the examples that draw no random numbers")
