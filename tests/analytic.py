"""Closed-form truths about small scenes, derived on paper from the reference's sources — NOT from the oracle.
The oracle (tests/test_analytic.py) and the HIP path (tests/test_gpu_analytic.py) are each held against these, so
a misreading shared by oracle/rt_ref.cpp and the kernels would show.  What each case pins is listed in DESIGN.md §2.

Camera maths (camera.rs:43-74, main.rs:157-173) is restated here in numpy on its own: pixel (col, row) with row 0
the top row sees, at its centre, u = (col + 1)/W, v = (H - row)/H  (pi + 1, pj + 1 with pj = H - row - 1: the
half-pixel shift Q5), direction = lower_left + u*horizontal + v*vertical - origin (not normalised).
"""
import math

import numpy as np

L_SKY = (0.6, 0.7, 0.9)


def camera_text(frm, to, fov, focus=None):
    s = f"camera:\n  fov: {fov}\n  from: [{frm[0]}, {frm[1]}, {frm[2]}]\n  towards: [{to[0]}, {to[1]}, {to[2]}]\n"
    if focus is not None:
        s += f"  focus_distance: {focus}\n"
    return s


def camera_rays(frm, to, fov, aspect, w, h, focus=None, sub=1, edges=False):
    """Directions of the pixels' sub-sample centres: array (h, w, sub, sub, 3), and the origin.  Camera::new with
    vup = (0, 1, 0), aperture 0; sub x sub points per pixel at the strata centres of the jitter window
    [-0.5, 0.5)^2 around (col + 1, H - row) (main.rs:161-173), or — `edges` — spanning the window border to border."""
    frm, to = np.asarray(frm, float), np.asarray(to, float)
    fd = np.linalg.norm(frm - to) if focus is None else float(focus)
    vh = 2.0 * math.tan(math.radians(fov) / 2.0)
    vw = aspect * vh
    wv = (frm - to) / np.linalg.norm(frm - to)
    uv = np.cross([0.0, 1.0, 0.0], wv)
    uv /= np.linalg.norm(uv)
    vv = np.cross(wv, uv)
    hor, ver = fd * vw * uv, fd * vh * vv
    ll = frm - hor / 2 - ver / 2 - fd * wv
    cols, rows = np.meshgrid(np.arange(w), np.arange(h))
    offs = np.linspace(-0.5, 0.5, sub) if edges else (np.arange(sub) + 0.5) / sub - 0.5
    u = (cols[..., None, None] + 1 + offs[None, None, :, None]) / w
    v = (h - rows[..., None, None] + offs[None, None, None, :]) / h
    u, v = np.broadcast_arrays(u, v)
    d = ll + u[..., None] * hor + v[..., None] * ver - frm
    return frm, d


# ---------------------------------------------------------------- furnace: one convex body under a uniform sky
def furnace_sphere_yaml(tmp_path, material, name="furnace"):
    """Unit sphere at the origin seen from (0, 0, 5), fov 40, sky L_SKY, nothing else.  Every path that hits a
    convex body leaves it after one scatter (Lambertian: n + unit vector points outward, material.rs:99-118; metal
    with fuzz 0: the mirror direction, material.rs:125-138; dielectric: in, k internal reflections, out,
    material.rs:140-169) and ends in the sky (scene/mod.rs:122), so every sample is albedo * L exactly."""
    text = (camera_text((0, 0, 5), (0, 0, 0), 40)
            + f"scene:\n  background: [{L_SKY[0]}, {L_SKY[1]}, {L_SKY[2]}]\n  surfaces:\n"
            + f"    - type: sphere\n      radius: 1\n      position: [0, 0, 0]\n      material: {material}\n")
    p = tmp_path / f"{name}.yaml"
    p.write_text(text)
    return str(p)


def furnace_sphere_coverage(aspect=1.0, fov=40.0, dist=5.0, r=1.0):
    """Fraction of the view window the sphere's silhouette covers: a circle of radius r / sqrt(D^2 - r^2) on the
    plane at distance 1 (the tangent cone has half-angle asin(r / D)); the window is 2 tan(fov/2) high."""
    vh = 2.0 * math.tan(math.radians(fov) / 2.0)
    return math.pi * r * r / (dist * dist - r * r) / (vh * vh * aspect)


def furnace_sphere_masks(w, h, fov=40.0, dist=5.0, r=1.0):
    """(inside, outside): pixels whose whole jitter window lies inside / outside the silhouette (with a margin of
    one pixel)."""
    _, d = camera_rays((0, 0, dist), (0, 0, 0), fov, w / h, w, h)
    d = d[:, :, 0, 0]
    # on the plane z = dist - 1 ... use angles: tan of the angle to the axis
    tan_a = np.hypot(d[..., 0], d[..., 1]) / -d[..., 2]
    tan_s = r / math.sqrt(dist * dist - r * r)
    px = 2.0 * math.tan(math.radians(fov) / 2.0) / h  # one pixel on the plane at distance 1
    return tan_a < tan_s - 1.5 * px, tan_a > tan_s + 1.5 * px


def flat_mesh_yaml(tmp_path, n=24, name="flatmesh"):
    """A flat n x n grid of 2 n^2 triangles (one plane, so every area-weighted vertex normal is the plane's normal,
    surfaces.rs:146-169) facing the camera: a convex body made of triangles.  Inside its outline every sample is
    albedo * L; a ray slipping between two triangles would return L, a double hit albedo^2 * L."""
    lines = []
    for j in range(n + 1):
        for i in range(n + 1):
            lines.append(f"v {i / n * 2 - 1:.6f} {j / n * 2 - 1:.6f} 0")
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i + 1
            lines.append(f"f {a} {a + 1} {a + n + 2}")
            lines.append(f"f {a} {a + n + 2} {a + n + 1}")
    (tmp_path / f"{name}.obj").write_text("\n".join(lines) + "\n")
    text = (camera_text((0.3, 0.2, 5), (0, 0, 0), 40)
            + f"scene:\n  background: [{L_SKY[0]}, {L_SKY[1]}, {L_SKY[2]}]\n  surfaces:\n"
            + f"    - type: mesh\n      path: {name}.obj\n      material: {{type: lambertian, albedo: [0.8, 0.5, 0.2]}}\n")
    p = tmp_path / f"{name}.yaml"
    p.write_text(text)
    return str(p)


def flat_mesh_masks(w, h):
    """(inside, outside) of the square [-1, 1]^2 in the plane z = 0 seen from (0.3, 0.2, 5), margin of 1.5 pixels."""
    o, d = camera_rays((0.3, 0.2, 5), (0, 0, 0), 40, w / h, w, h)
    d = d[:, :, 0, 0]
    t = -o[2] / d[..., 2]
    x, y = o[0] + t * d[..., 0], o[1] + t * d[..., 1]
    m = 1.5 * 2.0 * math.tan(math.radians(20)) * 5.0 / h  # 1.5 pixels at the plane's distance
    inside = (np.abs(x) < 1 - m) & (np.abs(y) < 1 - m)
    outside = (np.abs(x) > 1 + m) | (np.abs(y) > 1 + m)
    return inside, outside


# ---------------------------------------------------------------- direct light of a sphere on a Lambertian plane
LIGHT = dict(frm=(0, 3, 6), to=(0, 0, 0), fov=20, c=(0.0, 2.0, 0.0), r=0.5, le=8.0, rho=(0.7, 0.6, 0.5))


def light_plane_yaml(tmp_path, name="lightplane"):
    """A Lambertian ground quad (y = 0) under a spherical DiffuseLight of radiance Le, black sky.  The light is
    outside the view.  A camera path hits the ground, scatters along n + unit vector — cosine-weighted — and either
    hits the light (emission Le, no scatter: material.rs:66-72, 94) or leaves into the black sky; the plane cannot
    be hit again.  So E[sample] = rho * Le * (cosine-weighted solid angle of the sphere) / pi
    = rho * Le * (r / d)^2 * cos(theta) for a sphere wholly above the horizon, d = distance from the ground point
    to the light's centre, theta = angle between the plane's normal and that direction."""
    g = LIGHT
    text = (camera_text(g["frm"], g["to"], g["fov"])
            + "scene:\n  background: [0, 0, 0]\n  surfaces:\n"
            + "    - type: quad\n      position: [-10, 0, -10]\n      sides: [[0, 0, 20], [20, 0, 0]]\n"
            + f"      material: {{type: lambertian, albedo: [{g['rho'][0]}, {g['rho'][1]}, {g['rho'][2]}]}}\n"
            + f"    - type: sphere\n      radius: {g['r']}\n      position: [{g['c'][0]}, {g['c'][1]}, {g['c'][2]}]\n"
            + f"      material: {{type: diffuseLight, albedo: [{g['le']}, {g['le']}, {g['le']}]}}\n")
    p = tmp_path / f"{name}.yaml"
    p.write_text(text)
    return str(p)


def light_plane_expected(w, h, sub=4):
    """(mean[h, w, 3], p[h, w]): the expected pixel value (average of the closed form over sub x sub points of the
    pixel's jitter window) and the probability that a sample hits the light (for the binomial error bar)."""
    g = LIGHT
    o, d = camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, sub=sub)
    t = -o[1] / d[..., 1]
    pt = o + t[..., None] * d
    to_c = np.asarray(g["c"]) - pt
    dist = np.linalg.norm(to_c, axis=-1)
    prob = (g["r"] / dist) ** 2 * (to_c[..., 1] / dist)
    prob = prob.mean(axis=(2, 3))
    return prob[..., None] * g["le"] * np.asarray(g["rho"]), prob


# ---------------------------------------------------------------- fuzzy metal plane: Q1 (re-sampling) and Q2
METAL = dict(frm=(0, 0.3, 3), to=(0, 0, 0), fov=20, focus=1.0, fuzz=0.5, a=(0.9, 0.8, 0.7))


def metal_plane_yaml(tmp_path, name="metalplane"):
    """A fuzzy metal quad under the uniform sky, seen at grazing incidence with |d| ~ focus_distance = 1.
    metallic_scatter (material.rs:125-138): dir = reflect(d, n) + fuzz * s, s uniform in the unit ball, d NOT
    normalised (Q2); absorbed iff dir . n <= 1e-8, i.e. iff s.n <= -c / fuzz with c = -d.n.  s.n has density
    3/4 (1 - x^2) on [-1, 1], so P(absorbed) = 1/2 - 3/4 q + 1/4 q^3 with q = c / fuzz (q < 1).
    An absorbed attempt costs one depth unit and the same hit is sampled again (Q1, scene/mod.rs:107-119); an
    accepted reflection leaves the plane for the sky.  With max_depth = 2 only a first-attempt success returns
    a * L (a second-attempt success reaches depth 2 and ends black): E = a L (1 - P).  With max_depth = 40 every
    path is accepted within the depth (P^39 ~ 0): a * L exactly."""
    g = METAL
    text = (camera_text(g["frm"], g["to"], g["fov"], g["focus"])
            + f"scene:\n  background: [{L_SKY[0]}, {L_SKY[1]}, {L_SKY[2]}]\n  surfaces:\n"
            + "    - type: quad\n      position: [-50, 0, -50]\n      sides: [[0, 0, 100], [100, 0, 0]]\n"
            + f"      material: {{type: metal, albedo: [{g['a'][0]}, {g['a'][1]}, {g['a'][2]}], fuzz: {g['fuzz']}}}\n")
    p = tmp_path / f"{name}.yaml"
    p.write_text(text)
    return str(p)


def metal_plane_absorbed(w, h, sub=4):
    """P(first attempt absorbed) per pixel (mean over the jitter window), and the mask of pixels that see the plane."""
    g = METAL
    o, d = camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, focus=g["focus"], sub=sub)
    c = -d[..., 1]  # n = (0, 1, 0) facing the camera
    q = np.clip(c / g["fuzz"], 0.0, 1.0)
    p_abs = 0.5 - 0.75 * q + 0.25 * q ** 3
    t = np.where(d[..., 1] < 0, -o[1] / np.minimum(d[..., 1], -1e-12), np.inf)
    x, z = o[0] + t * d[..., 0], o[2] + t * d[..., 2]
    sees = ((d[..., 1] < 0) & (np.abs(x) < 30) & (np.abs(z) < 30)).all(axis=(2, 3))
    return p_abs.mean(axis=(2, 3)), sees


# ---------------------------------------------------------------- sphere (u, v): surfaces.rs:108-118
# The six known answers the reference's source holds (surfaces.rs:112-114), normal -> (u, v)
SPHERE_UV_KAT = [((1, 0, 0), (0.50, 0.50)), ((-1, 0, 0), (0.00, 0.50)), ((0, 1, 0), (0.50, 1.00)),
                 ((0, -1, 0), (0.50, 0.00)), ((0, 0, 1), (0.25, 0.50)), ((0, 0, -1), (0.75, 0.50))]
UV_ALBEDO, UV_ODD, UV_SCALE = (0.9, 0.9, 0.9), (0.1, 0.2, 0.8), 5.0


def uv_view(direction):
    """Camera on the axis `direction` (one of the six KAT normals) at distance 5, looking at the unit sphere."""
    frm = tuple(5.0 * c for c in direction)
    if direction[1] != 0:  # looking along y: vup = (0, 1, 0) is parallel to the view; tilt the camera a little
        frm = (0.05, frm[1], 0.0)
    return frm, (0.0, 0.0, 0.0), 30.0


def uv_checker_expected(direction, w, h, scale=UV_SCALE):
    """Expected colour per pixel of a Lambertian uv-checker sphere under the uniform sky (colour * L), from the
    analytic hit point of each pixel's rays and the (u, v) the reference's comment documents; `safe` marks pixels
    whose whole jitter window hits the sphere inside one checker cell, away from the u seam and the poles."""
    frm, to, fov = uv_view(direction)
    o, d = camera_rays(frm, to, fov, w / h, w, h, sub=3, edges=True)  # the window's centre, corners and edge midpoints
    a = (d * d).sum(-1)
    hb = (d * o).sum(-1)
    disc = hb * hb - a * (o @ o - 1.0)
    hit = disc > 1e-3 * a
    t = (-hb - np.sqrt(np.where(hit, disc, 0.0))) / a
    n = o + t[..., None] * d
    theta = np.arccos(np.clip(-n[..., 1], -1, 1))
    phi = np.arctan2(-n[..., 2], n[..., 0]) + math.pi
    u, v = phi / (2 * math.pi), theta / math.pi
    su, sv = scale * u, scale * v
    cell = np.floor(su).astype(int) + np.floor(sv).astype(int)
    margin = 0.08
    fu, fv = su - np.floor(su), sv - np.floor(sv)
    interior = hit & (fu > margin) & (fu < 1 - margin) & (fv > margin) & (fv < 1 - margin) & (v > 0.05) & (v < 0.95)
    safe = interior.all(axis=(2, 3)) & (cell.max(axis=(2, 3)) == cell.min(axis=(2, 3)))
    even = cell[:, :, 1, 1] % 2 == 0
    colour = np.where(even[..., None], np.asarray(UV_ALBEDO), np.asarray(UV_ODD)) * np.asarray(L_SKY)
    return colour, safe
