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


# ---------------------------------------------------------------- thin lens: camera.rs:43-87
# get_ray (camera.rs:76-87): the ray starts at origin - offset, offset = lens_radius * (u_b * rd.x + v_b * rd.y) with
# rd uniform in the unit disk and lens_radius = aperture / 2 (camera.rs:65), and points at
# P(u, v) = lower_left + u * horizontal + v * vertical, which lies on the plane at `focus_distance` from the origin
# along the view axis (camera.rs:62-64).  So:  point(s) = origin + s * (P - origin) - (1 - s) * offset.
#  * s = 1: every lens sample of a pixel sample passes through the same point of the focus plane;
#  * at depth z along the axis (s = z / focus) the lens samples fill a disk of radius lens_radius * |1 - z / focus|
#    around the pinhole's hit point — the circle of confusion.
# All surfaces here are DiffuseLights (emit on both faces, never scatter: material.rs:66-72, 94) over a black sky, so a
# sample IS the colour of the quad it hits.
LENS = dict(frm=(0.0, 0.0, 6.0), to=(0.0, 0.0, 0.0), fov=24.0, focus=4.0)
BOARD_CELL, BOARD_N = 0.25, 8  # cells of the focus-plane board, which spans [-1, 1]^2 at z = frm.z - focus


def lens_camera_text(aperture):
    g = LENS
    return camera_text(g["frm"], g["to"], g["fov"], g["focus"]) + f"  aperture: {aperture}\n"


def board_colour(i, j):
    """Emission of cell (i, j): no two neighbours alike, every channel in (0, 1)."""
    return (0.15 + 0.1 * i, 0.15 + 0.1 * j, 0.1 + 0.8 * ((i + j) % 2))


def _light_quad(q, u, v, colour):
    return (f"    - type: quad\n      position: [{q[0]}, {q[1]}, {q[2]}]\n"
            f"      sides: [[{u[0]}, {u[1]}, {u[2]}], [{v[0]}, {v[1]}, {v[2]}]]\n"
            f"      material: {{type: diffuseLight, albedo: [{colour[0]}, {colour[1]}, {colour[2]}]}}\n")


def lens_board_yaml(tmp_path, aperture, name="lensboard"):
    """An 8 x 8 board of emissive cells ON the focus plane: the image does not depend on the aperture."""
    z = LENS["frm"][2] - LENS["focus"]
    text = lens_camera_text(aperture) + "scene:\n  background: [0, 0, 0]\n  surfaces:\n"
    for i in range(BOARD_N):
        for j in range(BOARD_N):
            text += _light_quad((-1 + i * BOARD_CELL, -1 + j * BOARD_CELL, z), (BOARD_CELL, 0, 0), (0, BOARD_CELL, 0),
                                board_colour(i, j))
    p = tmp_path / f"{name}_{aperture}.yaml"
    p.write_text(text)
    return str(p)


def lens_board_expected(w, h):
    """(colour[h, w, 3], safe[h, w]): the cell each pixel's focus-plane point lies in; `safe` = the pixel's whole
    jitter window lies inside one cell, 0.2 pixels or more from its border."""
    g = LENS
    o, d = camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, focus=g["focus"], sub=3, edges=True)
    p = o + d  # s = 1: the focus plane
    ci, cj = np.floor((p[..., 0] + 1) / BOARD_CELL), np.floor((p[..., 1] + 1) / BOARD_CELL)
    px = 2.0 * g["focus"] * math.tan(math.radians(g["fov"]) / 2.0) / h
    fx, fy = (p[..., 0] + 1) / BOARD_CELL - ci, (p[..., 1] + 1) / BOARD_CELL - cj
    m = 0.2 * px / BOARD_CELL
    inside = (fx > m) & (fx < 1 - m) & (fy > m) & (fy < 1 - m) & (ci >= 0) & (ci < BOARD_N) & (cj >= 0) & (cj < BOARD_N)
    safe = inside.all(axis=(2, 3)) & (ci.max(axis=(2, 3)) == ci.min(axis=(2, 3))) & (cj.max(axis=(2, 3)) == cj.min(axis=(2, 3)))
    i, j = ci[:, :, 1, 1].astype(int), cj[:, :, 1, 1].astype(int)
    colour = np.stack([0.15 + 0.1 * i, 0.15 + 0.1 * j, 0.1 + 0.8 * ((i + j) % 2)], axis=-1)
    return colour, safe


QUADRANT_Z, QUADRANT_X0, QUADRANT_Y0 = -2.0, 0.1, -0.05  # depth 8 = 2 x focus: circle of confusion = lens radius
QUADRANT_COLOURS = {(0, 0): (0.9, 0.1, 0.1), (1, 0): (0.1, 0.9, 0.1), (0, 1): (0.1, 0.1, 0.9), (1, 1): (0.8, 0.8, 0.2)}


def lens_quadrants_yaml(tmp_path, aperture, name="lensquadrants"):
    """Four large emissive quads that meet in one point, OFF the focus plane: each colour edge blurs by the circle of
    confusion, a disk of radius aperture / 2 * |1 - depth / focus| around the pinhole's hit point."""
    text = lens_camera_text(aperture) + "scene:\n  background: [0, 0, 0]\n  surfaces:\n"
    for (sx, sy), colour in QUADRANT_COLOURS.items():
        text += _light_quad((QUADRANT_X0, QUADRANT_Y0, QUADRANT_Z), (10.0 if sx else -10.0, 0, 0), (0, 10.0 if sy else -10.0, 0), colour)
    p = tmp_path / f"{name}_{aperture}.yaml"
    p.write_text(text)
    return str(p)


def lens_quadrants_expected(w, h, aperture, lens_grid=64):
    """(mean[h, w, 3], var[h, w, 3]): expectation and variance of ONE sample of each pixel — the colour under
    pinhole_hit - (1 - depth / focus) * offset.  Exact over the jitter window (the window maps to a rectangle on
    the quads' plane and the colour edges are axis-parallel: overlaps of intervals), a uniform grid of points over
    the lens disk."""
    g = LENS
    o, d = camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, focus=g["focus"], sub=2, edges=True)
    s = (g["frm"][2] - QUADRANT_Z) / g["focus"]
    pin = o + s * d  # (h, w, 2, 2, 3): the corners of each pixel's jitter window on the plane
    x_lo, x_hi = pin[..., 0].min(axis=(2, 3)), pin[..., 0].max(axis=(2, 3))
    y_lo, y_hi = pin[..., 1].min(axis=(2, 3)), pin[..., 1].max(axis=(2, 3))
    a = (np.arange(lens_grid) + 0.5) / lens_grid * 2 - 1
    lx, ly = np.meshgrid(a, a)
    keep = lx * lx + ly * ly < 1
    # the camera looks down -z with vup = +y: u_b = +x, v_b = +y (camera.rs:57-59)
    off = 0.5 * aperture * (1 - s) * np.stack([lx[keep], ly[keep]], axis=-1)  # (n_lens, 2)
    # fraction of the (shifted) window on the high side of each edge, per lens point
    fx = np.clip((x_hi[..., None] - off[:, 0] - QUADRANT_X0) / (x_hi - x_lo)[..., None], 0.0, 1.0)
    fy = np.clip((y_hi[..., None] - off[:, 1] - QUADRANT_Y0) / (y_hi - y_lo)[..., None], 0.0, 1.0)
    mean = np.zeros((h, w, 3))
    second = np.zeros((h, w, 3))
    for (qx, qy), colour in QUADRANT_COLOURS.items():
        frac = ((fx if qx else 1 - fx) * (fy if qy else 1 - fy)).mean(axis=-1)
        c = np.asarray(colour)
        mean += frac[..., None] * c
        second += frac[..., None] * c * c
    return mean, np.maximum(second - mean * mean, 0.0)


# ---------------------------------------------------------------- dielectric: material.rs:140-181
# A glass PANE is one quad: a single interface.  A camera ray that meets it is reflected with probability
# R = 1 if ratio * sin(theta) > 1 else schlick(cos(theta), ratio), ratio = 1 / index on the front face (d . n < 0,
# trace.rs:59-76) and index on the back face (material.rs:142-152), and refracted otherwise, along
# refract(unit(d), n, ratio) (material.rs:175-181).  The reflected ray leaves into the uniform sky (L, attenuation 1).
# What the refracted ray meets decides the scene:
#  * a black Lambertian floor (albedo 0): the path's throughput is 0 from there on  ->  E[sample] = L * R;
#  * a floor of emissive cells at depth `drop` under the pane  ->  E[sample] = L * R + (1 - R) * E(cell under the
#    Snell-displaced point), the sky's L where the refracted ray misses the cells.
GLASS_FRONT = dict(frm=(0.0, 2.0, 4.0), to=(0.0, 0.0, 0.0), fov=40.0, ior=1.5, up=True)
# seen from its back face the pane refracts only steeper than asin(1 / index) from the normal: look down at it
GLASS_BACK = dict(frm=(0.0, 3.0, 1.5), to=(0.0, 0.0, 0.0), fov=60.0, ior=1.5, up=False)
GLASS_DROP, GLASS_CELL = 1.0, 0.5
GLASS_CELLS_X, GLASS_CELLS_Z = (-3.0, 3.0), (-8.0, 3.0)
GLASS_PALETTE = ((0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9))


def glass_cell_colour(i, k):
    return GLASS_PALETTE[(i + 2 * k) % 3]


def glass_pane_yaml(tmp_path, g, floor, name="glass"):
    """`floor`: "black" (one Lambertian quad of albedo 0) or "cells" (emissive cells).  g["up"]: the pane's normal
    u x v (surfaces.rs:313-327) points up, at the camera (front face), or down (the camera sees its back)."""
    sides = "[[0, 0, 100], [100, 0, 0]]" if g["up"] else "[[100, 0, 0], [0, 0, 100]]"
    text = (camera_text(g["frm"], g["to"], g["fov"])
            + f"scene:\n  background: [{L_SKY[0]}, {L_SKY[1]}, {L_SKY[2]}]\n  surfaces:\n"
            + f"    - type: quad\n      position: [-50, 0, -50]\n      sides: {sides}\n"
            + f"      material: {{type: dielectric, index: {g['ior']}}}\n")
    if floor == "black":
        text += ("    - type: quad\n      position: [-100, -1, -100]\n      sides: [[0, 0, 200], [200, 0, 0]]\n"
                 "      material: {type: lambertian, albedo: [0, 0, 0]}\n")
    else:
        nx = int(round((GLASS_CELLS_X[1] - GLASS_CELLS_X[0]) / GLASS_CELL))
        nz = int(round((GLASS_CELLS_Z[1] - GLASS_CELLS_Z[0]) / GLASS_CELL))
        for i in range(nx):
            for k in range(nz):
                text += _light_quad((GLASS_CELLS_X[0] + i * GLASS_CELL, -GLASS_DROP, GLASS_CELLS_Z[0] + k * GLASS_CELL),
                                    (0, 0, GLASS_CELL), (GLASS_CELL, 0, 0), glass_cell_colour(i, k))
    p = tmp_path / f"{name}_{'front' if g['up'] else 'back'}_{floor}.yaml"
    p.write_text(text)
    return str(p)


def glass_pane_expected(g, w, h, floor, sub=12):
    """(mean[h, w, 3], var[h, w, 3], r[h, w], quad_err[h, w, 3]): expectation and variance of one sample per pixel, the
    mean reflection probability (1 where the pixel looks at the sky: there the sample is L with certainty), and a bound on
    the error of the sub x sub midpoint quadrature over the jitter window: steps inside a pixel — cell edges under the
    pane, the critical angle where R jumps from Schlick's value to 1 — cost at most their height per sub-interval (found
    on a grid that spans the window border to border)."""
    sky = np.asarray(L_SKY)
    n = np.array([0.0, 1.0, 0.0])  # the normal against the ray (Hit::new flips it on the back face)
    ratio = 1.0 / g["ior"] if g["up"] else g["ior"]  # front face iff the geometric normal points up, at the camera
    nx = int(round((GLASS_CELLS_X[1] - GLASS_CELLS_X[0]) / GLASS_CELL))
    nz = int(round((GLASS_CELLS_Z[1] - GLASS_CELLS_Z[0]) / GLASS_CELL))

    def moments(o, d):
        down = d[..., 1] < 0
        ud = d / np.linalg.norm(d, axis=-1, keepdims=True)
        cos_t = np.minimum(-(ud @ n), 1.0)
        sin_t = np.sqrt(np.maximum(1.0 - cos_t * cos_t, 0.0))
        r0 = ((1.0 - ratio) / (1.0 + ratio)) ** 2
        refl = np.where(ratio * sin_t > 1.0, 1.0, r0 + (1.0 - r0) * (1.0 - cos_t) ** 5)
        refl = np.where(down, refl, 1.0)
        if floor == "black":
            below = np.zeros(d.shape)
        else:
            perp = ratio * (ud + cos_t[..., None] * n)
            par = -np.sqrt(np.abs(1.0 - (perp * perp).sum(-1)))
            t_dir = perp + par[..., None] * n
            t_pane = np.where(down, -o[1] / np.where(down, d[..., 1], -1.0), 0.0)
            hit = o + t_pane[..., None] * d
            fl = hit + t_dir * (GLASS_DROP / np.maximum(-t_dir[..., 1], 1e-12))[..., None]
            i = np.floor((fl[..., 0] - GLASS_CELLS_X[0]) / GLASS_CELL).astype(int)
            k = np.floor((fl[..., 2] - GLASS_CELLS_Z[0]) / GLASS_CELL).astype(int)
            on = (i >= 0) & (i < nx) & (k >= 0) & (k < nz)
            pal = np.asarray(GLASS_PALETTE)[(np.clip(i, 0, nx - 1) + 2 * np.clip(k, 0, nz - 1)) % 3]
            below = np.where(on[..., None], pal, sky)
        first = refl[..., None] * sky + (1.0 - refl[..., None]) * below
        second = refl[..., None] * sky * sky + (1.0 - refl[..., None]) * below * below
        return first, second, refl

    first, second, refl = moments(*camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, sub=sub))
    mean = first.mean(axis=(2, 3))
    var = np.maximum(second.mean(axis=(2, 3)) - mean * mean, 0.0)
    across, _, _ = moments(*camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, sub=sub + 1, edges=True))
    spread = across.max(axis=(2, 3)) - across.min(axis=(2, 3))
    return mean, var, refl.mean(axis=(2, 3)), spread / sub


def statistically_equal(img, mean, var, n, quad_err=None, k_pixel=5.5, k_image=4.5, floor=2e-3):
    """img (h, w, 3) = average of n samples per pixel whose one-sample expectation / variance are mean / var;
    quad_err bounds the error of `mean` itself per pixel (treated as uniform in +-quad_err where pixels are averaged).
    Per pixel within k_pixel sigma + quad_err + floor (f32, the jitter quadrature of smooth terms); the image's mean
    and its row and column profiles — each averaging many pixels — within k_image of THEIR sigma.  Returns a message
    for the first violation, or None."""
    if quad_err is None:
        quad_err = np.zeros_like(mean)
    sigma = np.sqrt(var / n)
    bad = np.abs(img - mean) > k_pixel * sigma + quad_err + floor
    if bad.any():
        at = tuple(np.argwhere(bad)[0])
        return f"pixel {at}: {img[at]:.5f} vs {mean[at]:.5f} +- {sigma[at]:.5f} (quadrature {quad_err[at]:.5f}); {bad.sum()} such"
    for axes, what in (((0, 1), "image"), ((0,), "column profile"), ((1,), "row profile")):
        got, want = img.mean(axis=axes), mean.mean(axis=axes)
        count = np.prod([img.shape[a] for a in axes])
        s = np.sqrt(var.sum(axis=axes) / n + (quad_err ** 2).sum(axis=axes) / 3.0) / count
        if (np.abs(got - want) > k_image * s + 0.25 * floor).any():
            return f"{what}: off by {np.abs(got - want).max():.5f}, sigma {s.max():.5f}"
    return None


# ---------------------------------------------------------------- the widened path: media, motion, checker
# Scenes the YAML schema cannot express (load.rs builds stationary spheres and solid colours only): a base scene is
# loaded from YAML and its arrays are then edited, the way the example generator fills them.  `ra` is the product's
# Python binding (host side only: no GPU is needed to build a description); the oracle takes the same arrays through
# RefScene.from_desc with a camera of its own.
WIDE_CAM = dict(frm=(0.0, 0.0, 5.0), to=(0.0, 0.0, 0.0), fov=30.0)


def _wide_base(ra, tmp_path, name, background, surface_yaml, aspect):
    g = WIDE_CAM
    text = (camera_text(g["frm"], g["to"], g["fov"])
            + f"scene:\n  background: [{background[0]}, {background[1]}, {background[2]}]\n  surfaces:\n" + surface_yaml)
    p = tmp_path / f"{name}.yaml"
    p.write_text(text)
    return ra.load_scene(str(p), aspect)


def wide_camera_params(aspect):
    g = WIDE_CAM
    return [g["fov"], aspect, *g["frm"], *g["to"], 0.0, 0.0, 0.0]


# ConstantMedium::hit (surfaces.rs:611-651): the free path is -ln(U) / density, and the ray scatters inside the boundary
# iff that is at most the distance inside, (t_out - t_in) * |d|.  A sphere of BLACK isotropic smoke (albedo 0: a
# scattered path carries nothing from there on) under the uniform sky therefore shows
#     E[sample] = L * exp(-density * chord),  chord = 2 sqrt(R^2 - b^2),  b = the ray's distance from the centre.
SMOKE = dict(radius=1.0, density=0.8)


def smoke_ball_desc(ra, tmp_path, aspect=1.0):
    base = _wide_base(ra, tmp_path, "smoke", L_SKY,
                      f"    - type: sphere\n      radius: {SMOKE['radius']}\n      position: [0, 0, 0]\n"
                      "      material: {type: lambertian, albedo: [0, 0, 0]}\n", aspect)
    mats = base.materials.copy()
    mats["kind"][0] = ra.RT_MAT_ISOTROPIC
    media = np.zeros(1, ra.api.MEDIUM_DTYPE)
    media[0] = (ra.RT_SURF_SPHERE, 0, 1, 0, SMOKE["density"])
    surfaces = np.array([(ra.RT_SURF_MEDIUM, 0)], dtype=ra.api.SURFACE_DTYPE)  # the boundary sphere itself is not a surface
    return ra.SceneDesc.from_arrays(mats, base.spheres, base.triangles, base.quads, surfaces, base.background, base.camera,
                                    media=media)


def smoke_ball_expected(w, h, sub=8):
    """(mean[h, w, 3], var[h, w, 3]) of one sample per pixel."""
    g = WIDE_CAM
    o, d = camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, sub=sub)
    ud = d / np.linalg.norm(d, axis=-1, keepdims=True)
    b2 = (o @ o) - (ud @ o) ** 2
    chord = 2.0 * np.sqrt(np.maximum(SMOKE["radius"] ** 2 - b2, 0.0))
    p = np.exp(-SMOKE["density"] * chord)  # P(the ray makes it through)
    sky = np.asarray(L_SKY)
    mean = p.mean(axis=(2, 3))[..., None] * sky
    var = np.maximum((p.mean(axis=(2, 3))[..., None] * sky * sky) - mean * mean, 0.0)
    return mean, var


# Sphere::moving / Sphere::center (surfaces.rs:43-54): the centre at ray time t is c0 + t (c1 - c0), and the camera
# draws t uniformly in [0, 1) (camera.rs:80).  An EMISSIVE moving sphere over a black sky: a sample is the sphere's
# colour iff the ray meets the sphere where it is at the ray's time, so
#     E[sample] = colour * |{t in [0, 1) : dist(ray, c(t)) < r}|   — the roots of a quadratic in t.
MOVER = dict(c0=(-0.8, -0.1, 0.0), c1=(0.8, 0.3, 0.0), r=0.4, colour=(0.9, 0.6, 0.2))


def moving_light_desc(ra, tmp_path, aspect=1.0):
    m = MOVER
    base = _wide_base(ra, tmp_path, "mover", (0, 0, 0),
                      f"    - type: sphere\n      radius: {m['r']}\n      position: [{m['c0'][0]}, {m['c0'][1]}, {m['c0'][2]}]\n"
                      f"      material: {{type: diffuseLight, albedo: [{m['colour'][0]}, {m['colour'][1]}, {m['colour'][2]}]}}\n", aspect)
    spheres = base.spheres.copy()
    spheres["moving"][0] = 1
    spheres["center1"][0] = m["c1"]
    return ra.SceneDesc.from_arrays(base.materials, spheres, base.triangles, base.quads, base.surfaces, base.background,
                                    base.camera)


def moving_light_expected(w, h, sub=6):
    g, m = WIDE_CAM, MOVER
    o, d = camera_rays(g["frm"], g["to"], g["fov"], w / h, w, h, sub=sub)
    ud = d / np.linalg.norm(d, axis=-1, keepdims=True)
    c0, dc = np.asarray(m["c0"]) - o, np.asarray(m["c1"]) - np.asarray(m["c0"])
    w0 = c0 - (ud @ c0)[..., None] * ud          # (c(t) - o) perpendicular to the ray = w0 + t w1
    w1 = dc - (ud @ dc)[..., None] * ud
    qa, qb, qc = (w1 * w1).sum(-1), (w0 * w1).sum(-1), (w0 * w0).sum(-1) - m["r"] ** 2
    disc = qb * qb - qa * qc
    root = np.sqrt(np.maximum(disc, 0.0))
    t_lo, t_hi = (-qb - root) / qa, (-qb + root) / qa
    frac = np.where(disc > 0, np.clip(t_hi, 0, 1) - np.clip(t_lo, 0, 1), 0.0)
    frac = frac.mean(axis=(2, 3))
    col = np.asarray(m["colour"])
    mean = frac[..., None] * col
    var = np.maximum(frac[..., None] * col * col - mean * mean, 0.0)
    return mean, var


# Checker::color_at (material.rs:237-247): even iff floor(x / s) + floor(y / s) + floor(z / s) is even, s the texture's
# scale.  A Lambertian checker PLANE under the uniform sky is a furnace: every sample is colour(hit point) * L exactly.
CHECKER = dict(scale=0.7, even=(0.9, 0.9, 0.9), odd=(0.2, 0.3, 0.1), frm=(0.4, 2.0, 5.0), y=-0.35)


def checker_plane_desc(ra, tmp_path, aspect=1.0):
    c = CHECKER
    g = dict(WIDE_CAM, frm=c["frm"])
    text = (camera_text(g["frm"], g["to"], g["fov"])
            + f"scene:\n  background: [{L_SKY[0]}, {L_SKY[1]}, {L_SKY[2]}]\n  surfaces:\n"
            + f"    - type: quad\n      position: [-40, {c['y']}, -40]\n      sides: [[0, 0, 80], [80, 0, 0]]\n"
            + f"      material: {{type: lambertian, albedo: [{c['even'][0]}, {c['even'][1]}, {c['even'][2]}]}}\n")
    p = tmp_path / "checker.yaml"
    p.write_text(text)
    base = ra.load_scene(str(p), aspect)
    mats = base.materials.copy()
    mats["texture"][0] = ra.RT_TEX_CHECKER
    mats["albedo2"][0] = c["odd"]
    mats["tex_scale"][0] = c["scale"]
    return ra.SceneDesc.from_arrays(mats, base.spheres, base.triangles, base.quads, base.surfaces, base.background, base.camera)


def checker_camera_params(aspect):
    return [WIDE_CAM["fov"], aspect, *CHECKER["frm"], *WIDE_CAM["to"], 0.0, 0.0, 0.0]


def checker_plane_expected(w, h):
    """(colour[h, w, 3], safe[h, w]): `safe` = the pixel's whole jitter window sees the plane inside one cell."""
    c = CHECKER
    o, d = camera_rays(c["frm"], WIDE_CAM["to"], WIDE_CAM["fov"], w / h, w, h, sub=3, edges=True)
    down = d[..., 1] < -1e-9
    t = np.where(down, (c["y"] - o[1]) / np.where(down, d[..., 1], -1.0), 0.0)
    p = o + t[..., None] * d
    s = p / c["scale"]
    s[..., 1] = c["y"] / c["scale"]  # the plane's own y, exactly
    cell = np.floor(s).astype(np.int64)
    parity = cell.sum(-1) % 2
    f = s - np.floor(s)
    inside = down & (np.abs(p[..., 0]) < 39) & (np.abs(p[..., 2]) < 39)
    for a in (0, 2):
        inside &= (f[..., a] > 0.06) & (f[..., a] < 0.94)
    same = (cell[..., 0].max(axis=(2, 3)) == cell[..., 0].min(axis=(2, 3))) & (cell[..., 2].max(axis=(2, 3)) == cell[..., 2].min(axis=(2, 3)))
    safe = inside.all(axis=(2, 3)) & same
    colour = np.where(parity[:, :, 1, 1, None] == 0, np.asarray(c["even"]), np.asarray(c["odd"])) * np.asarray(L_SKY)
    return colour, safe
