"""Synthetic scene files (reference YAML schema, load.rs:346-466) for the edge cases no shipped scene covers."""
import numpy as np

CAM = "camera:\n  fov: 40\n  from: [0, 1.5, 9]\n  towards: [0, 0.5, 0]\n"


def _mat(rng, kind=None):
    kind = kind or rng.choice(["lambertian", "lambertian", "metal", "dielectric"])
    if kind == "lambertian":
        a = rng.uniform(0.2, 0.9, 3)
        return f"{{type: lambertian, albedo: [{a[0]:.3f}, {a[1]:.3f}, {a[2]:.3f}]}}"
    if kind == "metal":
        a = rng.uniform(0.5, 0.95, 3)
        return f"{{type: metal, albedo: [{a[0]:.3f}, {a[1]:.3f}, {a[2]:.3f}], fuzz: {rng.uniform(0, 0.4):.3f}}}"
    if kind == "light":
        return "{type: diffuseLight, albedo: [4, 4, 4]}"
    return "{type: dielectric, index: 1.5}"


def many_prims(tmp_path, n_spheres=28, n_quads=5, with_mesh=True, seed=7):
    """More than 16 spheres+quads beside a mesh (more than 32 without one: n_spheres=40), so they live in the render tree (f64 tests inside the leaf stage), next to
    a small triangle mesh, a light and a hollow glass sphere (negative radius inside a positive one)."""
    rng = np.random.default_rng(seed)
    s = [CAM, "scene:\n  background: [0.5, 0.6, 0.8]\n  surfaces:\n"]
    s.append("    - type: sphere\n      radius: 100\n      position: [0, -100, 0]\n"
             "      material: {type: lambertian, albedo: [0.5, 0.5, 0.5]}\n")
    for k in range(n_spheres):
        x, z = rng.uniform(-4, 4), rng.uniform(-4, 3)
        r = rng.uniform(0.15, 0.45)
        s.append(f"    - type: sphere\n      radius: {r:.3f}\n      position: [{x:.3f}, {r:.3f}, {z:.3f}]\n"
                 f"      material: {_mat(rng)}\n")
    s.append("    - type: sphere\n      radius: 0.6\n      position: [0, 0.6, 4]\n      material: {type: dielectric, index: 1.5}\n")
    s.append("    - type: sphere\n      radius: -0.5\n      position: [0, 0.6, 4]\n      material: {type: dielectric, index: 1.5}\n")
    for k in range(n_quads):
        x, z = rng.uniform(-4, 3), rng.uniform(-5, -2)
        kind = "light" if k == 0 else None
        s.append(f"    - type: quad\n      position: [{x:.3f}, 0, {z:.3f}]\n      sides: [[1.2, 0, 0.3], [0, 1.5, 0]]\n"
                 f"      material: {_mat(rng, kind)}\n")
    if with_mesh:
        n = 10
        lines = []
        for j in range(n + 1):
            for i in range(n + 1):
                lines.append(f"v {i / n * 2 - 1:.4f} {0.15 * np.sin(3 * i / n) * np.cos(4 * j / n):.4f} {j / n * 2 - 1:.4f}")
        for j in range(n):
            for i in range(n):
                a = j * (n + 1) + i + 1
                lines.append(f"f {a} {a + 1} {a + n + 2}")
                lines.append(f"f {a} {a + n + 2} {a + n + 1}")
        (tmp_path / "grid.obj").write_text("\n".join(lines) + "\n")
        s.append("    - type: mesh\n      path: grid.obj\n      material: {type: metal, albedo: [0.9, 0.8, 0.6], fuzz: 0.05}\n"
                 "      transform:\n        - {type: translate, dir: [0, 1.6, 0]}\n")
    p = tmp_path / "many.yaml"
    p.write_text("".join(s))
    return str(p)


def room(tmp_path, n_spheres=20, seed=5):
    """Six walls around n small spheres and a small mesh: more spheres and quads than stay loose beside a mesh, so the
    small ones go into the render tree — the walls, each a third of the room's box, must not (api.cpp loose_surfaces)."""
    rng = np.random.default_rng(seed)
    s = [CAM, "scene:\n  background: [0, 0, 0]\n  surfaces:\n"]
    walls = [([-5, 0, -5], [10, 0, 0], [0, 0, 15]), ([-5, 6, -5], [10, 0, 0], [0, 0, 15]),   # floor, ceiling
             ([-5, 0, -5], [0, 6, 0], [0, 0, 15]), ([5, 0, -5], [0, 6, 0], [0, 0, 15]),      # left, right
             ([-5, 0, -5], [10, 0, 0], [0, 6, 0]), ([-5, 0, 10], [10, 0, 0], [0, 6, 0])]     # back, front (behind the camera)
    for k, (q, u, v) in enumerate(walls):
        kind = "light" if k == 1 else "lambertian"
        s.append(f"    - type: quad\n      position: {q}\n      sides: [{u}, {v}]\n      material: {_mat(rng, kind)}\n")
    for k in range(n_spheres):
        x, z, r = rng.uniform(-4, 4), rng.uniform(-4, 3), rng.uniform(0.15, 0.45)
        s.append(f"    - type: sphere\n      radius: {r:.3f}\n      position: [{x:.3f}, {r:.3f}, {z:.3f}]\n"
                 f"      material: {_mat(rng)}\n")
    (tmp_path / "tri.obj").write_text("v -1 0 0\nv 1 0 0\nv 0 1.5 0\nv 0 0.5 1\nf 1 2 3\nf 1 2 4\nf 2 3 4\nf 1 3 4\n")
    s.append("    - type: mesh\n      path: tri.obj\n      material: {type: metal, albedo: [0.9, 0.8, 0.6], fuzz: 0.05}\n"
             "      transform:\n        - {type: translate, dir: [0, 0.2, 1]}\n")
    p = tmp_path / "room.yaml"
    p.write_text("".join(s))
    return str(p)


def many_quads(tmp_path, n=40, seed=3):
    """A floor, a light and n upright panels: more than 32 quads and nothing else, so the render tree holds quads only
    (the kernels' TREE = 3 build)."""
    rng = np.random.default_rng(seed)
    s = [CAM, "scene:\n  background: [0.5, 0.6, 0.8]\n  surfaces:\n"]
    s.append("    - type: quad\n      position: [-6, 0, -6]\n      sides: [[12, 0, 0], [0, 0, 12]]\n"
             "      material: {type: lambertian, albedo: [0.5, 0.5, 0.5]}\n")
    s.append("    - type: quad\n      position: [-1, 3, -1]\n      sides: [[2, 0, 0], [0, 0, 2]]\n"
             f"      material: {_mat(rng, 'light')}\n")
    for k in range(n):
        x, z, turn = rng.uniform(-4, 4), rng.uniform(-4, 3), rng.uniform(0, np.pi)
        wdt, hgt = rng.uniform(0.3, 0.9), rng.uniform(0.3, 1.2)
        s.append(f"    - type: quad\n      position: [{x:.3f}, 0, {z:.3f}]\n"
                 f"      sides: [[{wdt * np.cos(turn):.3f}, 0, {wdt * np.sin(turn):.3f}], [0, {hgt:.3f}, 0]]\n"
                 f"      material: {_mat(rng)}\n")
    p = tmp_path / "panels.yaml"
    p.write_text("".join(s))
    return str(p)


def deep_tree(tmp_path, n, ratio=0.7, name="deep"):
    """n triangles whose size and position shrink geometrically: the SAH builder peels a few off per level,
    so the tree is about n/2 levels deep (exercises the larger traversal-stack builds)."""
    # mirrored about the origin so the mesh centre (which the loader subtracts, load.rs:332-344) is the cluster
    # point and f32 keeps resolving the small triangles
    lines, x = [], 4.0
    for k in range(n):
        w = 0.45 * x
        sgn = 1.0 if k % 2 == 0 else -1.0
        lines += [f"v {sgn * x:.9g} 0 0", f"v {sgn * (x + w):.9g} 0 0", f"v {sgn * x:.9g} {w:.9g} 0"]
        if k % 2 == 1:
            x *= ratio
    lines += ["v 5.8 0 0", "v -5.8 0 0", "v 0 1e-30 0"]  # pins the bounding box's centre to x = 0 (unreferenced by faces)
    for k in range(n):
        lines.append(f"f {3 * k + 1} {3 * k + 2} {3 * k + 3}")
    (tmp_path / f"{name}.obj").write_text("\n".join(lines) + "\n")
    text = ("camera:\n  fov: 60\n  from: [2, 1, 6]\n  towards: [2, 0.5, 0]\n"
            "scene:\n  background: [0.7, 0.8, 1.0]\n  surfaces:\n"
            f"    - type: mesh\n      path: {name}.obj\n      material: {{type: lambertian, albedo: [0.8, 0.3, 0.3]}}\n")
    p = tmp_path / f"{name}.yaml"
    p.write_text(text)
    return str(p)
