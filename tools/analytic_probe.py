"""Prints the deviations of the HIP path from the closed forms of tests/analytic.py (run on the GPU box)."""
import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import analytic as A
import raytracer_amd as ra
tmp = pathlib.Path(tempfile.mkdtemp())
SKY = np.array(A.L_SKY)
w = h = 64; sn = 64
for material, albedo in (("{type: lambertian, albedo: [0.8, 0.5, 0.2]}", (0.8, 0.5, 0.2)), ("{type: dielectric, index: 1.5}", (1., 1., 1.))):
    scene = ra.Scene(ra.load_scene(A.furnace_sphere_yaml(tmp, material), 1.0), 0)
    inside, outside = A.furnace_sphere_masks(w, h)
    want = np.array(albedo) * SKY
    for kernel in (0, 1, 2):
        for sampler in (0, 1):
            img, _ = scene.render(w, h, sn, kernel=kernel, sampler=sampler)
            img = img.astype(np.float64) / (sn * sn)
            print(material[:18], kernel, sampler, "outside", np.abs(img[outside] / SKY - 1).max(), "inside", np.abs(img[inside] / want - 1).max())
scene = ra.Scene(ra.load_scene(A.flat_mesh_yaml(tmp), 1.0), 0)
inside, outside = A.flat_mesh_masks(w, h)
want = np.array([0.8, 0.5, 0.2]) * SKY
for kernel in (0, 1, 2):
    for sn_ in (8, 64):
        img, _ = scene.render(w, h, sn_, kernel=kernel, sampler=1)
        img = img.astype(np.float64) / (sn_ * sn_)
        dev = img[inside] / want - 1
        bad = np.abs(dev).max(axis=1) > 3e-5
        print("flat mesh", kernel, sn_, "max dev", np.abs(dev).max(), "pixels off", bad.sum(), "of", inside.sum(), "worst", dev[np.abs(dev).max(axis=1).argmax()] * sn_ * sn_)
