"""Generates the committed golden vectors from the CPU oracle (oracle/rt_ref.cpp + oracle/loader_ref.py).

The reference is Rust and cannot be built or run in this image (no cargo/rustc), and its repository holds no
golden images or render-path fixtures, so these vectors are the ORACLE's outputs, not the reference's:
they pin the oracle against accidental change and give the GPU tests a fixed target that needs no CPU
re-render.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = ["cornell", "cornell_lucy", "teapot", "lucy", "dual", "distant", "light", "quads"]
PRIMARY = (64, 64)        # per-scene primary-hit maps at 64x64
IMAGE = (48, 32, 4)       # width, height, sn  (16 spp)


def main():
    streams = np.stack([ref.rng_stream(s, p, k, 8) for (s, p, k) in [(1, 0, 0), (1, 1, 0), (1, 0, 1), (2, 7, 9)]])
    np.save(os.path.join(HERE, "rng_stream.npy"), streams)
    for name in SCENES:
        path = os.path.join(ROOT, "scenes", f"{name}.yaml")
        w, h = PRIMARY
        o = ref.RefScene.from_yaml(path, w / h)
        ids, t = o.primary_hits(w, h)
        order, stats = o.leaf_order()
        iw, ih, sn = IMAGE
        o2 = ref.RefScene.from_yaml(path, iw / ih)
        img, cnt = o2.render(iw, ih, sn, max_depth=40, seed=1)                 # reference sampler (rejection)
        img_d, cnt_d = o2.render(iw, ih, sn, max_depth=40, seed=1, sampler=1)  # direct sampler
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), prim_id=ids, t=t, leaf_order_head=order[:4096],
                            leaf_order_sum=np.int64(np.sum(order.astype(np.int64) * np.arange(1, len(order) + 1))),
                            tree=np.array([stats["nodes"], stats["leaves"], stats["depth"], stats["max_leaf"]]),
                            image_sum=img, segments=np.int64(cnt["segments"]), image_sum_direct=img_d,
                            segments_direct=np.int64(cnt_d["segments"]))
        print(name, "hits", int((ids >= 0).sum()), "tree", stats, "mean", float(img.mean()) / (sn * sn))


if __name__ == "__main__":
    main()
