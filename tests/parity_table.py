"""Full-size primary-hit parity of the f32 render arithmetic against the bit-exact f64 AOV kernel, and image
residuals against the oracle (run on the GPU box; prints markdown tables for DESIGN.md §2)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra

def classify(scene, w, h):
    i64, t64 = scene.primary_hits(w, h, precision=64)
    i32, t32 = scene.primary_hits(w, h, precision=32)
    both = (i64 >= 0) & (i32 >= 0)
    rel = np.zeros_like(t64)
    rel[both] = np.abs(t32[both] - t64[both]) / t64[both]
    same = i64 == i32
    swap = ~same & both & (rel <= 1e-5)
    flip = ~same & ~swap
    return dict(pixels=i64.size, hit=int((i64 >= 0).sum()), swaps=int(swap.sum()), flips=int(flip.sum()),
                hit_miss=int((~same & ~both).sum()), max_rel_same=float(rel[same & both].max(initial=0.0)),
                n_rel_same=int((rel[same & both] > 1e-5).sum()))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "ids"):
        print("| scene | frame | pixels that hit | tie-swaps | flips | of which hit<->miss | flip rate | max rel dt (ids equal) | pixels with rel dt > 1e-5 (ids equal) |")
        print("|---|---|---|---|---|---|---|---|---|")
        for name, w, h in (("lucy", 1920, 1080), ("teapot", 1920, 1080), ("dual", 1920, 1080), ("distant", 1920, 1080),
                           ("cornell", 400, 400), ("cornell_lucy", 3840, 2160), ("light", 1920, 1080), ("quads", 1920, 1080),
                           ("cornell", 64, 64), ("cornell_lucy", 64, 64), ("teapot", 64, 64), ("lucy", 64, 64), ("dual", 64, 64),
                           ("distant", 64, 64), ("light", 64, 64), ("quads", 64, 64)):
            s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h), 0)
            t0 = time.time()
            r = classify(s, w, h)
            print(f"| {name} | {w}x{h} | {r['hit']} | {r['swaps']} | {r['flips']} | {r['hit_miss']} | {r['flips'] / r['pixels']:.2e} | {r['max_rel_same']:.1e} | {r['n_rel_same']} |", flush=True)
            s.close()
    if which in ("all", "img"):
        from oracle import ref
        print("\n| scene | frame | spp | sampler | RMSE r g b | rel. luminance diff |")
        print("|---|---|---|---|---|---|")
        g = lambda n: np.load(os.path.join(ROOT, "tests", "golden", f"{n}.npz"))
        for name in ("cornell", "cornell_lucy", "teapot", "lucy", "dual", "distant", "light", "quads"):
            gold = g(name)
            for sampler, key in ((0, "image_sum"), (1, "image_sum_direct")):
                want = gold[key] / 16
                ih, iw, _ = want.shape
                s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), iw / ih), 0)
                img, _ = s.render(iw, ih, 4, sampler=sampler)
                got = img.astype(np.float64) / 16
                rmse = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
                print(f"| {name} | {iw}x{ih} | 16 | {sampler} | {rmse[0]:.4f} {rmse[1]:.4f} {rmse[2]:.4f} | {abs(got.mean() - want.mean()) / want.mean():.2e} |", flush=True)
                s.close()
        for name, w, h, sn in (("teapot", 96, 54, 32), ("cornell_lucy", 96, 96, 32), ("lucy", 96, 54, 32), ("dual", 96, 54, 32)):
            o = ref.RefScene.from_yaml(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h)
            s = ra.Scene(ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), w / h), 0)
            for sampler in (0, 1):
                t0 = time.time()
                want, _ = o.render(w, h, sn, threads=16, sampler=sampler)
                dt = time.time() - t0
                want /= sn * sn
                for kernel in (0, 1, 2):
                    img, _ = s.render(w, h, sn, sampler=sampler, kernel=kernel)
                    got = img.astype(np.float64) / (sn * sn)
                    rmse = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
                    print(f"| {name} (kernel {kernel}) | {w}x{h} | {sn * sn} | {sampler} | {rmse[0]:.4f} {rmse[1]:.4f} {rmse[2]:.4f} | {abs(got.mean() - want.mean()) / want.mean():.2e} | oracle {dt:.1f} s", flush=True)
            s.close()
