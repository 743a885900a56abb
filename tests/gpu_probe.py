"""First-contact GPU script (not collected by pytest): primary-hit parity + small renders + timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from oracle import ref

def main():
    for name, w, h in (("quads", 200, 150), ("teapot", 320, 180), ("lucy", 320, 180), ("cornell", 200, 200), ("light", 160, 90)):
        path = f"scenes/{name}.yaml"
        asp = w / h
        d = ra.load_scene(path, asp)
        s = ra.Scene(d, 0)
        o = ref.RefScene.from_yaml(path, asp)
        rid, rt = o.primary_hits(w, h, threads=16)
        for prec in (64, 32):
            ids, t = s.primary_hits(w, h, prec)
            mism = int((ids != rid).sum())
            both = (ids == rid) & (rid >= 0)
            rel = float(np.max(np.abs(t[both] - rt[both]) / rt[both])) if both.any() else 0.0
            print(f"{name} f{prec}: id mismatches {mism}/{w*h}  max rel t err {rel:.3e}", flush=True)
        sn = 4
        want, cnt = o.render(w, h, sn, threads=16, sampler=1)
        want /= sn * sn
        for k, kn in ((0, "persistent"), (1, "wavefront")):
            img, st = s.render(w, h, sn, kernel=k, collect_stats=True)
            got = img.astype(np.float64) / (sn * sn)
            rmse = float(np.sqrt(np.mean((got - want) ** 2)))
            print(f"  {kn}: lum {got.mean():.5f} vs {want.mean():.5f} rmse {rmse:.5f} seg/sample {st.segments/st.samples:.2f} "
                  f"(oracle {cnt['segments']/(w*h*sn*sn):.2f}) nodes/seg {st.node_visits/max(st.segments,1):.1f} "
                  f"tri/seg {st.tri_tests/max(st.segments,1):.1f} ms {st.kernel_ms:.2f}", flush=True)
    # timing at 1080p, 16 spp
    for name in ("teapot", "lucy"):
        d = ra.load_scene(f"scenes/{name}.yaml", 1920 / 1080)
        s = ra.Scene(d, 0)
        for k, kn in ((0, "persistent"), (1, "wavefront")):
            for sn in (4, 8):
                img, st = s.render(1920, 1080, sn, kernel=k)
                print(f"{name} 1080p sn={sn} {kn}: {st.kernel_ms:.1f} ms  {st.samples/st.kernel_ms/1e3:.1f} Msamples/s", flush=True)

main()
