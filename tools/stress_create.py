"""Start-up stress: load + create + one tiny render of a scene in a fresh process (run many times by the caller)."""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
name = sys.argv[1] if len(sys.argv) > 1 else "lucy"
print("load", flush=True)
d = ra.load_scene(os.path.join(ROOT, "scenes", f"{name}.yaml"), 16 / 9)
print("create", flush=True)
s = ra.Scene(d, 0)
print("hits", flush=True)
s.primary_hits(320, 180, precision=64)
print("render", flush=True)
s.render(160, 90, 2)
s.close()
print("ok", flush=True)
