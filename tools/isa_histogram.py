"""Static instruction histogram of the persistent render kernel's gfx950 code, by loop region and issue class
(the classes of profiles/r1_valu_issue_rates.txt).  usage: python tools/isa_histogram.py [mangled-name-substring]
Default: the triangles-only lean build every mesh scene runs (<STATS=0, DIRECT=1, EXT=0, MEDIA=0, TREE=1>)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else "render_persistent_kernelILb0ELb1ELb0ELb0ELi1E"
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include",
                    "--cuda-device-only", "-S", "-o", out, f"{ROOT}/raytracer_amd/csrc/hip/render_persistent.hip"],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(rf"^_ZN2rt\d+{want[:24]}", l) and want in l and l.rstrip().endswith(":") or (want in l and re.match(r"^_Z\S+:", l)))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
meta = [l for l in text[end:end + 80] if re.search(r"NumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs", l)]

def cls(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "sgpr<->vgpr lane moves"
    if op.startswith(("v_mov", "v_accvgpr")): return "v_mov"
    if re.match(r"v_(fma|fmac|mul|add|sub|pk_fma|pk_mul|pk_add)_f32", op) or re.match(r"v_(mad|fma)_(mix|legacy)", op): return "f32 fma/mul/add (fast class)"
    if re.match(r"v_(add|sub|subrev|xor|or|and|not|add3|lshl_add|and_or|or3|xad)_", op) and "f64" not in op: return "int add/logic (fast class)"
    if "f64" in op: return "f64"
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", op): return "transcendental"
    if op.startswith("v_"): return "min/max/cmp/select/cvt/shift/mul_lo (slow class)"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vector memory"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "scalar memory"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "waits/nops"
    return "SALU/branch"

region = "straight-line / outer loop"
hist = collections.defaultdict(collections.Counter)
for l in text[start:end + 1]:
    m = re.search(r"Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
    if l.startswith(".LBB") or l.startswith("; %bb"):
        region = f"inner loop {m.group(1)}" if m and int(m.group(2)) >= 2 else "straight-line / outer loop"
    m2 = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)", l)
    if m2:
        hist[region][cls(m2.group(1))] += 1
print("kernel:", text[start].split(":")[0])
for l in meta: print("  ", l.strip("; ").strip())
classes = sorted({c for h in hist.values() for c in h})
print("\n| region | " + " | ".join(classes) + " | all |")
print("|---|" + "---|" * (len(classes) + 1))
for r, h in sorted(hist.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"| {r} | " + " | ".join(str(h.get(c, 0)) for c in classes) + f" | {sum(h.values())} |")
tot = collections.Counter()
for h in hist.values(): tot.update(h)
print("| **whole kernel** | " + " | ".join(str(tot.get(c, 0)) for c in classes) + f" | {sum(tot.values())} |")
