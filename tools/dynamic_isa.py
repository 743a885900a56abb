"""Dynamic instruction-class table of the persistent kernel, per stage: the static histogram of each stage's code
(regions marked by RT_REGION comments, compiled with -DRT_REGION_MARKERS; inner loops counted apart) times how often
the counting build says that code ran (stage trips, node visits, loose tests: `RT_STAGE_DEBUG=1 tools/perf_probe.py
--stages` output, or --trips given by hand).  An UPPER bound per class: a divergent sub-branch is counted whenever its
stage runs, though a trip skips it when no lane takes it.

    python tools/dynamic_isa.py --kernel six --stages-log gpurun_out/stage_split_lucy.txt --scene lucy

kernel: "five" = <STATS=0, DIRECT=1, EXT=0, MEDIA=0, TREE=1, NOISE=1, SIX=0>, "six" = the same with SIX=1."""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", choices=["five", "six"], default="six")
ap.add_argument("--stages-log", help="stderr+stdout of RT_STAGE_DEBUG=1 tools/perf_probe.py --stages SCENE")
ap.add_argument("--scene", default="lucy")
ap.add_argument("--node-steps", type=int, default=4)
a = ap.parse_args()
want = "render_persistent_kernelILb0ELb1ELb0ELb0ELi1ELb1ELb%dE" % (1 if a.kernel == "six" else 0)


def cls(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane moves"
    if op.startswith(("v_mov", "v_accvgpr", "v_pk_mov", "v_swap")): return "v_mov"
    if re.match(r"v_(fma|fmac|fmamk|fmaak|mul|add|sub|subrev)_f32", op) or op.startswith("v_fma_mix"): return "f32 fma/mul/add"
    if re.match(r"v_pk_(fma|mul|add)_f32", op): return "packed f32 (2 ops)"
    if re.match(r"v_(add|sub|subrev|xor|or|and|not|add3|lshl_add|and_or|or3|xad|bfi)_", op) and "f64" not in op: return "int add/logic"
    if "f64" in op: return "f64"
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", op): return "transcendental"
    if op.startswith("v_"): return "min/max/cmp/select/cvt/shift/mul_lo"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vector memory"
    if op.startswith(("s_load", "s_buffer")): return "scalar memory"
    if op.startswith(("s_waitcnt", "s_nop")): return "waits"
    return "SALU/branch"


with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", "-DRT_REGION_MARKERS",
                    "--cuda-device-only", "-S", "-o", out, f"{ROOT}/raytracer_amd/csrc/hip/render_persistent.hip"], check=True,
                   stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
start = next(i for i, l in enumerate(text) if want in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
region, loop = "prologue", None
hist = collections.defaultdict(collections.Counter)
for l in text[start:end + 1]:
    m = re.search(r"; RT_REGION (\S+)", l)
    if m:
        region = m.group(1)
        continue
    if l.startswith(".LBB") or l.startswith("; %bb"):
        m = re.search(r"Loop: Header=(BB\d+_\d+) Depth=(\d+)", l) or re.search(r"Inner Loop Header: Depth=(\d+)", l)
        depth2 = bool(re.search(r"Depth=2|Depth 2", l)) and "Child Loop" not in l
        loop = "loop" if ("Depth=2" in l) else None
    m2 = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)", l)
    if m2:
        hist[f"{region}{'.loop' if loop else ''}"][cls(m2.group(1))] += 1

# how often each region ran
runs = {}
if a.stages_log:
    log = open(a.stages_log).read()
    blk = log[log.index(f"{a.scene}:"):]
    trips = [int(x) for x in re.findall(r"\[stages\] \d: trips (\d+)", blk)[:3]]
    m = re.search(r"seg/sample ([\d.]+) nodes/seg ([\d.]+) tris/seg ([\d.]+) sph/seg ([\d.]+) quad/seg ([\d.]+)", blk)
    loose_per_seg = float(m.group(4)) + float(m.group(5))
    runs = {"select": sum(trips), "node": trips[0], "node.loop": trips[0] * a.node_steps, "node.deep": 0, "node.deep.loop": 0,
            "leaf": trips[1], "leaf.loop": 0, "shade.bounce": trips[2], "shade.bounce.loop": 0, "shade.regenerate": trips[2],
            "shade.regenerate.loop": trips[2], "shade.loose": trips[2], "shade.loose.loop": trips[2] * loose_per_seg,
            "prologue": 0, "prologue.loop": 0, "select.loop": 0}
classes = sorted({c for h in hist.values() for c in h})
print(f"kernel: {text[start].split(':')[0]}\n")
print("static instructions per region (the code of one trip through it; '.loop' = its inner loop's body)\n")
print("| region | " + " | ".join(classes) + " | all | runs |")
print("|---|" + "---|" * (len(classes) + 2))
for r in sorted(hist):
    h = hist[r]
    print(f"| {r} | " + " | ".join(str(h.get(c, 0)) for c in classes) + f" | {sum(h.values())} | {runs.get(r, '')} |")
if runs:
    tot = collections.Counter()
    per_stage = collections.defaultdict(collections.Counter)
    for r, h in hist.items():
        for c, n in h.items():
            tot[c] += n * runs.get(r, 0)
            per_stage[r.split(".")[0]][c] += n * runs.get(r, 0)
    valu = [c for c in classes if c not in ("LDS", "vector memory", "scalar memory", "waits", "SALU/branch")]
    print(f"\ndynamic wave-instructions, {a.scene} (static x runs; upper bound), in units of 1e9\n")
    print("| stage | " + " | ".join(classes) + " | VALU | all |")
    print("|---|" + "---|" * (len(classes) + 2))
    for st in ("select", "node", "leaf", "shade"):
        h = per_stage[st]
        print(f"| {st} | " + " | ".join(f"{h.get(c, 0) / 1e9:.2f}" for c in classes) + f" | {sum(h[c] for c in valu) / 1e9:.2f} | {sum(h.values()) / 1e9:.2f} |")
    print("| **all** | " + " | ".join(f"{tot.get(c, 0) / 1e9:.2f}" for c in classes) + f" | {sum(tot[c] for c in valu) / 1e9:.2f} | {sum(tot.values()) / 1e9:.2f} |")
