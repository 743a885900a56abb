"""rocprofv3 --pmc pass directories (tools/prof_pmc.sh) -> the JSON bench.py reads for `roofline.traffic`.
usage: python tools/pmc_to_json.py PMC_DIR COMMIT [WORKLOAD] > profiles/rN_final_pmc.json
The file records the workload and a hash of the kernel sources it was taken on; bench.py ignores it when either
no longer matches (bench.pmc_traffic)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

root, commit = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "scenes/lucy.yaml 1920x1080 1024 spp, max_depth 40, persistent kernel, direct sampler"
tot = collections.defaultdict(float)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "render_persistent_kernel" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
    sys.exit(f"no FETCH_SIZE / WRITE_SIZE rows for the render kernel under {root}")
print(json.dumps({
    "workload": workload,  # bench.py's config.workload string for the run the passes repeat
    "kernel_sources_sha": bench.kernel_sources_sha(),
    "commit": commit,
    "launches": 1,
    "source": "tools/prof_pmc.sh: one rocprofv3 --pmc run per pass over one render call of the bench workload",
    "FETCH_SIZE_KB": tot["FETCH_SIZE"],
    "WRITE_SIZE_KB": tot["WRITE_SIZE"],
    "note": "L2 memory-side (fabric) bytes; Infinity-Cache hits are counted (MI355X_MICROARCH.md, HBM section). "
            "Scattered 16-B loads: the guide's x2 FETCH_SIZE correction applies to wide coalesced streams only, so the "
            "raw value is kept.",
    "counters": dict(sorted(tot.items())),
}, indent=1))
