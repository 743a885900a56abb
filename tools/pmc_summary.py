"""Summarise rocprofv3 --pmc CSV output directories (one per pass) into per-kernel counter totals."""
import csv, glob, os, sys, collections
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"][:60]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add((f, row["Dispatch_Id"]))
for k, c in tot.items():
    print(f"== {k}")
    for name in sorted(c):
        print(f"   {name:36s} {c[name]:.6g}")
