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

# Derived figures for the render kernel (one launch per pass; counters summed over all XCDs / SEs / CUs).
# SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count in units of 4 clocks; a wave is resident on one SIMD, WAVES_PER_SIMD
# of them at this kernel's occupancy (argv[2], default 5: the triangles-only build), so
# SIMD-time = SQ_WAVE_CYCLES / WAVES_PER_SIMD.
WAVES_PER_SIMD = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
for k, c in tot.items():
    if "render_persistent_kernel" not in k or c.get("SQ_INSTS_VALU", 0) < 1e9:
        continue
    simd_time = c["SQ_WAVE_CYCLES"] / WAVES_PER_SIMD
    print(f"== derived: {k}")
    print(f"   VALU busy (SQ_ACTIVE_INST_VALU / SIMD-time)         {c['SQ_ACTIVE_INST_VALU'] / simd_time:.3f}")
    print(f"   lane utilisation (THREAD_CYCLES_VALU / 64 / INSTS)  {c['SQ_THREAD_CYCLES_VALU'] / 64.0 / c['SQ_INSTS_VALU']:.3f}")
    print(f"   L2 hit rate (TCC_HIT / TCC_REQ)                     {c['TCC_HIT_sum'] / c['TCC_REQ_sum']:.3f}")
    print(f"   L1 hit rate (1 - TCP_TCC_READ_REQ / TCP accesses)   {1.0 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")
    print(f"   memory-side bytes (FETCH_SIZE + WRITE_SIZE, KB->B)  {(c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024:.4g}")
    print(f"   wave instructions: VALU {c['SQ_INSTS_VALU']:.4g} SALU {c['SQ_INSTS_SALU']:.4g} VMEM_RD {c['SQ_INSTS_VMEM_RD']:.4g} "
          f"LDS {c['SQ_INSTS_LDS']:.4g} SMEM {c['SQ_INSTS_SMEM']:.4g}")
