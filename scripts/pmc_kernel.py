"""usage: python scripts/pmc_kernel.py <kernel substring> DIR [DIR ...]  (rocprofv3 -d directories, one per counter pass)

Per-invocation median of every counter over the invocations of the kernel whose name contains the substring, largest
grid only (the timed groups; bench.py's context legs launch smaller ones), plus the kernel-trace durations."""
import collections
import csv
import glob
import statistics
import sys


def main():
    # PMC_TOP=n: per counter the mean of the n LARGEST invocations instead of the median (fused kernels launch one chip-full of blocks
    # whatever the number of batches in the group: the grid does not tell the timed 8-batch groups from the 4-batch warm-up ones)
    import os
    top = int(os.environ.get("PMC_TOP", "0"))
    kern, dirs = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(list)
    durs = []
    grid_seen = 0
    for d in dirs:
        for fn in glob.glob(f"{d}/*/*_counter_collection.csv"):
            rows = [r for r in csv.DictReader(open(fn)) if kern in r["Kernel_Name"]]
            if not rows:
                continue
            big = max(int(r.get("Grid_Size", 0) or 0) for r in rows)
            grid_seen = max(grid_seen, big)
            for r in rows:
                if int(r.get("Grid_Size", 0) or 0) == big:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for fn in glob.glob(f"{d}/*/*_kernel_trace.csv"):
            rows = [r for r in csv.DictReader(open(fn)) if kern in r["Kernel_Name"]]
            if not rows:
                continue
            big = max(int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) for r in rows)
            for r in rows:
                if int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) == big:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"kernel *{kern}*, grid {grid_seen} work-items, per invocation (median over {max((len(v) for v in vals.values()), default=0)} invocations; counters are summed over the chip's XCDs as rocprofv3 reports them)")
    if durs:
        print(f"  duration_us (kernel trace, under counter collection)   median {statistics.median(durs):.1f}  min {min(durs):.1f}  n {len(durs)}")
    pick = (lambda x: sum(sorted(x)[-top:]) / len(sorted(x)[-top:])) if top else statistics.median
    if top:
        print(f"  (mean of the {top} largest invocations per counter)")
        if durs:
            print(f"  duration_us of the {top} longest invocations: {pick(durs):.1f}")
    for c in sorted(vals):
        print(f"  {c:28s} {pick(vals[c]):.6g}")
    v = {c: pick(x) for c, x in vals.items()}
    if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
        print(f"  VALU instructions per wave      {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f}")
    if "SQ_WAVE_CYCLES" in v and "SQ_INSTS_VALU" in v:
        tot = v["SQ_INSTS_VALU"] + v.get("SQ_INSTS_SALU", 0) + v.get("SQ_INSTS_LDS", 0) + v.get("SQ_INSTS_SMEM", 0)
        print(f"  wave-cycles per instruction (VALU+SALU+LDS+SMEM)   {v['SQ_WAVE_CYCLES'] / tot:.2f}")
    if "FETCH_SIZE" in v:
        print(f"  HBM-side traffic: 2 x FETCH_SIZE (gfx950 correction) = {2 * v['FETCH_SIZE'] / 1024:.1f} MB read" + (f", WRITE_SIZE {v['WRITE_SIZE'] / 1024:.1f} MB written" if "WRITE_SIZE" in v else ""))


main()
