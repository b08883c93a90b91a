"""Kernel timeline of the LAST `n` first-pass launches of a rocprofv3 --kernel-trace csv: start/end relative to the first."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")) for r in rows]
ks.sort()
first = [i for i, k in enumerate(ks) if "k_sample_lw" in k[2]]
i0 = first[-n]
t0 = ks[i0][0]
for s, e, name, q in ks[i0:]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f} us  q{q}  {name}")
