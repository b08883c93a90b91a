"""Per-lane gaps and throughput of the first-pass kernels in the LAST burst of >= n launches of a kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
lw = [k for k in ks if "k_sample_lw" in k[2]]
# bursts: split where the gap between consecutive first-pass starts exceeds 80 us
bursts, cur = [], [lw[0]]
for a, b in zip(lw, lw[1:]):
    if b[0] - a[0] > 80_000: bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
for bi, b in enumerate(bursts):
    if len(b) < 5: continue
    span = (b[-1][1] - b[0][0]) / 1e3
    lanes = {}
    for k in b: lanes.setdefault(k[3], []).append(k)
    gaps = []
    for q, v in lanes.items():
        gaps += [(y[0] - x[1]) / 1e3 for x, y in zip(v, v[1:])]
    h = [k for k in ks if "4h" in k[2] and b[0][0] <= k[0] <= b[-1][1] + 200_000]
    tail = (max(k[1] for k in h) - b[-1][1]) / 1e3 if h else 0
    print(f"burst {bi}: {len(b)} first passes on lanes {sorted(lanes)}, span {span:.0f} us = {span/len(b):.1f} us/launch, "
          f"mean kernel {sum(k[1]-k[0] for k in b)/len(b)/1e3:.1f} us, mean lane gap {sum(gaps)/max(1,len(gaps)):.1f} us, hard-row tail {tail:.0f} us")
