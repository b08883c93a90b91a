"""Print the kernel timeline (queue, start, duration) of a rocprofv3 --kernel-trace CSV."""
import csv, glob, re, sys
fn = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(fn)) if "tsimk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo, hi = int(sys.argv[2]), int(sys.argv[3])
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    n = re.search(r"tsimk::(k_\w+)", r["Kernel_Name"]).group(1)
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{n:16s} q={r['Queue_Id']:>2s} start={(st - t0) / 1000:8.1f} dur={(en - st) / 1000:6.1f}")
