#!/bin/bash
# kernel timeline of the last timed region of `bench.py --steps 20` (run on the GPU box)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tr20
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr20 -- python $R/bench.py --steps ${1:-20} --warmup 3 --repeats 3 --spinup-ms ${2:-30} --no-cpu-baseline --no-extra-legs --no-config-legs > $R/gpurun_out/tr20.json 2>/dev/null
python - <<PY
import csv, glob, re
fn = glob.glob("$R/gpurun_out/tr20/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(fn)) if "tsimk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# regions = bursts separated by > 200 us of nothing
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 200_000: bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
big = [b for b in bursts if sum("lw_fast" in r["Kernel_Name"] or "lw_multi" in r["Kernel_Name"] for r in b) >= 2]
for b in big[-2:]:
    t0 = int(b[0]["Start_Timestamp"])
    print("--- region", len(b), "kernels, span %.1f us" % ((max(int(r["End_Timestamp"]) for r in b) - t0) / 1e3))
    for r in b:
        n = re.search(r"tsimk::(k_\w+)", r["Kernel_Name"]).group(1)
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{n:18s} q={r['Queue_Id']:>2s} grid={r['Grid_Size_X'] if 'Grid_Size_X' in r else '?':>8s} start={(st - t0) / 1000:8.1f} dur={(en - st) / 1000:6.1f}")
PY
tail -1 $R/gpurun_out/tr20.json | cut -c1-160
