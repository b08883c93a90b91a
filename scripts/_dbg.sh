cd /tmp && export TMPDIR=/tmp
for st in 0 8; do
TSIM_BENCH_START_SLOT=$st TSIM_BENCH_NO_PROFILE=1 timeout 200 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tlh$st -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 2 --no-extra-legs --no-cpu-baseline > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/tlh$st/*/ | head
f=$(find $GRAFT_REPO_ROOT/gpurun_out/tlh$st -name "*hip_api_trace.csv" | head -1)
echo "start $st: $f"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
calls=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in rows]
calls.sort()
# find the last burst of 20 hipLaunchKernel of LW: take the final 400 calls and print compactly those in the last timed rep: locate by gaps
names=[c[2] for c in calls]
# print the last 140 API calls before the final hipDeviceSynchronize-ish
idx=[i for i,c in enumerate(calls) if "LaunchKernel" in c[2] or "ModuleLaunch" in c[2]]
# group launches into bursts by >200us gaps
b=[];cur=[idx[0]]
for a,c in zip(idx,idx[1:]):
    if calls[c][0]-calls[a][0] > 200_000: b.append(cur); cur=[]
    cur.append(c)
b.append(cur)
reps=[x for x in b if 24<=len(x)<=26]
r=reps[-1]
t0=calls[r[0]][0]
for i in range(r[0]-3, min(len(calls), r[0]+75)):
    s,e,n=calls[i]
    print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:6.1f} {n}")
PY
done
