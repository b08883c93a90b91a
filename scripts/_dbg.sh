cd /tmp && export TMPDIR=/tmp
TSIM_AMD_BATCH_LANES=2 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tlc4b -- python $GRAFT_REPO_ROOT/bench.py --config C4 --shots 100000 --steps 100 --warmup 5 --repeats 1 --no-extra-legs --no-cpu-baseline > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/tlc4b -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","")) for r in rows)
h=[k for k in ks if "4h_multi" in k[2]]
t0=h[0][0]
for s,e,n,q in h[-24:]:
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} q{q}")
lw=[k for k in ks if "k_sample_lw" in k[2]]
print("LW queues:", sorted(set(k[3] for k in lw)))
PY
