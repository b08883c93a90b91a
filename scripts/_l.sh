cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_gpu_hard_wave.py tests/test_gpu_steps.py tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $O/hw3.txt
{ scripts/bq.sh --steps 20; scripts/bq.sh --steps 200; scripts/bq.sh --config C4 --shots 100000 --steps 100; } >> $O/hw3.txt 2>&1
cd /tmp && export TMPDIR=/tmp; rm -rf $O/ks_s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_s -- python $GRAFT_REPO_ROOT/scripts/lwm_probe.py C2 > $O/ks_s.txt 2>&1
f=$(find $O/ks_s -name "*kernel_stats.csv" | head -1); grep k_sample_hw $f | cut -c1-150 >> $O/hw3.txt; rm -rf $O/ks_s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_s -- python $GRAFT_REPO_ROOT/scripts/lwm_probe.py C4 > $O/ks_s4.txt 2>&1
f=$(find $O/ks_s -name "*kernel_stats.csv" | head -1); grep k_sample_hw $f | cut -c1-150 >> $O/hw3.txt; rm -rf $O/ks_s
