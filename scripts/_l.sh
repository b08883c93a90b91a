cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
python scripts/_dbg.py 2>&1 | grep "bad rows per batch" | cut -c1-120 > $O/ps2.txt
python -m pytest tests -q -m gpu -x > $O/tests_full.txt 2>&1; grep -E "passed|failed|error" $O/tests_full.txt | tail -3 >> $O/ps2.txt
timeout 300 python scripts/fuzz_steps.py 6 8 400 2>&1 | tail -1 >> $O/ps2.txt
python scripts/e2e_postselect.py 16000000 0.02 2>&1 | tail -4 >> $O/ps2.txt
