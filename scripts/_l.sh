cd $GRAFT_REPO_ROOT; O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/tests_full.txt 2>&1; grep -E "passed|failed|error" $O/tests_full.txt | tail -3 > $O/tests4.txt
timeout 600 python scripts/fuzz_steps.py 8 10 100 2>&1 | tail -3 >> $O/tests4.txt
timeout 300 python scripts/fuzz_pipeline.py 100 C2 C4 2>&1 | tail -3 >> $O/tests4.txt
{ scripts/bq.sh --steps 20; scripts/bq.sh --steps 200; scripts/bq.sh --config C4 --shots 100000 --steps 100; scripts/bq.sh --config C4 --steps 100; scripts/bq.sh --config C3 --steps 100; scripts/bq.sh --shots 100000 --steps 200; } >> $O/tests4.txt 2>&1
