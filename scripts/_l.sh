cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_gpu_hard_wave.py tests/test_gpu_steps.py tests/test_gpu_sampler.py tests/test_zz_gpu_bench_dist.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $O/lag.txt
timeout 600 python scripts/fuzz_steps.py 8 8 300 2>&1 | tail -1 >> $O/lag.txt
{ for k in 10 20 50 200 1000; do scripts/bq.sh --steps $k; done; echo "== lag off"; for k in 20 200; do TSIM_AMD_HARD_LAG=0 scripts/bq.sh --steps $k; done
  echo "== shapes"; scripts/bq.sh --config C4 --shots 100000 --steps 100; scripts/bq.sh --config C4 --steps 100; scripts/bq.sh --config C3 --steps 100; scripts/bq.sh --config C3 --steps 20; scripts/bq.sh --shots 100000 --steps 200; scripts/bq.sh --steps 20 --approx; } >> $O/lag.txt 2>&1
