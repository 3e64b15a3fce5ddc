R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5_run33_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5_run33_tests.log
for rep in 1 2; do
  for w in old new; do
    if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
    r=$(timeout -k 10 200 python tools/head_bench.py --ragged --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_by_mode'])")
    echo "head ragged $w rep$rep: $r" | tee -a gpurun_out/ab_head_sums.log
    r=$(timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_by_mode'])")
    echo "tail ragged $w rep$rep: $r" | tee -a gpurun_out/ab_head_sums.log
    for c in c2 c3 "c2 --ragged"; do
    r=$(timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$c $w rep$rep: $r" | tee -a gpurun_out/ab_head_sums.log
    done
  done
done
