R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_producer_gpu.py tests/test_head_gpu.py tests/test_tail_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  for w in old new; do
    if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
    r=$(timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_by_mode'])")
    echo "tail ragged $w rep$rep: $r" | tee -a gpurun_out/ab_pair_ww.log
    r=$(timeout -k 10 200 python tools/head_bench.py --ragged --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_by_mode'])")
    echo "head ragged $w rep$rep: $r" | tee -a gpurun_out/ab_pair_ww.log
  done
done
