R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py tests/test_tail_gpu.py -m gpu -x -q > gpurun_out/r5_run16_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_run16_tests.log
for c in c2 c3; do
  for rep in 1 2; do
    for w in old new; do
      if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
      r=$(timeout -k 10 200 python bench.py --config $c --ragged --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c ragged $w rep$rep: $r" | tee -a gpurun_out/ab_rowblk16_v2.log
    done
  done
done
for w in old new; do
  if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
  r=$(timeout -k 10 200 python bench.py --config c2 --ragged --global-batch 128 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "c2 ragged B=128 $w: $r" | tee -a gpurun_out/ab_rowblk16_v2.log
done
