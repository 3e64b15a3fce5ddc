R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -x -q -k "row_block or ragged or empty_document or determin" > gpurun_out/r5_run19_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_run19_tests.log
for c in c2 c3; do
  for rep in 1 2; do
    for w in old new; do
      if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
      r=$(timeout -k 10 200 python bench.py --config $c --ragged --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c ragged rb-halved-work $w rep$rep: $r" | tee -a gpurun_out/ab_rb_halfwork.log
    done
  done
done
unset GCGCN_LIB
for c in c1; do
  for w in old new; do
      if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
      r=$(timeout -k 10 200 python bench.py --config $c --ragged --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c ragged rb-halved-work $w: $r" | tee -a gpurun_out/ab_rb_halfwork.log
  done
done
