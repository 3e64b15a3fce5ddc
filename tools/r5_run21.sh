# four register sets of operand requests in the stand-alone GEMM kernels (new) against two (old = same tree, gemm.hip -DGC_GEMM_NS=2)
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r5_run21_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_run21_tests.log
for c in "c2" "c3" "c1" "c5" "c2 --ragged" "c3 --ragged"; do
  for rep in 1 2; do
    for w in old new; do
      if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
      r=$(timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c ns4 $w rep$rep: $r" | tee -a gpurun_out/ab_gemm_ns4.log
    done
  done
done
