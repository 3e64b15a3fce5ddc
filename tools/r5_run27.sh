R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r5_run27_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_run27_tests.log
for c in "c2" "c1" "c3" "c2 --ragged" "c5"; do
  for rep in 1 2; do
    for v in 1 0; do
      r=$(GCGCN_GEMM_PANEL=$v timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c gemm_panel=$v rep$rep: $r" | tee -a gpurun_out/ab_gemm_panel.log
    done
  done
done
bash tools/tl.sh c2 | tail -22
