R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -x -q -k "row_block or ragged or empty_document or determin" > gpurun_out/r5_run17_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_run17_tests.log
for c in c2 c3; do
  for rep in 1 2; do
    for v in 1 0; do
      r=$(GCGCN_SPLIT_WIDEN=$v timeout -k 10 200 python bench.py --config $c --ragged --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c ragged split_widen=$v rep$rep: $r" | tee -a gpurun_out/ab_split_widen.log
    done
  done
done
for v in 1 0; do
  work=/tmp/w_$v; rm -rf $work; mkdir -p $work
  (cd /tmp && export TMPDIR=/tmp && GCGCN_SPLIT_WIDEN=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $work -o st -- python3 $R/bench.py --config c2 --ragged --mode eager --steps 20 --warmup 5 --no-cpu-baseline > $work/log 2>&1)
  f=$(find $work -name "*kernel_stats.csv" | head -1)
  echo "== split_widen=$v"; python3 tools/kstats.py $f 12
done
