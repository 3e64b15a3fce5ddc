# row-block k-tiles of two live 16-row blocks (new) against live 32-row k-tiles (build/ab_old.so = HEAD): parity, then A/B
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -x -q -k "row_block or ragged or empty_document or determin" > gpurun_out/r5_run13_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_run13_tests.log
for c in c2 c3; do
  for rep in 1 2; do
    for w in old new; do
      if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
      r=$(timeout -k 10 200 python bench.py --config $c --ragged --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c ragged $w rep$rep: $r" | tee -a gpurun_out/ab_rowblk16.log
    done
  done
done
unset GCGCN_LIB
timeout -k 10 200 python bench.py --config c2 --ragged --global-batch 128 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=128 new', d['ms_per_step'], d['value'])"
GCGCN_LIB=$R/build/ab_old.so timeout -k 10 200 python bench.py --config c2 --ragged --global-batch 128 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=128 old', d['ms_per_step'], d['value'])"
