R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "row_block or ragged or empty_document" 2>&1 | tail -2
for c in "c2 --ragged" "c3 --ragged"; do
  for rep in 1 2; do
    for w in old new; do
      if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
      r=$(timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c widen-floor16 $w rep$rep: $r" | tee -a gpurun_out/ab_widen_floor.log
    done
  done
done
