R=$GRAFT_REPO_ROOT; cd $R
for c in c3 c2 "c3 --ragged" c5; do
  for rep in 1 2; do
    for v in 0 36 56; do
      r=$(GCGCN_SPLIT_MIN_ITERS=$v timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c split_min_iters=$v rep$rep: $r" | tee -a gpurun_out/ab_split_min_iters.log
    done
  done
done
