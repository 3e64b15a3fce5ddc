R=$GRAFT_REPO_ROOT; cd $R
for c in c2 c3 c1; do
  for pct in 250 125 175 350 250; do
      r=$(GCGCN_SPLIT_PCT=$pct timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c split_pct=$pct: $r" | tee -a gpurun_out/ab_split_pct.log
  done
done
