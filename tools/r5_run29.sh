R=$GRAFT_REPO_ROOT; cd $R
for c in "c2"; do
  for rep in 1 2; do
    for v in 50 75 100 25; do
      r=$(GCGCN_CHAIN_CARRY_PCT=$v timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c chain_carry_pct=$v rep$rep: $r" | tee -a gpurun_out/ab_chain_carry_pct.log
    done
  done
done
