R=$GRAFT_REPO_ROOT; cd $R
for c in "c2 --ragged" "c3 --ragged" "c1 --ragged"; do
  for rep in 1 2; do
    for v in 100 50 35; do
      r=$(GCGCN_CHAIN_T_RB_PCT=$v timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c rb_pct=$v rep$rep: $r" | tee -a gpurun_out/ab_chain_t_rb_pct.log
    done
  done
done
GCGCN_CHAIN_T_RB_PCT=50 bash tools/tl.sh c2 --ragged | tail -10
