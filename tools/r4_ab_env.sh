# same-session A/B of an ENVIRONMENT switch over several configs: tools/r4_ab_env.sh GCGCN_HOIST_EDGE_TERM "c1 c2 c3" [steps] [extra bench args]
set -e
R=$GRAFT_REPO_ROOT; cd $R
var=$1; cfgs=${2:-"c2"}; steps=${3:-40}; extra=$4
for c in $cfgs; do
  for rep in 1 2; do
    for v in 1 0; do
      r=$(env $var=$v timeout -k 10 200 python bench.py --config $c --steps $steps --warmup 10 --no-cpu-baseline $extra | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c $extra $var=$v rep$rep: $r" | tee -a gpurun_out/ab_$var.log
    done
  done
done
