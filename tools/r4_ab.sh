# same-session A/B of one run-time option over several configs: tools/r4_ab.sh OPTION "c1 c2 c3" [steps]
set -e
R=$GRAFT_REPO_ROOT; cd $R
opt=$1; cfgs=${2:-"c2"}; steps=${3:-40}
up=$(echo $opt | tr a-z A-Z)
for c in $cfgs; do
  for rep in 1 2; do
    for v in 1 0; do
      r=$(env GCGCN_$up=$v timeout -k 10 200 python bench.py --config $c --steps $steps --warmup 10 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c $opt=$v rep$rep: $r" | tee -a gpurun_out/ab_$opt.log
    done
  done
done
