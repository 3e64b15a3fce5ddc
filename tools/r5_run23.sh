# re-test of the block path's tuning knobs after this round's kernel changes: each against the default, same session
R=$GRAFT_REPO_ROOT; cd $R
run() { # label, env assignment
  for c in c2 c3; do
    r=$(env $2 timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$c $1: $r" | tee -a gpurun_out/ab_knobs.log
  done
}
run default X=1
run carry_spread=70 GCGCN_CARRY_SPREAD=70
run carry_spread=100 GCGCN_CARRY_SPREAD=100
run carry_cohort=128 GCGCN_CARRY_COHORT=128
run carry_cohort=512 GCGCN_CARRY_COHORT=512
run chain_spread=70 GCGCN_CHAIN_SPREAD=70
run chain_spread=100 GCGCN_CHAIN_SPREAD=100
run chain_cohort=64 GCGCN_CHAIN_COHORT=64
run chain_cohort=256 GCGCN_CHAIN_COHORT=256
run default X=1
run chain_carry=0 GCGCN_CHAIN_CARRY=0
run gat_ride=0 GCGCN_GAT_RIDE=0
run mha_ride=0 GCGCN_MHA_RIDE=0
run fold_slices=16 GCGCN_FOLD_SLICES=16
run fold_slices=64 GCGCN_FOLD_SLICES=64
run head_sum_fold=0 GCGCN_HEAD_SUM_FOLD=0
run att_in_chain=0 GCGCN_ATT_IN_CHAIN=0
run chain_t=2 GCGCN_CHAIN_T=2
run default X=1
