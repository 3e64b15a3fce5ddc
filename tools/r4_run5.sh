set -e
R=$GRAFT_REPO_ROOT; cd $R
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
for cfg in "--config c2" "--config c2 --global-batch 128" "--config c3" "--config c1"; do
  for rep in 1 2; do for v in 1 0; do
    echo "ragged $cfg row_blocks=$v: $(GCGCN_ROW_BLOCKS=$v timeout -k 10 200 python bench.py $cfg --ragged --steps 40 --warmup 10 --no-cpu-baseline | python3 -c "$pick")" | tee -a gpurun_out/ab_row_blocks.log
  done; done
done
