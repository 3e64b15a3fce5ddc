R=$GRAFT_REPO_ROOT; cd $R
for c in c2 c3 c1; do
GCGCN_GROUP_DUMP=1 timeout -k 10 200 python bench.py --config $c --mode eager --steps 1 --warmup 0 --no-cpu-baseline 2> gpurun_out/r5_group_dump_${c}.txt | tail -1 | cut -c1-60
done
