# ragged cfg 2: what is inside the 47 us group launch behind MAGGC's chain backward? mha_ride on / off, kernel stats of each
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/prof
for v in 1 0; do
  work=/tmp/w_$v; rm -rf $work; mkdir -p $work
  (cd /tmp && export TMPDIR=/tmp && GCGCN_MHA_RIDE=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $work -o st -- python3 $R/bench.py --config c2 --ragged --mode eager --steps 20 --warmup 5 --no-cpu-baseline > $work/log 2>&1)
  f=$(find $work -name "*kernel_stats.csv" | head -1)
  echo "== mha_ride=$v"; python3 tools/kstats.py $f 14
done
