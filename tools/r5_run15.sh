# ragged: per-kernel times, old (live 32-row k-tiles) against new (pairs of live 16-row blocks); new first this time
R=$GRAFT_REPO_ROOT; cd $R
for c in c2 c3; do
for w in new old; do
  if [ $w = old ]; then export GCGCN_LIB=$R/build/ab_old.so; else unset GCGCN_LIB; fi
  work=/tmp/w_$w; rm -rf $work; mkdir -p $work
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $work -o st -- python3 $R/bench.py --config $c --ragged --mode eager --steps 20 --warmup 5 --no-cpu-baseline > $work/log 2>&1)
  f=$(find $work -name "*kernel_stats.csv" | head -1)
  echo "== $c $w"; python3 tools/kstats.py $f 30 | grep -E "edge_bwd_carry|gemm_group|chain_t_bwd"
done
done
