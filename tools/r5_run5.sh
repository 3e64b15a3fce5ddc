R=$GRAFT_REPO_ROOT; cd $R
cd /tmp && export TMPDIR=/tmp
for d in 0 2 4 6; do
  rm -rf $R/gpurun_out/ks; GCGCN_PROD_TOK_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks -o st -- python3 $R/tools/producer_bench.py --ids uint8 --steps 5 --mode eager > /dev/null 2>&1
  f=$(find $R/gpurun_out/ks -name "*kernel_stats.csv" | head -1); echo "dbg=$d $(grep tok_mfma $f | awk -F, '{print $(NF-4)}')" >> $R/gpurun_out/r5_tok_dbg.txt
done
rm -rf $R/gpurun_out/ks; cat $R/gpurun_out/r5_tok_dbg.txt
