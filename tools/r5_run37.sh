R=$GRAFT_REPO_ROOT; cd $R
work=/tmp/w_prod; rm -rf $work; mkdir -p $work
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $work -o st -- python3 $R/tools/producer_bench.py --ids uint8 --steps 20 --mode eager > $work/log 2>&1)
f=$(find $work -name "*kernel_stats.csv" | head -1)
python3 tools/kstats.py $f 40
