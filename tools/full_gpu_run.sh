# One gpurun call for the end-of-round evidence: GPU test suite, bench lines of every config, rocprofv3 kernel stats and the
# per-dispatch timeline of the default bench command (outputs under gpurun_out/; copy what is judged into profiles/).
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu.log 2>&1 || { tail -30 gpurun_out/full_gpu.log; exit 1; }
tail -3 gpurun_out/full_gpu.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json
for c in c1 c3 c5; do timeout -k 10 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$c.json; done
timeout -k 10 200 python bench.py --ragged --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c2_ragged.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_v2 -o st -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/stats_bench.log 2>&1
find $R/gpurun_out/stats_v2 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02_c2_default_bench_kernel_stats_v2.csv \;
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 && python3 $R/tools/timeline.py $R/gpurun_out/tl > $R/gpurun_out/r02_c2_step_timeline_v2.txt
rm -rf $R/gpurun_out/stats_v2 $R/gpurun_out/tl
for f in c2 c1 c3 c5 c2_ragged; do python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
