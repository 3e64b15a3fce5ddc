# round 5, first GPU call after the chain_t rewrite: the GPU suite, then bench lines of the configs chain_t serves
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu1.log 2>&1 || { tail -40 gpurun_out/r5_gpu1.log; exit 1; }
tail -3 gpurun_out/r5_gpu1.log
for c in c3 c1 c2; do timeout -k 10 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_$c.json 2> gpurun_out/r5_bench_$c.err || { tail -5 gpurun_out/r5_bench_$c.err; exit 1; }; done
timeout -k 10 200 python bench.py --config c3 --ragged --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_c3_ragged.json 2>/dev/null
timeout -k 10 200 python bench.py --ragged --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_c2_ragged.json 2>/dev/null
for f in c3 c1 c2 c3_ragged c2_ragged; do python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/r5_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
