# One gpurun call: GPU test suite, then the bench line of every config (outputs under gpurun_out/).
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh tests bench'      (any subset of: tests bench)
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for what in "$@"; do
  case $what in
    tests)
      timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu.log 2>&1 || { tail -40 gpurun_out/full_gpu.log; exit 1; }
      tail -3 gpurun_out/full_gpu.log;;
    bench)
      timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json
      for c in c1 c3 c5; do timeout -k 10 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$c.json; done
      timeout -k 10 200 python bench.py --ragged --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c2_ragged.json
      for f in c2 c1 c3 c5 c2_ragged; do python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), {k: v['ms_per_step'] for k, v in d['time_shares_ms_per_step'].items()})"; done;;
  esac
done
