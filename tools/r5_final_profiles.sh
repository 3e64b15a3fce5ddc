#!/bin/bash
# Round-5 evidence in one gpurun call: bench lines (CPU baseline on the default config), per-launch timelines, the chain kernels'
# phase trace (build/trace.so: make -C gcgcn_amd/csrc trace), rocprofv3 stats + PMC passes per workload (tools/profile_all.sh),
# the rows either side of the path in both timing modes.  Outputs in gpurun_out/ (copy what is judged into profiles/).
set -e
R=$GRAFT_REPO_ROOT; cd $R
tag=${1:-r05}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_c2.json
for c in c1 c3 c5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/${tag}_bench_$c.json; done
timeout -k 10 300 python bench.py --config c2 --ragged --no-cpu-baseline > gpurun_out/${tag}_bench_c2_ragged.json
timeout -k 10 300 python bench.py --config c3 --ragged --no-cpu-baseline > gpurun_out/${tag}_bench_c3_ragged.json
timeout -k 10 300 python bench.py --config c2 --ragged --global-batch 128 --no-cpu-baseline > gpurun_out/${tag}_bench_c2_ragged_b128.json
echo "== bench lines done"
for c in c1 c2 c3 c5; do bash tools/tl.sh $c > /dev/null; cp gpurun_out/timeline_$c.txt gpurun_out/${tag}_${c}_step_timeline.txt; done
bash tools/tl.sh c2 --ragged > /dev/null; cp gpurun_out/timeline_c2.txt gpurun_out/${tag}_c2_ragged_step_timeline.txt
bash tools/tl.sh c3 --ragged > /dev/null; cp gpurun_out/timeline_c3.txt gpurun_out/${tag}_c3_ragged_step_timeline.txt
echo "== timelines done"
if [ -f build/trace.so ]; then
  for c in c1 c2 c3; do GCGCN_LIB=$R/build/trace.so timeout -k 10 120 python tools/trace_chain.py --config $c --iters 2 > gpurun_out/${tag}_chain_phase_trace_$c.txt 2>&1; done
  echo "== trace done"
fi
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 > gpurun_out/${tag}_tail_bench_ragged.json 2>/dev/null
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 --layers 4 --heads 4 > gpurun_out/${tag}_tail_bench_ragged_bert.json 2>/dev/null
timeout -k 10 200 python tools/head_bench.py --ragged --steps 20 > gpurun_out/${tag}_head_bench_ragged.json 2>/dev/null
timeout -k 10 200 python tools/head_bench.py --steps 10 > gpurun_out/${tag}_head_bench_n64.json 2>/dev/null
timeout -k 10 200 python tools/producer_bench.py --ids uint8 --steps 20 > gpurun_out/${tag}_producer_bench.json 2>/dev/null
timeout -k 10 300 python tools/train_step_bench.py --steps 10 > gpurun_out/${tag}_train_step_bench.json 2>/dev/null
echo "== row benches done"
bash tools/profile_all.sh $tag c2 c3 c5 c2_ragged
