#!/bin/bash
# Round-4 evidence in one gpurun call: bench lines (with CPU baseline for the default config), rocprofv3 stats + PMC passes per
# workload (tools/profile_all.sh), per-launch timelines, the chain kernels' phase trace.  Outputs in gpurun_out/ (copy to profiles/).
set -e
R=$GRAFT_REPO_ROOT; cd $R
tag=${1:-r04}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_c2.json
for c in c1 c3 c5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/${tag}_bench_$c.json; done
timeout -k 10 300 python bench.py --config c2 --ragged --no-cpu-baseline > gpurun_out/${tag}_bench_c2_ragged.json
timeout -k 10 300 python bench.py --config c3 --ragged --no-cpu-baseline > gpurun_out/${tag}_bench_c3_ragged.json
timeout -k 10 300 python bench.py --config c2 --ragged --global-batch 128 --no-cpu-baseline > gpurun_out/${tag}_bench_c2_ragged_b128.json
echo "== bench lines done"
for c in c1 c2 c3 c5; do bash tools/tl.sh $c > /dev/null; cp gpurun_out/timeline_$c.txt gpurun_out/${tag}_${c}_step_timeline.txt; done
bash tools/tl.sh c2 --ragged > /dev/null; cp gpurun_out/timeline_c2.txt gpurun_out/${tag}_c2_ragged_step_timeline.txt
echo "== timelines done"
if [ -f build/trace.so ]; then
  for c in c2 c3; do GCGCN_LIB=$R/build/trace.so timeout -k 10 120 python tools/trace_chain.py --config $c > gpurun_out/${tag}_chain_phase_trace_$c.txt; done
  echo "== trace done"
fi
bash tools/profile_all.sh $tag c2 c3 c5 c2_ragged
