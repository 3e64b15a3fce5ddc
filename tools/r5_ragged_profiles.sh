#!/bin/bash
# Round-5 refresh after the row-block changes: the ragged workloads' bench lines, timelines, kernel stats and PMC passes (+ the
# default bench line of the same box).  Outputs in gpurun_out/ (copy what is judged into profiles/).
set -e
R=$GRAFT_REPO_ROOT; cd $R
tag=r05
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_c2.json
timeout -k 10 300 python bench.py --config c2 --ragged --no-cpu-baseline > gpurun_out/${tag}_bench_c2_ragged.json
timeout -k 10 300 python bench.py --config c3 --ragged --no-cpu-baseline > gpurun_out/${tag}_bench_c3_ragged.json
timeout -k 10 300 python bench.py --config c2 --ragged --global-batch 128 --no-cpu-baseline > gpurun_out/${tag}_bench_c2_ragged_b128.json
timeout -k 10 300 python bench.py --config c1 --ragged --no-cpu-baseline > gpurun_out/${tag}_bench_c1_ragged.json
echo "== bench lines done"
bash tools/tl.sh c2 --ragged > /dev/null; cp gpurun_out/timeline_c2.txt gpurun_out/${tag}_c2_ragged_step_timeline.txt
bash tools/tl.sh c3 --ragged > /dev/null; cp gpurun_out/timeline_c3.txt gpurun_out/${tag}_c3_ragged_step_timeline.txt
echo "== timelines done"
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 > gpurun_out/${tag}_tail_bench_ragged.json 2>/dev/null
timeout -k 10 300 python tools/train_step_bench.py --steps 10 > gpurun_out/${tag}_train_step_bench.json 2>/dev/null
echo "== row benches done"
bash tools/profile_all.sh $tag c2_ragged
