#!/bin/bash
# All rocprofv3 evidence of a round in one gpurun call (tools/profile_workload.sh per workload); outputs in gpurun_out/prof/.
#   gpurun --timeout 1200 -- 'bash tools/profile_all.sh r03 c2 c3 c5 head producer tail'
set -e
round=$1; shift
for w in "$@"; do
  case $w in
    c1|c2|c3|c5) bash tools/profile_workload.sh ${round}_$w bench.py --config $w --steps 20 --warmup 5 --no-cpu-baseline;;
    head)        bash tools/profile_workload.sh ${round}_head_n64 tools/head_bench.py --steps 5;;
    head42)      bash tools/profile_workload.sh ${round}_head_n42 tools/head_bench.py --N 42 --steps 5;;
    producer)    bash tools/profile_workload.sh ${round}_producer tools/producer_bench.py --steps 5;;
    tail)        bash tools/profile_workload.sh ${round}_tail tools/tail_bench.py --steps 5;;
    tail_ragged) bash tools/profile_workload.sh ${round}_tail_ragged tools/tail_bench.py --ragged --steps 5;;
    head_ragged) bash tools/profile_workload.sh ${round}_head_ragged tools/head_bench.py --ragged --steps 5;;
    c2_ragged)   bash tools/profile_workload.sh ${round}_c2_ragged bench.py --config c2 --ragged --steps 20 --warmup 5 --no-cpu-baseline;;
  esac
  echo "== $w done"
done
ls -la gpurun_out/prof/
