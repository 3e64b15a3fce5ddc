#!/bin/bash
# per-dispatch timeline of one step of a config:  bash tools/tl.sh c3 [extra bench.py args]   (run on the GPU box)
cfg=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_$cfg
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$cfg -o tl -- python3 $R/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 $R/tools/timeline.py $R/gpurun_out/tl_$cfg > $R/gpurun_out/timeline_$cfg.txt
rm -rf $R/gpurun_out/tl_$cfg
cat $R/gpurun_out/timeline_$cfg.txt
