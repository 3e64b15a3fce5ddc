#!/bin/bash
# the rows either side of the path, both timing modes (outputs in gpurun_out/; copy into profiles/)
R=$GRAFT_REPO_ROOT; cd $R
tag=r05
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 > gpurun_out/${tag}_tail_bench_ragged.json 2>/dev/null
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 --layers 4 --heads 4 > gpurun_out/${tag}_tail_bench_ragged_bert.json 2>/dev/null
timeout -k 10 200 python tools/head_bench.py --ragged --steps 20 > gpurun_out/${tag}_head_bench_ragged.json 2>/dev/null
timeout -k 10 200 python tools/head_bench.py --steps 10 > gpurun_out/${tag}_head_bench_n64.json 2>/dev/null
timeout -k 10 200 python tools/producer_bench.py --ids uint8 --steps 20 > gpurun_out/${tag}_producer_bench.json 2>/dev/null
timeout -k 10 300 python tools/train_step_bench.py --steps 10 > gpurun_out/${tag}_train_step_bench.json 2>/dev/null
bash tools/profile_all.sh $tag producer tail_ragged head_ragged
