R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 python -m pytest tests/test_tail_gpu.py -x -q 2>&1 | tail -15
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 > gpurun_out/r5_tail_ragged.json 2> gpurun_out/r5_tail_ragged.err; tail -3 gpurun_out/r5_tail_ragged.err; cat gpurun_out/r5_tail_ragged.json
