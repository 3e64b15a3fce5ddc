R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py tests/test_bench_gpu.py tests/test_tail_gpu.py tests/test_head_gpu.py tests/test_producer_gpu.py tests/test_loss_gpu.py tests/test_optim_gpu.py tests/test_data_gpu.py -m gpu -x -q 2>&1 | tail -4
