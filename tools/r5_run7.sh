R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu3.log 2>&1 || { tail -40 gpurun_out/r5_gpu3.log; exit 1; }
tail -3 gpurun_out/r5_gpu3.log
for a in "--config c1" "--config c2" "--config c2 --ragged" "--config c2 --ragged --global-batch 128" "--config c3 --ragged"; do echo -n "$a: "; python bench.py $a --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail', d['ms_per_step_by_mode'], d['gpu_ms_per_step_by_part'])"
timeout -k 10 200 python tools/tail_bench.py --ragged --steps 20 --layers 4 --heads 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail bert', d['ms_per_step_by_mode'], d['gpu_ms_per_step_by_part'])"
