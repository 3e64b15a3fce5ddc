R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "mha or maggc or fused or full_size or golden or stack or replay" > gpurun_out/r5_gpu7.log 2>&1 || { tail -30 gpurun_out/r5_gpu7.log; exit 1; }
tail -2 gpurun_out/r5_gpu7.log
for v in 1 0 1 0; do echo -n "mha_ride=$v c3: "; GCGCN_MHA_RIDE=$v python bench.py --config c3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
bash tools/tl.sh c3 > /dev/null 2>&1; sed -n 14,20p gpurun_out/timeline_c3.txt | cut -c1-110
