R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q > gpurun_out/r5_gpu6.log 2>&1 || { tail -30 gpurun_out/r5_gpu6.log; exit 1; }
tail -2 gpurun_out/r5_gpu6.log
for v in 1 0 1 0; do echo -n "ride_split=$v c2: "; GCGCN_RIDE_SPLIT=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
bash tools/tl.sh c2 > /dev/null 2>&1; sed -n 4,7p gpurun_out/timeline_c2.txt | cut -c1-110
