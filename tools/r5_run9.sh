R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "chain_t or deterministic or fused or full_size" > gpurun_out/r5_gpu5.log 2>&1 || { tail -30 gpurun_out/r5_gpu5.log; exit 1; }
tail -2 gpurun_out/r5_gpu5.log
for a in "--config c3" "--config c3" "--config c3 --ragged" "--config c1" "--config c2 --ragged"; do echo -n "$a: "; python bench.py $a --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
bash tools/tl.sh c3 > /dev/null 2>&1; grep "chain" gpurun_out/timeline_c3.txt | cut -c1-110
