R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_producer_gpu.py tests/test_tail_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -5
timeout -k 10 200 python tools/producer_bench.py --ids uint8 --steps 20 > gpurun_out/r5_producer.json 2> gpurun_out/r5_producer.err; tail -2 gpurun_out/r5_producer.err; python3 -c "
import json; d=json.load(open('gpurun_out/r5_producer.json')); print(d['value'], d['ms_per_step_by_mode'], {k:v['ms_per_step'] for k,v in d['time_shares_ms_per_step'].items()})"
bash tools/profile_workload.sh r05_producer tools/producer_bench.py --ids uint8 --steps 5 --mode eager > gpurun_out/r5_prof_producer.log 2>&1; tail -3 gpurun_out/r5_prof_producer.log
python3 -c "
import json; d=json.load(open('gpurun_out/prof/r05_producer_pmc.json'))['kernels']; [print(k[:70], {a:b for a,b in v.items() if a in ('launches','hbm_bytes','hbm_MB','fetch_kib','write_kib')}) for k,v in d.items() if 'prod_word' in k]"
