R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu2.log 2>&1 || { tail -40 gpurun_out/r5_gpu2.log; exit 1; }
tail -3 gpurun_out/r5_gpu2.log
for c in c1 c2; do python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"; done
for v in 1 2; do for rg in "" "--ragged"; do for rep in 1 2; do echo -n "chain_t=$v $rg: "; GCGCN_CHAIN_T=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline $rg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done; done; done
echo -n "c1 chain_t_fuse=0: "; GCGCN_CHAIN_T_FUSE=0 python bench.py --config c1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
