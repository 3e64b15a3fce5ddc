# what the driver runs at round end: the GPU suite, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5_validate_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_validate_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 400 python bench.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','dtype')}); print(d['roofline']); print(d['cpu_baseline'])"
