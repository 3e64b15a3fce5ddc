for i in 1 2 3; do
  for d in 1 0; do
    echo -n "defer=$d: "; GCGCN_DEFER=$d timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm']['avg_launch_us'])"
  done
done
for d in 1 0; do echo -n "c3 defer=$d: "; GCGCN_DEFER=$d timeout -k 10 120 python bench.py --config c3 --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
for d in 1 0; do echo -n "c5 defer=$d: "; GCGCN_DEFER=$d timeout -k 10 120 python bench.py --config c5 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
