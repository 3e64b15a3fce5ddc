run() { # label lib args
  if [ "$1" = old ]; then export GCGCN_LIB=$PWD/build/ab_old.so; else unset GCGCN_LIB; fi
  shift
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --prof-kernel edge_bwd "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], end=' ')"
}
for cfg in "--config c2 --mode graph" "--config c2" "--config c3"; do
  for w in old new old new; do echo -n "$cfg $w: "; run $w $cfg; echo; done
done
