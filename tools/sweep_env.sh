#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench args]  -- ms/step of bench.py for each value of an env knob
var=$1; vals=$2; shift 2
for v in $vals; do
  echo -n "$var=$v $*: "
  env $var=$v timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --prof-kernel edge_bwd "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
