R=$GRAFT_REPO_ROOT; cd $R
for c in "c2" "c3" "c1" "c2 --ragged" "c5"; do
  for rep in 1 2; do
    for v in peel nopeel; do
      if [ $v = nopeel ]; then export GCGCN_NO_PEEL=1; else unset GCGCN_NO_PEEL; fi
      r=$(timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$c $v rep$rep: $r" | tee -a gpurun_out/ab_peel.log
    done
  done
done
