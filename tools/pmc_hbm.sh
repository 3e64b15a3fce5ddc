#!/bin/bash
# HBM bytes per kernel launch from the PMC counters, as MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc passes for
# FETCH_SIZE and WRITE_SIZE (kernel trace only, nothing else traced), then tools/pmc_hbm.py -> profiles/r02_c2_pmc_hbm.json.
# Run on the GPU box from the repo root:  bash tools/pmc_hbm.sh
set -e
root=$PWD
out=$root/gpurun_out/pmc
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o r -- python3 $root/bench.py --mode eager --steps 3 --warmup 2 --no-cpu-baseline > $out/$c.log 2>&1
done
cd $root
python3 tools/pmc_hbm.py $out gpurun_out/pmc/r02_c2_pmc_hbm.json   # gpurun merges only gpurun_out/: copy it to profiles/ afterwards
