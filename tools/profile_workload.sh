#!/bin/bash
# rocprofv3 evidence for one workload, as MI355X_MICROARCH.md prescribes (the program itself directly after `--`, counters
# in their own passes next to --kernel-trace only):
#   1. --kernel-trace --stats                     -> <tag>_kernel_stats.csv
#   2. --pmc FETCH_SIZE                           \
#   3. --pmc WRITE_SIZE                            > tools/pmc_summary.py -> <tag>_pmc.json
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  /
# Run on the GPU box from the repo root:
#   bash tools/profile_workload.sh r03_c3 bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline
#   bash tools/profile_workload.sh r03_head_n64 tools/head_bench.py --steps 5
# Outputs land in gpurun_out/prof/ (gpurun merges only gpurun_out/): copy what is judged into profiles/.
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/prof
work=$out/work_$tag
mkdir -p $out && rm -rf $work && mkdir -p $work
prog=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $work/stats -o st -- python3 $prog "$@" > $work/stats.log 2>&1
find $work/stats -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
# counter passes on a short eager run (a replayed hipGraph reports its kernels too, but three steps are enough to average)
pmcargs="$@"
case "$prog" in */bench.py) pmcargs="$pmcargs --mode eager --steps 3 --warmup 2";; esac
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$work/$(echo $c | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o r -- python3 $prog $pmcargs > $d.log 2>&1
done
cd $root
python3 tools/pmc_summary.py $work $out/${tag}_pmc.json "$(basename $prog) $pmcargs"
rm -rf $work
