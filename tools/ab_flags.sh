#!/bin/bash
# Build the working tree's library with extra compiler flags next to the default build, for same-session A/B timing:
#   tools/ab_flags.sh "-DGC_ROWBLOCKS=0" norb   ->  build/ab_norb.so   (then: GCGCN_LIB=$PWD/build/ab_norb.so python bench.py ...)
set -e
flags=$1; name=${2:-alt}
root=$(cd $(dirname $0)/.. && pwd)
d=/tmp/ab_flags_$name
rm -rf $d && mkdir -p $d
cd $root/gcgcn_amd/csrc
objs=""
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $f -o $d/${f%.hip}.o &
  objs="$objs $d/${f%.hip}.o"
done
wait
mkdir -p $root/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/ab_$name.so $objs
ls -la $root/build/ab_$name.so
