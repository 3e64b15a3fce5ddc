#!/bin/bash
# Build the library of another git revision next to the working tree's, for same-session A/B timing:
#   tools/ab_build.sh <rev>   ->  gpurun_ab_old.so     (then: GCGCN_LIB=$PWD/gpurun_ab_old.so python bench.py ...)
set -e
rev=${1:-HEAD}
d=/tmp/ab_src
rm -rf $d && mkdir -p $d/csrc $d/include
for f in api.hip gemm.hip edge.hip rowops.hip chain.hip mha_core.hip common.hpp gemm.hpp gemm_body.hpp rowops.hpp gcn_plan.hpp; do
  git show $rev:gcgcn_amd/csrc/$f > $d/csrc/$f
done
git show $rev:include/gcgcn.h > $d/include/gcgcn.h
sed -i "s#\"../../include/gcgcn.h\"#\"$d/include/gcgcn.h\"#" $d/csrc/api.hip
cd $d/csrc
for f in api gemm edge rowops chain mha_core; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o $f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/gpurun_ab_old.so api.o gemm.o edge.o rowops.o chain.o mha_core.o
ls -la /root/repo/gpurun_ab_old.so
