#!/bin/bash
# Build the library of another git revision next to the working tree's, for same-session A/B timing:
#   tools/ab_build.sh <rev>   ->  build/ab_old.so     (then: GCGCN_LIB=$PWD/build/ab_old.so python bench.py ...)
# Only meaningful while both revisions speak the same C ABI (include/gcgcn.h).
set -e
rev=${1:-HEAD}
d=/tmp/ab_src
rm -rf $d && mkdir -p $d/csrc $d/include
for f in $(git ls-tree --name-only $rev gcgcn_amd/csrc/ | grep -E '\.(hip|hpp)$'); do
  git show $rev:$f > $d/csrc/$(basename $f)
done
git show $rev:include/gcgcn.h > $d/include/gcgcn.h
sed -i "s#\"../../include/gcgcn.h\"#\"$d/include/gcgcn.h\"#" $d/csrc/api.hip
cd $d/csrc
objs=""
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o ${f%.hip}.o &
  objs="$objs ${f%.hip}.o"
done
wait
mkdir -p /root/repo/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build/ab_old.so $objs
ls -la /root/repo/build/ab_old.so
