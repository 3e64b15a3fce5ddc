#!/bin/bash
# Register / scratch / LDS use of every kernel in an object file built for gfx950:
#   tools/kres.sh gcgcn_amd/csrc/gemm.o [name-filter]
# (extracts the device code object with llvm-objdump --offloading and reads its AMDGPU metadata notes)
set -e
f=$(readlink -f $1); pat=${2:-.}
d=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
cp $f $d/x.o
(cd $d && $B/llvm-objdump --offloading x.o > /dev/null)
co=$(ls $d/x.o.*gfx950* | head -1)
$B/llvm-readelf --notes $co | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
blocks=txt.split('- .agpr_count:')[1:]
rows=[]
for b in blocks:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', b) or [None,'?'])[1]
    rows.append((g('vgpr_count'), b.split()[0], g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('name')))
names=subprocess.run(['c++filt'],input='\n'.join(r[6] for r in rows),capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    print('vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s  %s' % (r[0],r[1],r[2],r[3],r[4],r[5],n[:120]))
" | grep -E "$pat"
rm -rf $d
