#!/bin/bash
# per-dispatch timeline of the last step of any bench script:  bash tools/tl_any.sh <first-kernel-substring> <script> [args]   (on the GPU box)
key=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_any
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_any -o tl -- python3 $R/"$@" > /dev/null 2>&1
python3 - /tmp/tl_any "$key" <<'PY'
import csv, glob, os, sys
d, key = sys.argv[1], sys.argv[2]
f = max(glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    nm = r['Kernel_Name'].replace('void gc::', '').replace('gc::', '')[:60]
    gx = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    print(f"{(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:6.1f} blocks=({gx},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {nm}")
print('step span us', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'busy us', busy / 1e3, 'launches', b - a)
PY
