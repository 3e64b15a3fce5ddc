#!/usr/bin/env python3
"""Print the per-dispatch timeline of the last full step from a rocprofv3 --kernel-trace CSV."""
import csv, glob, os, sys
d = sys.argv[1]
f = max(glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'gat_fold_fwd' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    nm = r['Kernel_Name'].replace('void gc::', '').replace('gc::', '')[:56]
    gx = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    print(f"{(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:6.1f} blocks=({gx},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {nm}")
print('step span us', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'busy us', busy / 1e3, 'launches', b - a)
