R=$GRAFT_REPO_ROOT; cd $R
for c in c3 c2; do
GCGCN_GROUP_DUMP=1 timeout -k 10 200 python bench.py --config $c --ragged --mode eager --steps 1 --warmup 0 --no-cpu-baseline 2> gpurun_out/r5_group_dump_${c}_ragged.txt | tail -1 | cut -c1-60
done
work=/tmp/w_c3; rm -rf $work; mkdir -p $work
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $work -o st -- python3 $R/bench.py --config c3 --ragged --mode eager --steps 6 --warmup 3 --no-cpu-baseline > $work/log 2>&1)
f=$(find $work -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: find last gat_fold_fwd
idx=[i for i,r in enumerate(rows) if "gat_fold_fwd" in r["Kernel_Name"]]
s=idx[-1]
t0=int(rows[s]["Start_Timestamp"])
for r in rows[s:]:
    print("%8.1f dur=%7.1f grid=%s wg=%s %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,int(r["Grid_Size_X"])//max(int(r["Workgroup_Size_X"]),1),r["Workgroup_Size_X"],r["Kernel_Name"][:70]))
PY
