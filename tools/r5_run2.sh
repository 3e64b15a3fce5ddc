# A/B: chain_t (rewritten, round 5) in place of chain_s at cfg 2, dense and ragged; timelines of cfg 3 dense / ragged
R=$GRAFT_REPO_ROOT; cd $R
for v in 1 2; do for rg in "" "--ragged"; do
  echo "chain_t=$v $rg" >> gpurun_out/r5_ab_chain_t.txt
  for rep in 1 2; do GCGCN_CHAIN_T=$v timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $rg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], d['ms_per_step'])" >> gpurun_out/r5_ab_chain_t.txt; done
done; done
cat gpurun_out/r5_ab_chain_t.txt
bash tools/tl.sh c3 > /dev/null; bash tools/tl.sh c3 --ragged > /dev/null 2>&1; mv gpurun_out/timeline_c3.txt gpurun_out/r5_timeline_c3_ragged.txt; bash tools/tl.sh c3 > /dev/null; mv gpurun_out/timeline_c3.txt gpurun_out/r5_timeline_c3.txt
bash tools/tl.sh c2 --ragged > /dev/null 2>&1; mv gpurun_out/timeline_c2.txt gpurun_out/r5_timeline_c2_ragged.txt
bash tools/tl.sh c2 > /dev/null 2>&1; mv gpurun_out/timeline_c2.txt gpurun_out/r5_timeline_c2.txt
