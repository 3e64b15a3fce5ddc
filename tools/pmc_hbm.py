#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_hbm.sh: average FETCH_SIZE / WRITE_SIZE (KiB) per kernel name and
the corrected HBM bytes per launch, (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950's FETCH_SIZE reports half of wide coalesced
reads, MI355X_MICROARCH.md; check: the E2 mean pass riding in gcn_chain_fwd reads 134 217 728 B at cfg 2)."""
import csv, glob, json, os, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
for c in acc:
    files = glob.glob(os.path.join(src, c, "**", "*counter_collection.csv"), recursive=True)
    f = max(files, key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[c][r["Kernel_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    fe = acc["FETCH_SIZE"].get(k, [0.0]); wr = acc["WRITE_SIZE"].get(k, [0.0])
    fa, wa = sum(fe) / len(fe), sum(wr) / len(wr)
    kernels[k] = {"launches": len(fe), "fetch_KiB_raw": round(fa, 1), "write_KiB_raw": round(wa, 1),
                  "hbm_bytes_per_launch_corrected": int(round((2 * fa + wa) * 1024))}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_hbm.sh), bench.py --mode eager "
                   "--steps 3 --warmup 2, cfg c2 (B=32,N=64,D=256,L=2,H=8); corrected bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB "
                   "(gfx950 FETCH_SIZE reports half of wide coalesced reads); averages over all launches of a kernel name (GEMM "
                   "kernels carry different problems under one name; edge_bwd_carry also reads the operands of the parked "
                   "weight-gradient products it carries)", "kernels": kernels}, open(dst, "w"), indent=1)
for k, v in kernels.items():
    if "edge" in k or "chain" in k:
        print(k[:70], v)
