#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/profile_workload.sh per kernel name.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
streaming reads (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on edge_fwd_kernel<4,false>, which reads exactly
134.2 MB at cfg 2).  Matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's MFMA pipe is busy, summed
over the chip (v_mfma_f32_32x32x2_f32 = 64 cycles per instruction on its SIMD, so BUSY * 64 flop/cycle = the fp32 flops
executed -- the `flops_from_mfma_busy` column can be checked against 2MNK); mfma_busy = BUSY / (GRBM_GUI_ACTIVE / 8 XCDs *
1024 SIMDs) = the fraction of the kernel's duration the average SIMD's matrix pipe was busy.

    python tools/pmc_summary.py <dir with FETCH_SIZE/ WRITE_SIZE/ SQ_VALU_MFMA_BUSY_CYCLES/ subdirs> <out.json> [note]
"""
import csv, glob, json, os, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
SIMDS, XCDS = 1024, 8


def read(sub, names):
    acc = {n: defaultdict(list) for n in names}
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return acc
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fe = read("FETCH_SIZE", ["FETCH_SIZE"])["FETCH_SIZE"]
wr = read("WRITE_SIZE", ["WRITE_SIZE"])["WRITE_SIZE"]
sq = read("SQ_VALU_MFMA_BUSY_CYCLES", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
avg = lambda v: sum(v) / len(v) if v else None
kernels = {}
for k in sorted(set(fe) | set(wr) | set(sq["SQ_VALU_MFMA_BUSY_CYCLES"])):
    e = {"launches": len(fe.get(k, [])) or len(sq["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, []))}
    fa, wa = avg(fe.get(k, [])), avg(wr.get(k, []))
    if fa is not None and wa is not None:
        e.update(fetch_KiB_raw=round(fa, 1), write_KiB_raw=round(wa, 1), hbm_bytes_per_launch_corrected=int(round((2 * fa + wa) * 1024)))
    mb, gui, sb = (avg(sq[n].get(k, [])) for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"))
    if mb is not None and gui:
        e.update(mfma_busy_cycles=round(mb, 0), grbm_gui_active=round(gui, 0), sq_busy_cycles=None if sb is None else round(sb, 0),
                 mfma_busy=round(mb / (gui / XCDS * SIMDS), 4), flops_from_mfma_busy=round(mb * 64.0, 0))
    kernels[k] = e
json.dump({"note": "rocprofv3 --pmc passes (tools/profile_workload.sh): FETCH_SIZE, WRITE_SIZE, {SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, "
                   "GRBM_GUI_ACTIVE}, each with --kernel-trace only; averages over all launches of a kernel name (a GEMM kernel carries "
                   "different problems under one name).  hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB; mfma_busy = MFMA busy cycles / "
                   "(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).  Command: " + note, "kernels": kernels}, open(dst, "w"), indent=1)
rows = sorted(kernels.items(), key=lambda kv: -(kv[1].get("mfma_busy_cycles") or 0))
for k, v in rows[:14]:
    print(k[:64].ljust(64), "mfma_busy", v.get("mfma_busy"), "hbm MB", round(v.get("hbm_bytes_per_launch_corrected", 0) / 1e6, 1))
