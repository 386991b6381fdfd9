"""summary of tools/alu_roofline.sh: per kernel, mean duration (kernel trace) and VALU counters -> the ALU roofline figures bench.py
loads from profiles/alu_roofline.json:
  valu_busy      = 4 x SQ_ACTIVE_INST_VALU / (SIMDs x cycles)      (quad-cycle units, MI355X_MICROARCH.md; cycles from GRBM_GUI_ACTIVE / 8)
  achieved       = SQ_INSTS_VALU x 64 lanes / duration              lane-operations per second, all VALU instructions
  peak_fp64      = 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz  = 39.3 T fp64 lane-operations/s (78.6 TFLOP/s counts an FMA as two)
usage: python tools/alu_roofline_summary.py gpurun_out/<tag> [profiles/alu_roofline.json]"""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
PEAK = 256 * 4 * 16 * 2.4e9
res = {}
for pat in ("k_iou_clip", "k_giou_main"):
    dur = []
    for f in glob.glob(os.path.join(out, "trace_" + pat, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"] and "double" in row["Kernel_Name"]:
                dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    ctr = {}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES"):
        vals = []
        for f in glob.glob(os.path.join(out, "pmc_%s_%s" % (pat, c), "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if pat in row["Kernel_Name"] and "double" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    vals.append(float(row["Counter_Value"]))
        ctr[c] = sum(vals) / max(len(vals), 1)
    if not dur:
        continue
    d = sum(dur) / len(dur) * 1e-9
    cycles = ctr["GRBM_GUI_ACTIVE"] / 8.0
    busy = 4.0 * ctr["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles) if cycles else None
    ach = ctr["SQ_INSTS_VALU"] * 64.0 / d
    res[pat] = dict(duration_us=round(d * 1e6, 1), launches=len(dur), counters={k: round(v) for k, v in ctr.items()},
                    clock_GHz=round(cycles / d / 1e9, 3) if cycles else None, valu_busy=round(busy, 3) if busy else None,
                    achieved_Tlaneops=round(ach / 1e12, 2), peak_fp64_Tlaneops=round(PEAK / 1e12, 1), frac=round(ach / PEAK, 3),
                    valu_insts_per_wave=round(ctr["SQ_INSTS_VALU"] / max(ctr["SQ_WAVES"], 1)))
    print(pat, json.dumps(res[pat]))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
