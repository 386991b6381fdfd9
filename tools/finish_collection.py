#!/usr/bin/env python3
"""After `gpurun -- bash tools/collect_profiles.sh <tag> <commit>`: copy what the judge reads from gpurun_out/<tag>/ into profiles/
and summarise the PMC passes (tools/summarize_pmc.py per workload -> profiles/<tag>_<leg>_pmc_summary.csv, profiles/traffic.json).
usage: python tools/finish_collection.py <tag> [commit]"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else "unknown"
src = os.path.join(ROOT, "gpurun_out", tag)
prof = os.path.join(ROOT, "profiles")
LEGS = {  # leg -> (traffic.json workload, stats file name, operator calls per profiled process or 0)
    "cfg2": ("config2", "config2_only", 0), "large": ("config5_1gpu", "large_frame", 0), "sparse": ("config2_sparse", "sparse_only", 0),
    "iou": ("config3_iou", "config3_iou_only", 0), "nms": ("config3_nms", "config3_nms_only", 11),
    "iou3d": ("config4_iou3d", "config4_iou3d_only", 11),
}


def find(d, pat):
    got = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return got[0] if got else None


for leg, (workload, name, calls) in LEGS.items():
    st = find(os.path.join(src, "prof_" + leg), "*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(prof, "%s_kernel_stats_%s.csv" % (tag, name)))
    fe = find(os.path.join(src, "pmc_FETCH_SIZE_" + leg), "*counter_collection.csv")
    wr = find(os.path.join(src, "pmc_WRITE_SIZE_" + leg), "*counter_collection.csv")
    if fe and wr:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), fe, wr, "%s_%s" % (tag, leg), workload, commit]
        if calls:
            cmd.append(str(calls))
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
        print("summarised", leg, "->", workload)
    else:
        print("no PMC passes for", leg)
for f, dst in (("bench.json", "%s_bench.json" % tag), ("sq_bucket_index.txt", "%s_index_sq_pmc.txt" % tag)):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(prof, dst))
for leg in ("iou", "nms", "iou3d"):
    f = os.path.join(src, leg + "_prof.json")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(prof, "%s_%s_leg_under_rocprofv3.json" % (tag, leg)))
