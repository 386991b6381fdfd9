#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes) into per-kernel HBM bytes per launch.

    python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [workload] [commit] [calls]

`calls` (operators made of many launches: config3_nms, config4_iou3d): the operator calls the profiled process made; the
workload then also gets "<op>_op_total" = all its kernels' bytes / calls.

Corrections applied (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced read stream, so fetch is doubled (an upper bound for narrow/random accesses, which the
guide calls uncalibrated); WRITE_SIZE is taken as is.  Writes profiles/<tag>_pmc_summary.csv and merges
{workload: {kernel: bytes_per_launch}} into profiles/traffic.json (read by bench.py), together with the hashes of the kernel
sources the figures were collected on ("_sources": bench.py withholds a workload's figures once one of its files has changed).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    if not m:
        return name[:60]
    base, targs = m.group(1), m.group(2) or ""
    if base.startswith("k_scan_"):
        f = re.search(r"(NumberVoxels|FilterVoxels|FilterPoints|PopcountWords|MergeRecords|FirstWords)", targs)
        return "%s<%s>" % (base, f.group(1)) if f else base
    return "k_fill_c4" if base == "k_fill_c4_rows" else base       # (one kernel of the op, two forms: bench.py's name)


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    return acc


def main():
    fetch_csv, write_csv, tag = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "config2"
    commit = sys.argv[5] if len(sys.argv) > 5 else None
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    rows, traffic = [], {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("k_"):
            continue
        nf, f = fe.get(k, [0, 0.0])
        nw, w = wr.get(k, [0, 0.0])
        fb = 2.0 * 1024.0 * f / max(nf, 1)      # KiB -> B, x2 (gfx950 FETCH_SIZE correction)
        wb = 1024.0 * w / max(nw, 1)
        rows.append((k, nf, f / max(nf, 1), w / max(nw, 1), fb, wb, fb + wb))
        traffic[k] = int(fb + wb)
    out = os.path.join(ROOT, "profiles", "%s_pmc_summary.csv" % tag)
    with open(out, "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KiB_raw_avg,WRITE_SIZE_KiB_avg,fetch_bytes_corrected_x2,write_bytes,hbm_bytes_per_launch\n")
        for r in sorted(rows, key=lambda r: -r[6]):
            f.write("%s,%d,%.1f,%.1f,%.0f,%.0f,%.0f\n" % r)
    if workload.endswith("_sparse"):       # every kernel of the operator runs once per step: the operator's traffic per step
        traffic["sparse_op_total"] = int(sum(v for k, v in traffic.items()))
    calls = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    if calls > 0:                          # an operator of many launches: all bytes of the process / its calls
        tot = sum(2.0 * 1024.0 * fe.get(k, [0, 0.0])[1] + 1024.0 * wr.get(k, [0, 0.0])[1] for k in set(fe) | set(wr)
                  if k.startswith("k_") and not k.startswith("k_probe"))
        traffic[workload.split("_", 1)[1] + "_op_total"] = int(tot / calls)
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    allt = json.load(open(tj)) if os.path.exists(tj) else {}
    allt[workload] = traffic
    prof = allt.get("_profile")
    if not isinstance(prof, dict):
        prof = {}
    prof[workload] = tag if not commit else "%s @ commit %s" % (tag, commit)
    allt["_profile"] = prof
    sys.path.insert(0, ROOT)
    import bench
    allt.setdefault("_sources", {})[workload] = bench.source_hashes()
    json.dump(allt, open(tj, "w"), indent=1, sort_keys=True)
    print(open(out).read())


if __name__ == "__main__":
    main()
