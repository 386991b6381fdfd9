#!/bin/bash
# SQ counters of one kernel: one rocprofv3 pass per counter (--kernel-trace + --pmc only), mean per launch.
# usage (GPU box): bash tools/sq_counters.sh <tag> <kernel pattern> <script.py> [script args] -- counters...
tag=$1; pat=$2; shift 2
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/${args[0]} "${args[@]:1}" > /dev/null 2> $out/pmc_$c.err
  python3 - "$out/pmc_$c" "$pat" "$c" <<'PY' >> $out/summary.txt
import csv, glob, os, sys
d, pat, c = sys.argv[1:4]
vals = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"] and row["Counter_Name"] == c:
            vals.append(float(row["Counter_Value"]))
print(c, len(vals), sum(vals) / max(len(vals), 1))
PY
done
cat $out/summary.txt
