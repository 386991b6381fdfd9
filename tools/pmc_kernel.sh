#!/bin/bash
# per-kernel counter averages of one command: tools/pmc_kernel.sh <tag> <kernel substring> <counter> [<counter> ...] -- python3 <script> [args]
# (one rocprofv3 pass per counter: --kernel-trace + --pmc only, as gpurun requires; run on the GPU box)
tag=$1; shift
pat=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "${ctrs[@]}"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- "$@" > /dev/null 2> $out/pmc_$c.err
  f=$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$pat" "$c" <<'PY'
import csv, sys
f, pat, c = sys.argv[1:4]
tot, n = 0.0, 0
for row in csv.DictReader(open(f)):
    if pat in row["Kernel_Name"] and row["Counter_Name"] == c:
        tot += float(row["Counter_Value"]); n += 1
print("%-28s %-24s launches %4d  mean %.1f" % (pat, c, n, tot / max(n, 1)))
PY
done
