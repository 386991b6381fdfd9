#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag> [commit]
# One collection = the bench line + rocprofv3 kernel stats of EXACTLY the three workloads the line's legs run -- config 2 alone
# (bench.py --skip-cpu --skip-extra --skip-large), the large frame (--large-only), the sparse operator (--sparse-only) -- + the PMC
# passes of each (FETCH_SIZE / WRITE_SIZE in SEPARATE runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes) + the SQ
# counters of k_bucket_index (LDS bank conflicts) -> gpurun_out/<tag>/ ;
# back in the build container: python tools/finish_collection.py <tag> <commit> copies the kernel stats to profiles/ and runs
# tools/summarize_pmc.py per workload -> profiles/<tag>_*_pmc_summary.csv + profiles/traffic.json (tag, commit, source hashes).
set -u
tag=${1:-r05}
commit=${2:-unknown}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
echo "$commit" > $out/commit.txt
python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg2 -o cfg2 -- python3 $B --skip-cpu --skip-extra --skip-large > $out/cfg2_prof.json 2> $out/prof_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_large -o large -- python3 $B --large-only > $out/large_prof.json 2> $out/prof_large.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_sparse -o sparse -- python3 $B --sparse-only --steps 50 --warmup 5 > $out/sparse_prof.json 2> $out/prof_sparse.err
# round 6: the box-side legs, each alone -- config 3's one 100 k x 100 k box2d_iou launch, config 3's NMS,
# config 4's iou3d -- kernel trace + FETCH_SIZE + WRITE_SIZE, so that their rooflines can be recomputed from profiles/ alone
# (--steps 5 --warmup 1: 11 operator calls per process for the NMS / iou3d legs, the figure tools/finish_collection.py divides by)
for leg in iou nms iou3d; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$leg -o $leg -- python3 $B --$leg-only --steps 5 --warmup 1 > $out/${leg}_prof.json 2> $out/prof_$leg.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_$leg -o pmc -- python3 $B --$leg-only --steps 5 --warmup 1 > /dev/null 2> $out/pmc_${c}_$leg.err
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_sparse -o pmc -- python3 $B --sparse-only --steps 5 --warmup 2 > /dev/null 2> $out/pmc_${c}_sparse.err
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_cfg2 -o pmc -- python3 $B --skip-cpu --skip-extra --skip-large --steps 5 --warmup 2 > /dev/null 2> $out/pmc_${c}_cfg2.err
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_large -o pmc -- python3 $B --large-only > /dev/null 2> $out/pmc_${c}_large.err
done
for c in SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/sq_$c -o pmc -- python3 $B --skip-cpu --skip-extra --skip-large --steps 5 --warmup 2 > /dev/null 2> $out/sq_$c.err
done
python3 - $out <<'PY' > $out/sq_bucket_index.txt
import csv, glob, os, sys
out = sys.argv[1]
vals = {}
for c in ("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
    for kern in ("k_bucket_index", "k_tile_sort", "k_emit", "k_emit_split"):
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(out, "sq_" + c, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kern + "<" in row["Kernel_Name"] or kern + "(" in row["Kernel_Name"]:
                    if row["Counter_Name"] == c:
                        tot += float(row["Counter_Value"]); n += 1
        vals[(kern, c)] = tot / max(n, 1)
        print("%-16s %-24s launches %4d  mean per launch %.0f" % (kern, c, n, tot / max(n, 1)))
for kern in ("k_bucket_index", "k_tile_sort", "k_emit", "k_emit_split"):
    a = vals.get((kern, "SQ_ACTIVE_INST_LDS"), 0)
    if a:
        print("%-16s SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = %.3f" % (kern, vals[(kern, "SQ_LDS_BANK_CONFLICT")] / a))
PY
cat $out/sq_bucket_index.txt
find $out -name "*stats*.csv" | head -20
