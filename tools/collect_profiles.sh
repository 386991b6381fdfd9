#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag>
# bench line, rocprofv3 kernel stats and the PMC passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs with --kernel-trace only,
# as MI355X_MICROARCH.md prescribes) for config 2 and for config 5's frame on one GPU
# -> gpurun_out/<tag>/ ; tools/summarize_pmc.py turns the PMC passes into profiles/<tag>_pmc_*.csv + profiles/traffic.json
set -u
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --skip-cpu --skip-large > $out/bench_prof.json 2> $out/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_large -o large -- python3 $GRAFT_REPO_ROOT/bench.py --large-only > $out/large_prof.json 2> $out/prof_large.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_sparse -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --sparse-only --steps 5 --warmup 2 > /dev/null 2> $out/pmc_${c}_sparse.err
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_cfg2 -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --skip-cpu --skip-extra --skip-large --steps 5 --warmup 2 > /dev/null 2> $out/pmc_${c}_cfg2.err
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_large -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --large-only > /dev/null 2> $out/pmc_${c}_large.err
done
find $out -name "*.csv" | head -40
