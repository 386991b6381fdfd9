#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag>
# bench line, rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only)
# -> gpurun_out/<tag>/ ; copy what should be judged into profiles/ afterwards (tools/summarize_pmc.py for the PMC passes)
set -u
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --skip-cpu > $out/bench_prof.json 2> $out/prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --skip-cpu --skip-extra --steps 5 --warmup 2 > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --skip-cpu --skip-extra --steps 5 --warmup 2 > /dev/null 2> $out/pmc_write.err
ls $out $out/pmc_fetch $out/pmc_write
