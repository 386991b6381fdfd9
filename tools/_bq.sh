timeout 500 python tools/binned_check.py > gpurun_out/binned_check.log 2>&1; tail -3 gpurun_out/binned_check.log
python bench.py --steps 30 --warmup 5 --skip-cpu --skip-extra > gpurun_out/bench_binned.json 2>gpurun_out/bench_binned.err
python -c "
import json;d=json.loads(open('gpurun_out/bench_binned.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step']);print(d['kernels_us'])"
