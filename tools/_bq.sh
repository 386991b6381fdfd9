python bench.py --steps 40 --warmup 5 --skip-cpu --skip-extra > gpurun_out/bench_binned.json 2>gpurun_out/bench_binned.err
python -c "
import json;d=json.loads(open('gpurun_out/bench_binned.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step']);print(d['kernels_us'])"
