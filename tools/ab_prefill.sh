cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --skip-cpu --skip-extra --skip-large 2>&1 | grep -v amdgpu | python -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']
print(b['value'], b['ms_per_step'], r['frac'], r['frac_net'], r['traffic'], r['traffic_source'][:90])"
