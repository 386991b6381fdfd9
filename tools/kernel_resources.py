"""Kernel resource table (VGPRs, SGPRs, occupancy, LDS, scratch) from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> remarks.txt ; python tools/kernel_resources.py remarks.txt [name filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2:]
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    s = m.group(1).strip()
    s = re.sub(r'^\S+:\d+:\d+:\s*', '', s)          # (remarks carry "file:line:col:" in front when the source path is absolute)
    if s.startswith('Function Name:'):
        cur = s.split(': ')[1]
        rows[cur] = {}
    elif cur and ':' in s:
        k, v = s.split(':', 1)
        rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    d = re.sub(r'\(anonymous namespace\)::', '', d)
    d = d.split('(')[0]
    if not filt or any(x in d for x in filt):
        print(d[:100].ljust(100), 'VGPR', v.get('VGPRs'), 'SGPR', v.get('TotalSGPRs'), 'occ', v.get('Occupancy [waves/SIMD]'),
              'lds', v.get('LDS Size [bytes/block]'), 'scratch', v.get('ScratchSize [bytes/lane]'))
