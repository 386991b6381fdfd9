"""VGPRs / LDS / scratch / occupancy of the kernels of one .hip file, from hipcc's resource remarks.
usage: python tools/kernel_resources.py d3d_amd/csrc/voxel.hip [name-substring ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
want = sys.argv[2:]
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Iinclude", "-I../../include",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "")
        cur = dem.split("(")[0].replace("void ", "")
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    if want and not any(w in k for w in want):
        continue
    print("%-70s vgpr %4s agpr %3s sgpr %4s scratch %5s lds %6s occ %s" % (k[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"),
          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
