cd $GRAFT_REPO_ROOT
python tests/nccl_worldN_child.py 0 1 29577 100000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu" | tail -8
