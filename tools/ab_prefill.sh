cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" clip3 clip5 clip6; do
  if [ -z "$v" ]; then L=""; else L=$GRAFT_REPO_ROOT/d3d_amd/libd3d_x_$v.so; fi
  echo "== ${v:-default}"; D3D_X_LIB=$L python tools/riou_bwd_ab.py 2>&1 | grep "5000x5000 forward"
done; done
