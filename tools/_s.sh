cd $GRAFT_REPO_ROOT
echo "# dense5_f64 grid2d_2000_f64 grid2d_4096_f64 grid3d_200_f64 dense32_f32 dense32_f64 band5_f32 web rmat22_f64 c4_f32"
for rep in 1 2 3; do
for v in old new; do
  if [ $v = new ]; then unset MSPMV_LIB; else export MSPMV_LIB=$GRAFT_REPO_ROOT/merge_spmv_amd/libmspmv_old.so; fi
  echo -n "$v (round $rep): "
  SWEEP_DEFAULT_SHAPE=1 SWEEP_NO_ROCSPARSE=1 timeout 600 python tools/sweep.py dense5d grid2d grid2d4096 grid3d dense32 dense32d band web rmat c4 2>&1 | grep -E "DEFAULT" | awk '{print $4}' | tr '\n' ' '; echo
done; done
unset MSPMV_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
