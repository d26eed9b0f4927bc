cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in old new; do
  if [ $v = new ]; then unset MSPMV_LIB; else export MSPMV_LIB=$GRAFT_REPO_ROOT/merge_spmv_amd/libmspmv_old.so; fi
  for dt in f32 f64; do
  python bench.py --no-configs --no-cpu-baseline --no-plan --dtype $dt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $dt', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done; done
unset MSPMV_LIB
python tools/band_passes_bench.py u8MB_f32 2>&1 | grep -v amdgpu | head -3
timeout 900 python -m pytest tests/test_band_passes.py -m gpu -x -q 2>&1 | tail -2
