#!/bin/bash
# round 4, first GPU call: device cache attributes, the self-launching bench test, band counts beyond 4
O=gpurun_out/r04_probe1; mkdir -p $O
python -c "import merge_spmv_amd as M; print(M.device_caches())" > $O/device_caches.txt 2>&1
timeout 900 python -m pytest tests/test_mg_plan.py -x -q -m gpu -k "starts_its_own_ranks" > $O/test_self_launch.txt 2>&1
BAND_FORCE=3,4,5,6,8 timeout 1500 python tools/band_passes_bench.py c2_f32 c2_f64 u16MB_f32 u24MB_f32 u32MB_f32 u16MB_f64 u24MB_f64 u32MB_f64 2>&1 | grep -v "^never vs\|^band_\|^rmat" > $O/band_passes_5to8.txt
tail -3 $O/test_self_launch.txt; cat $O/device_caches.txt; cat $O/band_passes_5to8.txt
