#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r2_tests_full3.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r2_bench_c2_final.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --dtype f64 --no-cpu-baseline > $O/r2_bench_c2_f64_final.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --workload dense32 --no-cpu-baseline > $O/r2_bench_dense32_final.txt 2>&1
( time timeout 1500 python tools/c3_ingest.py ) > $O/r2_c3_ingest2.txt 2>&1
cd merge_spmv_amd
( ./cpu_spmv --dense=5 --i=100 --pin; ./cpu_spmv --grid2d=4096 --i=30 --pin ) 2>&1 | grep -E "Merge|fp64|threads" > ../$O/r2_cpu_pin2.txt
