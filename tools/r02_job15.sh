#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r2_tests_full7.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/r2_smoke.txt 2>&1
