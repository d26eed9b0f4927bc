#!/usr/bin/env python3
"""tools/spmm_probe.py [workload] [k] [calls] -- a few SpMM calls and nothing else (what a counter pass is pointed at)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import merge_spmv_amd as M
import sweep
name = sys.argv[1] if len(sys.argv) > 1 else "dense32"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for label, A, x in sweep.workloads([name]):
    X = torch.empty(A.cols, k, dtype=A.values.dtype, device="cuda").uniform_(-1, 1)
    Y = torch.empty(A.rows, k, dtype=A.values.dtype, device="cuda")
    tmp = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(calls):
        M.csrmm(A.values, A.row_offsets, A.column_indices, X, Y=Y, temp=tmp)
    torch.cuda.synchronize()
    print(label, "k", k, "done")
    break
