#!/usr/bin/env python3
"""tools/mg_bench.py -- the C multi-GPU operator (mspmv_mg_plan_*) with G parts on the visible device(s): time per
SpMV step with the carry exchange, time of the y -> x row all-gather, payload bytes of both.
With one GPU all parts share the device (peer exchange = events + one kernel; pushes are local copies): that measures
the operator's launch structure, not xGMI.  usage: python tools/mg_bench.py [workload] [parts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import multi_gpu as MG
import sweep

name = sys.argv[1] if len(sys.argv) > 1 else "grid2d"
parts_list = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
ndev = torch.cuda.device_count()
for label, A, x in sweep.workloads([name]):
    off = A.row_offsets.cpu().numpy().astype(np.int64)
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype).prepare(A.row_offsets); y0 = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    def one(): M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y0, num_cols=A.cols, workspace=ws)
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): one()
    torch.cuda.synchronize(); base = (time.perf_counter() - t0) / 50 * 1e3
    print(f"== {label}: rows {A.rows} nnz {A.nnz}; single-GPU prepared call {base:.4f} ms", flush=True)
    for parts in parts_list:
        row_split, nz_split = MG.partition(off, parts)
        devs = [g % ndev for g in range(parts)] if ndev >= parts else [0] * parts
        plan = MG.MgPlan(row_split, nz_split, A.cols, A.values.dtype, list(range(parts)), devs)
        for g in range(parts):
            lo = MG.local_offsets(off, row_split[g], row_split[g + 1], nz_split[g], nz_split[g + 1])
            a, b = int(nz_split[g]), int(nz_split[g + 1])
            d = torch.device("cuda", devs[g])
            plan.set_part(g, A.values[a:b].to(d).clone(), torch.from_numpy(lo).to(d), A.column_indices[a:b].to(d).clone())
            plan.x(g).copy_(x.to(d))
        torch.cuda.synchronize()
        def t(fn, iters=50):
            for _ in range(3): fn()
            plan.synchronize(); t0 = time.perf_counter()
            for _ in range(iters): fn()
            plan.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3
        spmv = t(plan.csrmv)
        info = plan.info()
        line = (f"  {parts} parts on {len(set(devs))} device(s), exchange {'peer reads' if info['exchange'] == 2 else 'RCCL'}: "
                f"SpMV step {spmv:.4f} ms (carries {info['carry_bytes_per_step']} B)")
        if A.rows == A.cols:
            both = t(lambda: (plan.csrmv(), plan.allgather_rows()))
            line += f"; SpMV + y->x all-gather {both:.4f} ms (+{both - spmv:.4f} ms, {info['allgather_bytes_per_step']} B over links, {A.rows * A.values.element_size()} B of rows)"
        print(line, flush=True)
        plan.close()
