#!/usr/bin/env python3
"""tools/plan_bench.py [workload ...] -- the prepared band-major plan (mspmv_csrmv_plan_*) against the stateless call on
gather-bound matrices: set-up time, SpMV time per band count, per-kernel split, agreement of the results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import merge_spmv_amd as M
import sweep


def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


names = sys.argv[1:] or ["c2", "c2d", "rmat"]
band_sets = [int(b) for b in os.environ.get("PLAN_BANDS", "0,8,16,32").split(",")]
for label, A, x in sweep.workloads(names):
    vb = A.values.element_size()
    balg = A.nnz * (vb + 4) + (A.rows + 1) * 4 + A.rows * vb + A.cols * vb
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype); y0 = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y0, num_cols=A.cols, workspace=ws)
    # PLAN_SKIP_BASE=1 (profiling runs): only one stateless call, so the kernel statistics are the plan's
    base = float("nan") if os.environ.get("PLAN_SKIP_BASE") == "1" else \
        t(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y0, num_cols=A.cols, workspace=ws))
    print(f"== {label}: rows {A.rows} nnz {A.nnz} x {A.cols * vb / 1e6:.1f} MB | stateless {base:.4f} ms = {balg / base / 1e6:.0f} GB/s B_alg", flush=True)
    for bands in band_sets:
        try:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            plan = M.CsrMVPlan(A.values, A.row_offsets, A.column_indices, A.cols, bands=bands)
            torch.cuda.synchronize(); setup = (time.perf_counter() - t0) * 1e3
        except M.MspmvError as e:
            print(f"  bands {bands}: {e}"); continue
        y = torch.empty_like(y0)
        ms = t(lambda: plan(x, y))
        M.profile_begin(20)
        for _ in range(20): plan(x, y)
        torch.cuda.synchronize(); p = M.profile_end()
        err = float((y.double() - y0.double()).abs().max())
        print(f"  bands {plan.bands:2d}: setup {setup:8.2f} ms, {plan.bytes / 1e6:7.0f} MB | SpMV {ms:.4f} ms = {balg / ms / 1e6:6.0f} GB/s B_alg "
              f"({base / ms:.2f}x) | tile {p['tile_ms']:.4f} fix {p['fixup_ms']:.4f} | max diff {err:.2e}", flush=True)
        del plan
