"""Worker of tests/test_mg_plan.py::test_hot_columns_are_automatic (a process of its own: MSPMV_FAKE_INFINITY_CACHE_MIB is read once).
A plan of 3 parts on cuda:0 over (a) a scale-free matrix, (b) uniformly spread columns, (c) a band: the automatic hot-column decision
(the plan's default: x beyond the -- here faked, 1 MiB -- Infinity Cache AND columns that come back) must pick (a) only; y must be bit
for bit the same with the plans, without them, and forced on; mspmv_mg_plan_exchange_ms must answer after a step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import multi_gpu as MG, generators as G

assert os.environ.get("MSPMV_FAKE_INFINITY_CACHE_MIB") == "1"
M.use_library("dev")           # (the environment override exists in the development library only)
lib = M.load_library()
dev = torch.device("cuda", 0)
parts = 3


def build(A):
    off = A.row_offsets.cpu().numpy().astype(np.int64)
    row_split, nz_split = MG.partition(off, parts)
    plan = MG.MgPlan(row_split, nz_split, A.cols, torch.float64, list(range(parts)), [0] * parts, exchange=MG.EXCHANGE_PEER)
    keep = []
    for g in range(parts):
        lo = torch.from_numpy(MG.local_offsets(off, row_split[g], row_split[g + 1], nz_split[g], nz_split[g + 1])).cuda()
        a, b = int(nz_split[g]), int(nz_split[g + 1])
        v, c = A.values[a:b].clone(), A.column_indices[a:b].clone()
        keep.append((v, lo, c))
        plan.set_part(g, v, lo, c)
    return plan, row_split, keep


def y_of(plan, row_split):
    plan.csrmv(); plan.synchronize()
    return torch.cat([plan.y(g)[: int(row_split[g + 1] - row_split[g])] for g in range(parts)]).clone()


n = 1 << 18                                                      # x = 2 MiB (uniform) ... 16 MiB (scale-free) > the faked 1 MiB cache
cases = {
    "scale_free": G.rmat_csr(21, 12_000_000, dtype=torch.float64, device=dev, seed=G.SEED_C5),      # (x = 16 MiB: lines that keep coming back)
    "uniform": G.uniform_csr(n, n, 16, dtype=torch.float64, device=dev),
    "band": G.grid2d_csr(512, torch.float64, device=dev),
}
for name, A in cases.items():
    median, wide = ctypes.c_int32(), ctypes.c_int32()
    assert lib.mspmv_csrmv_hotcols_skew(ctypes.c_void_p(A.column_indices.data_ptr()), A.cols, A.nnz, 8, None, ctypes.byref(median), ctypes.byref(wide)) == 0
    plan, row_split, keep = build(A)
    try:
        x = G.uniform_pm1(7, A.cols, torch.float64, dev)
        plan.x(0).copy_(x); torch.cuda.synchronize()
        chosen = plan.info()["hot_parts"]
        print(f"{name}: distinct lines {median.value} per mille of a uniform draw, wide windows {wide.value} / 512, parts with the plan {chosen}", flush=True)
        assert chosen == (parts if name == "scale_free" else 0), (name, chosen, median.value, wide.value)
        y_auto = y_of(plan, row_split)
        assert plan.exchange_ms(0) >= 0.0
        plan.hot_columns(False); assert plan.info()["hot_parts"] == 0
        y_off = y_of(plan, row_split)
        plan.hot_columns(True); assert plan.info()["hot_parts"] == parts
        y_on = y_of(plan, row_split)
        plan.hot_columns(-1); assert plan.info()["hot_parts"] == chosen
        assert torch.equal(y_auto, y_off) and torch.equal(y_on, y_off)
        # against the plain call
        y_ref = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
        torch.cuda.synchronize()
        bound = 1e-9 * float((A.values.abs().max() * x.abs().max()).item()) * max(1, int((A.row_offsets[1:] - A.row_offsets[:-1]).max().item()))
        assert float((y_ref - y_off).abs().max().item()) <= bound
    finally:
        plan.close()
print("AUTO-HOT-OK")
