"""The opt-in hot-column plan (include/mspmv.h: mspmv_csrmv_hotcols_*): size / argument conventions on the CPU; on the GPU
the plan's SpMV must be BIT FOR BIT the stateless call's (a column permutation only changes where x is read), the stored
permutation must be a bijection that puts more-referenced columns first (by class), on every shape family and for
alpha / beta, and a scale-free matrix with an x beyond the caches must actually get faster."""
import ctypes

import numpy as np
import pytest

import merge_spmv_amd as M
from oracle import oracle as O

torch = pytest.importorskip("torch")
gpu = pytest.mark.gpu


def test_hotcols_size_and_argument_conventions():
    lib = M.load_library()
    size = ctypes.c_size_t(0)
    assert lib.mspmv_csrmv_hotcols_size(1000, 2000, 5000, 8, ctypes.byref(size)) == 0
    # renumbered indices + permutation + permuted x + counts + the inner call's temp storage; no copy of values / offsets
    assert size.value >= 5000 * 4 + 2000 * 4 + 2000 * 8 + 2000 * 4
    rows, cols, nnz = 1 << 26, 1 << 26, 2_000_000_000
    assert lib.mspmv_csrmv_hotcols_size(rows, cols, nnz, 8, ctypes.byref(size)) == 0
    assert 4 * nnz + (4 + 8 + 4) * cols <= size.value <= 4 * nnz + 16 * cols + (1 << 27)      # 8 GB of indices + ~1.1 GB: config 5
    assert lib.mspmv_csrmv_hotcols_size(1000, 2000, 5000, 2, ctypes.byref(size)) == 1
    assert lib.mspmv_csrmv_hotcols_size(-1, 2000, 5000, 4, ctypes.byref(size)) == 1
    assert lib.mspmv_csrmv_hotcols_size(1000, 2000, 5000, 4, None) == 1
    assert lib.mspmv_csrmv_hotcols_size(1 << 30, 5, (1 << 30) + 5, 4, ctypes.byref(size)) == 1          # rows + nnz beyond the int32 path range
    assert lib.mspmv_csrmv_hotcols_build(ctypes.c_void_p(256), 16, None, None, 1000, 2000, 5000, 4, None, 0) == 1
    assert lib.mspmv_csrmv_hotcols_apply_f64(None, 1 << 30, None, None, None, None, 1000, 2000, 5000, 1.0, 0.0, None, 0) == 1
    assert lib.mspmv_csrmv_hotcols_order(None, 1000, 2000, 5000, 4) is None


def _random(rng, rows, cols, lens, dtype, skew):
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    if skew:          # scale-free column references: most of them go to few columns, scattered over [0, cols)
        hot = rng.permutation(cols)
        col = hot[np.minimum((rng.pareto(0.9, nnz) * 3).astype(np.int64), cols - 1)].astype(np.int32)
    else:
        col = rng.integers(0, cols, nnz).astype(np.int32)
    return O.Csr(rows, cols, off.astype(np.int32), col, rng.uniform(-1, 1, nnz).astype(dtype))


SHAPES = {
    "short_rows": lambda rng: (20000, 50000, rng.integers(0, 12, 20000)),
    "power_law": lambda rng: (8000, 30000, np.minimum((rng.pareto(1.1, 8000) * 2).astype(np.int64), 20000)),
    "giant_row": lambda rng: (3000, 100000, np.where(np.arange(3000) == 1500, 300000, rng.integers(0, 3, 3000))),
    "mostly_empty": lambda rng: (40000, 7000, np.where(np.arange(40000) % 97 == 0, 50, 0)),
    "all_empty": lambda rng: (500, 500, np.zeros(500, np.int64)),
    "single_col": lambda rng: (5000, 1, rng.integers(0, 3, 5000)),
    "more_cols_than_references": lambda rng: (300, 1_000_000, rng.integers(0, 9, 300)),
    "tiny": lambda rng: (3, 5, np.array([2, 0, 1])),
}


@gpu
@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("skew", [False, True])
def test_hotcols_spmv_is_bitwise_the_stateless_result(shape, prec, skew):
    dtype = np.float32 if prec == "f32" else np.float64
    rng = np.random.default_rng(sum(map(ord, shape)) + skew)
    rows, cols, lens = SHAPES[shape](rng)
    csr = _random(rng, rows, cols, np.asarray(lens, np.int64), dtype, skew)
    x = rng.uniform(-1, 1, cols).astype(dtype)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    val, off, col, xd = d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x)
    y_ref = M.csrmv(val, off, col, xd, num_cols=cols)
    plan = M.CsrMVHotColumns(val, off, col, cols)
    y = plan(xd)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    # the permutation: a bijection, consistent with the renumbered indices, hotter classes first
    order = plan.order().cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(order), np.arange(cols))
    new_cols = plan.columns().cpu().numpy().astype(np.int64)
    assert np.array_equal(order[new_cols], csr.column_indices.astype(np.int64))
    counts = np.bincount(csr.column_indices, minlength=cols)
    cls = np.floor(np.log2(counts[order] + 1.0)).astype(np.int64)
    assert np.all(np.diff(cls) <= 0)
    # alpha / beta
    y0 = rng.uniform(-1, 1, rows).astype(dtype)
    a = M.csrmv(val, off, col, xd, y=d(y0.copy()), alpha=-0.5, beta=2.0, num_cols=cols)
    b = plan(xd, d(y0.copy()), alpha=-0.5, beta=2.0)
    assert torch.equal(a, b)
    # and again (the plan's hints and x buffer are reused)
    assert torch.equal(plan(xd), y_ref)
    # a caller that keeps x in the plan's numbering (mspmv_csrmv_hotcols_permute_* once, _apply_permuted_* per SpMV: no per-call pass over x):
    # the permuted vector is x[order], y comes out in the original row order, bit for bit the same -- with alpha / beta too
    xp = plan.permute(xd)
    assert torch.equal(xp, xd[plan.order().long()]) if cols else xp.numel() == 0
    assert torch.equal(plan(xp, x_is_permuted=True), y_ref)
    assert torch.equal(plan(xp, d(y0.copy()), alpha=-0.5, beta=2.0, x_is_permuted=True), a)
    with pytest.raises(M.MspmvError):
        plan.permute(xd, out=xd)                                  # not in place


@gpu
def test_hotcols_pays_on_a_scale_free_matrix_with_a_large_x():
    """R-MAT scale 24 with 200 M edges (x = 134 MB fp64, beyond an XCD's L2 and half the Infinity Cache): the plan's SpMV,
    x permutation included, must beat the stateless call clearly (config 5 at full size: 34.2 -> 20.6 ms, tools/hot_columns.py),
    and every row stays within the strict bound of the oracle (it is the same result bit for bit)."""
    import time
    from merge_spmv_amd import generators as G
    A = G.rmat_csr(24, 200_000_000, dtype=torch.float64, device="cuda", seed=G.SEED_C5)
    x = G.uniform_pm1(G.SEED_C5 + 2, A.cols, torch.float64, "cuda")
    ws = M.CsrMVWorkspace(A.rows, A.nnz, torch.float64)
    y = torch.empty(A.rows, dtype=torch.float64, device="cuda")
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    plan = M.CsrMVHotColumns(A.values, A.row_offsets, A.column_indices, A.cols)
    yp = torch.empty_like(y)

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    t_plain, t_plan = timed(call), timed(lambda: plan(x, yp))
    assert torch.equal(y, yp)
    assert t_plan < 0.85 * t_plain, (t_plain, t_plan)


@gpu
def test_plan_in_the_band_policy_size_range_equals_the_one_sweep_call():
    """Advisor (r03, low): for x of 5.5-40 MiB under spread columns the STATELESS call may take the column-band passes (a
    re-association), and their verdict is sampled from the column indices, which the plan renumbers.  The plan never takes
    them: it is bit for bit the stateless call with the passes off, and within the bound of the automatic call."""
    from merge_spmv_amd import generators as G
    A = G.uniform_csr(2_000_000, 2_000_000, 12, dtype=torch.float32)          # x = 7.6 MiB, 24 M nonzeros = 192 MiB of stream
    assert M.band_passes(A.rows, A.cols, A.nnz, 4) >= 2                       # (the stateless call is a candidate)
    x = G.uniform_pm1(9, A.cols, torch.float32, "cuda")
    plan = M.CsrMVHotColumns(A.values, A.row_offsets, A.column_indices, A.cols)
    y_plan = plan(x).clone()
    y_auto = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols).clone()
    try:
        M.set_band_passes(4, -1)
        y_one = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols).clone()
    finally:
        M.set_band_passes(4, 0)
    torch.cuda.synchronize()
    assert torch.equal(y_plan, y_one)
    lens = (A.row_offsets[1:] - A.row_offsets[:-1]).double()
    prod = A.values.double() * x.double()[A.column_indices.long()]
    s = torch.segment_reduce(prod.abs(), "sum", lengths=(A.row_offsets[1:] - A.row_offsets[:-1]).long(), unsafe=True)
    tol = 2.0 * (torch.ceil(torch.log2(lens + 1)) + M.serial_sum_depth(A.rows, A.cols, A.nnz, 4) + 8) * 2.0 ** -24 * s
    assert bool(((y_auto.double() - y_plan.double()).abs() <= 2 * tol).all())
