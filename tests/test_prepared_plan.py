"""The opt-in prepared band-major plan (include/mspmv.h: mspmv_csrmv_plan_*): size query conventions on the CPU;
on the GPU, parity of plan SpMV with the oracle for every band count, sorted and unsorted rows, degenerate
shapes, alpha/beta, and bitwise reproducibility."""
import ctypes

import numpy as np
import pytest

import merge_spmv_amd as M
from oracle import oracle as O

torch = pytest.importorskip("torch")
gpu = pytest.mark.gpu


def test_plan_size_query_conventions():
    lib = M.load_library()
    size = ctypes.c_size_t(0); bands = ctypes.c_int32(0)
    # automatic band count: 1 while x fits an XCD's L2, then the fewest of 2, 4, 8, 16, ... with <= 3.25 MiB of x per band
    for cols, vb, want in ((1000, 4, 1), (700_000, 4, 1), (1_500_000, 4, 2), (3_125_000, 4, 4), (3_125_000, 8, 8), (16_000_000, 4, 24),
                           (200_000_000, 8, 64)):
        assert lib.mspmv_csrmv_plan_size(1000, cols, 5000, vb, 0, ctypes.byref(size), ctypes.byref(bands)) == 0
        assert bands.value == want, (cols, vb, bands.value)
    assert lib.mspmv_csrmv_plan_size(3_125_000, 3_125_000, 100_000_000, 4, 0, ctypes.byref(size), ctypes.byref(bands)) == 0
    # a second copy of the matrix + stacked offsets + partial sums + the CsrMV temp of the stacked problem
    assert bands.value == 4 and 800_000_000 + 2 * 4 * 4 * 3_125_000 <= size.value <= 1_000_000_000
    assert lib.mspmv_csrmv_plan_size(1000, 1000, 5000, 4, 8, ctypes.byref(size), None) == 0
    assert lib.mspmv_csrmv_plan_size(1000, 1000, 5000, 2, 8, ctypes.byref(size), None) == 1
    assert lib.mspmv_csrmv_plan_size(1000, 1000, 5000, 4, 65, ctypes.byref(size), None) == 1
    assert lib.mspmv_csrmv_plan_size(-1, 1000, 5000, 4, 8, ctypes.byref(size), None) == 1
    assert lib.mspmv_csrmv_plan_size(1000, 1000, 5000, 4, 8, None, None) == 1
    # bands * rows + nnz must fit the int32 single-call range
    assert lib.mspmv_csrmv_plan_size(60_000_000, 60_000_000, 100_000_000, 8, 64, ctypes.byref(size), None) == 1
    # too little storage / missing arrays are refused
    assert lib.mspmv_csrmv_plan_build_f32(ctypes.c_void_p(256), 16, None, None, None, 1000, 1000, 5000, 8, None, 0) == 1
    assert lib.mspmv_csrmv_plan_apply_f64(None, 1 << 30, None, None, 1000, 1000, 5000, 8, 1.0, 0.0, None, 0) == 1


def _random(rng, rows, cols, lens, dtype, sort_cols=True):
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    col = rng.integers(0, cols, nnz).astype(np.int32)
    if sort_cols:
        for r in range(rows):
            col[off[r]:off[r + 1]].sort()
    return O.Csr(rows, cols, off.astype(np.int32), col, rng.uniform(-1, 1, nnz).astype(dtype))


SHAPES = {
    "short_rows": lambda rng: (20000, 50000, rng.integers(0, 12, 20000)),
    "power_law": lambda rng: (8000, 30000, np.minimum((rng.pareto(1.1, 8000) * 2).astype(np.int64), 20000)),
    "giant_row": lambda rng: (3000, 100000, np.where(np.arange(3000) == 1500, 300000, rng.integers(0, 3, 3000))),
    "mostly_empty": lambda rng: (40000, 7000, np.where(np.arange(40000) % 97 == 0, 50, 0)),
    "all_empty": lambda rng: (500, 500, np.zeros(500, np.int64)),
    "single_col": lambda rng: (5000, 1, rng.integers(0, 3, 5000)),
    "tiny": lambda rng: (3, 5, np.array([2, 0, 1])),
}


@gpu
@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("bands", [0, 1, 2, 8, 24])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_plan_matches_oracle(shape, bands, prec):
    dtype, tdt = (np.float32, torch.float32) if prec == "f32" else (np.float64, torch.float64)
    rng = np.random.default_rng(len(shape) * 7 + bands)
    rows, cols, lens = SHAPES[shape](rng)
    csr = _random(rng, rows, cols, np.asarray(lens, np.int64), dtype)
    x = rng.uniform(-1, 1, cols).astype(dtype)
    d = lambda a: torch.from_numpy(a).cuda()
    plan = M.CsrMVPlan(d(csr.values), d(csr.row_offsets), d(csr.column_indices), cols, bands=bands)
    assert plan.bands == (bands or 1)
    y = plan(d(x))
    torch.cuda.synchronize()
    g, s = O.spmv_gold_acc64(csr, x)
    ok, worst = O.strict_check(csr, y.cpu().numpy(), g, s, items_per_thread=M.serial_sum_depth(csr.rows * plan.bands, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=plan.bands))
    assert ok, (shape, bands, prec, worst)
    # bitwise reproducible; alpha / beta; beta == 0 never reads y
    y2 = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
    plan(d(x), y2)
    assert torch.equal(y, y2)
    y0 = rng.uniform(-1, 1, rows).astype(dtype)
    y3 = plan(d(x), d(y0.copy()), alpha=-0.5, beta=3.0).cpu().numpy()
    want = -0.5 * g + 3.0 * y0.astype(np.float64)
    tol = (2.0 ** -19 if dtype == np.float32 else 2.0 ** -47) * (0.5 * s + 3.0 * np.abs(y0) + 1e-30)
    assert np.all(np.abs(y3 - want) <= tol)


@gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_plan_with_unsorted_rows(prec):
    """The API does not promise sorted columns (the reference's CSR has them, sparse_matrix.h:666-728): rows in
    arbitrary column order take the cursor path of the scatter pass."""
    dtype = np.float32 if prec == "f32" else np.float64
    rng = np.random.default_rng(4)
    csr = _random(rng, 9000, 40000, rng.integers(0, 30, 9000), dtype, sort_cols=False)
    x = rng.uniform(-1, 1, 40000).astype(dtype)
    d = lambda a: torch.from_numpy(a).cuda()
    plan = M.CsrMVPlan(d(csr.values), d(csr.row_offsets), d(csr.column_indices), 40000, bands=8)
    y = plan(d(x)).cpu().numpy()
    g, s = O.spmv_gold_acc64(csr, x)
    ok, worst = O.strict_check(csr, y, g, s, items_per_thread=M.serial_sum_depth(csr.rows * 8, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=8))
    assert ok, worst
    assert np.array_equal(plan(d(x)).cpu().numpy(), y)


@gpu
def test_plan_large_uniform_matrix_every_band_count():
    """a C2-shaped matrix at 1/10 size (x = 1.25 MB) with the band count forced: the large-problem tile kernel with
    the contiguous-range mapping, 313 k rows x 8..32 bands of stacked rows"""
    from merge_spmv_amd import generators as G
    A = G.uniform_csr(312_500, 312_500, 32, dtype=torch.float32)
    x = G.uniform_pm1(3, A.cols, torch.float32, "cuda")
    csr = O.Csr(A.rows, A.cols, A.row_offsets.cpu().numpy(), A.column_indices.cpu().numpy(), A.values.cpu().numpy())
    g, s = O.spmv_gold_acc64(csr, x.cpu().numpy())
    y_plain = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
    for bands in (8, 16, 32):
        plan = M.CsrMVPlan(A.values, A.row_offsets, A.column_indices, A.cols, bands=bands)
        y = plan(x)
        ok, worst = O.strict_check(csr, y.cpu().numpy(), g, s, items_per_thread=M.serial_sum_depth(csr.rows * bands, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=bands))
        assert ok, (bands, worst)
        assert float((y - y_plain).abs().max()) < 1e-4
