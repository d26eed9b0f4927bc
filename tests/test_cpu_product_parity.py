"""The PRODUCT's OpenMP merge-path CsrMV (merge_spmv_amd/host/merge_csrmv.hpp, exported from
libmspmv_host.so; the kernel the cpu_spmv driver and bench.py's cpu_baseline time) against the oracle's
restatement of the reference's OmpMergeCsrmv (cpu_spmv.cpp:292-353): for an equal number of path
segments the two must agree BIT FOR BIT (same diagonals, same sequential sums, same carry fix-up
order) -- BASELINE.md 3.  CPU only."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import oracle as O

CASES = load_golden("matrices.json")["cases"]
IDS = [c["label"] for c in CASES]
SEGMENTS = [1, 2, 3, 4, 5, 6, 7, 8, 13, 64, 300]


@pytest.fixture(scope="module")
def H():
    lib = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
    vp, i = ctypes.c_void_p, ctypes.c_int
    for name in ("mspmv_host_merge_csrmv_f32", "mspmv_host_merge_csrmv_f64"):
        getattr(lib, name).argtypes = [i, i, i, i, vp, vp, vp, vp, vp]
        getattr(lib, name).restype = None
    for name in ("mspmv_host_merge_csrmv_bench_f32", "mspmv_host_merge_csrmv_bench_f64"):
        getattr(lib, name).argtypes = [i, i, i, i, i, vp, vp, vp, vp, ctypes.c_double, i, vp, vp, vp, vp, vp]
        getattr(lib, name).restype = i
    return lib


def product(H, csr, x, segments, threads=0):
    y = np.full(csr.rows, np.nan, dtype=csr.values.dtype)
    fn = H.mspmv_host_merge_csrmv_f32 if csr.values.dtype == np.float32 else H.mspmv_host_merge_csrmv_f64
    fn(segments, threads, csr.rows, csr.nnz, csr.row_offsets.ctypes.data, csr.column_indices.ctypes.data,
       csr.values.ctypes.data, x.ctypes.data, y.ctypes.data)
    return y


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32 if a.dtype == np.float32 else np.uint64),
                                                 b.view(np.uint32 if b.dtype == np.float32 else np.uint64))


def build(case, dtype):
    args = list(case["args"])
    if case["kind"] == "mtx":
        args = [os.path.join(ROOT, args[0])]
    return O.make(case["kind"], *args, dtype=dtype)


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_product_merge_csrmv_is_bitwise_the_oracle_on_the_golden_matrices(H, case, prec):
    dtype = np.float32 if prec == "f32" else np.float64
    csr = build(case, dtype)
    rng = np.random.default_rng(len(case["label"]))
    for x in (np.ones(csr.cols, dtype), rng.uniform(-1, 1, csr.cols).astype(dtype)):
        for s in SEGMENTS:
            want = O.omp_merge_csrmv(csr, x, s)
            got = product(H, csr, x, s)
            assert same_bits(got, want), (case["label"], prec, s)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_product_merge_csrmv_random_skewed_matrix(H, prec):
    """power-law row lengths with empty rows, a giant row and random values: every segment count,
    and a thread team smaller than the segment count gives the same bits (the result depends on the
    segments only)."""
    dtype = np.float32 if prec == "f32" else np.float64
    rng = np.random.default_rng(11)
    rows, cols = 5000, 3000
    lens = np.minimum((rng.pareto(1.1, rows) * 2).astype(np.int64), 4000)
    lens[rng.integers(0, rows, 500)] = 0
    lens[rows // 3] = 20000
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    csr = O.Csr(rows, cols, off.astype(np.int32), rng.integers(0, cols, nnz).astype(np.int32),
                rng.uniform(-1, 1, nnz).astype(dtype))
    x = rng.uniform(-1, 1, cols).astype(dtype)
    g, sabs = O.spmv_gold_acc64(csr, x)
    for s in SEGMENTS:
        want = O.omp_merge_csrmv(csr, x, s)
        got = product(H, csr, x, s)
        assert same_bits(got, want), (prec, s)
        assert same_bits(product(H, csr, x, s, threads=2), want), (prec, s, "team of 2")
        ok, worst = O.strict_check(csr, got, g, sabs, items_per_thread=8)
        assert ok, (prec, s, worst)


def test_timed_kernel_is_the_same_kernel(H):
    """mspmv_host_merge_csrmv_bench_* (what bench.py's cpu_baseline and the C1 run time: private
    first-touched copies, optional pinning) returns the y of the kernel above for segments == threads."""
    csr = O.make("grid2d", 60, dtype=np.float64)
    x = np.random.default_rng(2).uniform(-1, 1, csr.cols)
    for threads in (1, 3):
        y = np.zeros(csr.rows); avg = ctypes.c_double(); it = ctypes.c_int(); pinned = ctypes.c_int(); pk = ctypes.c_int()
        st = H.mspmv_host_merge_csrmv_bench_f64(threads, 0, csr.rows, csr.cols, csr.nnz, csr.row_offsets.ctypes.data,
                                                csr.column_indices.ctypes.data, csr.values.ctypes.data, x.ctypes.data, 0.05, 5,
                                                ctypes.byref(avg), ctypes.byref(it), ctypes.byref(pinned), ctypes.byref(pk),
                                                y.ctypes.data)
        assert st == 0 and 1 <= it.value <= 5 and avg.value > 0
        assert same_bits(y, O.omp_merge_csrmv(csr, x, threads))
