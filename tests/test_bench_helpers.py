"""bench.py's pieces that need no GPU: the Matrix Market path behind --mtx-dir (the product's ingest through libmspmv_host.so, the stand-in
marker of tools/make_standin_mtx.py), the general .mtx writer (mspmv_host_write_mtx), and the untimed correctness witness (sampled_check)."""
import ctypes
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import merge_spmv_amd as M  # noqa: E402
from merge_spmv_amd import generators as G  # noqa: E402


def _writer():
    H = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
    H.mspmv_host_write_mtx.restype = ctypes.c_int
    H.mspmv_host_write_mtx.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_char_p]
    return H


@pytest.mark.parametrize("values,symmetric,marked", [(True, False, True), (False, True, True), (False, False, False)])
def test_mtx_dir_path_reads_what_the_writer_wrote(tmp_path, values, symmetric, marked):
    """A file written by mspmv_host_write_mtx (real general / pattern symmetric / pattern general; with and without the stand-in marker)
    comes back through bench.load_mtx -- CooMatrix::InitMarket + CsrMatrix::Init of the product -- as the CSR of the same entries
    (mirrored when symmetric), and the marker decides how bench.py labels the record's `data`."""
    rng = np.random.default_rng(3 + values + 2 * symmetric)
    n, cnt = 300, 2000
    r = rng.integers(0, n, cnt).astype(np.int32); c = rng.integers(0, n, cnt).astype(np.int32)
    if symmetric:
        keep = r >= c; r, c = r[keep], c[keep]
    v = rng.uniform(-1, 1, r.size) if values else None
    path = str(tmp_path / "m.mtx")
    H = _writer()
    comment = (bench.STANDIN_MARK + ": a test matrix").encode() if marked else None
    assert H.mspmv_host_write_mtx(path.encode(), n, n, r.size, r.ctypes.data, c.ctypes.data, None if v is None else v.ctypes.data, int(symmetric), comment) == 0
    A, stand_in = bench.load_mtx(torch, G, path, torch.float64, "cpu")
    assert stand_in == marked and (A.rows, A.cols) == (n, n)
    rr, cc, vv = r.astype(np.int64), c.astype(np.int64), (v if values else np.ones(r.size))
    if symmetric:
        off = rr != cc
        rr, cc, vv = np.concatenate([rr, cc[off]]), np.concatenate([cc, rr[off] if False else r.astype(np.int64)[off]]), np.concatenate([vv, vv[off]])
    assert A.nnz == rr.size
    dense = np.zeros((n, n)); np.add.at(dense, (rr, cc), vv)
    offs = A.row_offsets.numpy(); cols = A.column_indices.numpy(); vals = A.values.numpy()
    got = np.zeros((n, n)); np.add.at(got, (np.repeat(np.arange(n), np.diff(offs)), cols), vals)
    assert np.allclose(got, dense, rtol=0, atol=1e-12)
    assert all(np.all(np.diff(cols[offs[i]: offs[i + 1]]) >= 0) for i in range(n))          # sorted by (row, column) like CsrMatrix::Init


def test_sampled_check_passes_a_right_result_and_catches_wrong_ones():
    """merge_spmv_amd.sampled_check on the CPU (it only needs torch tensors and a depth): right y passes well under 1; a wrong long row, a
    nonzero empty row and a NaN are violations -- the first, last and longest rows are always among the rows looked at."""
    rng = np.random.default_rng(9)
    rows, cols = 5000, 700
    lens = rng.integers(0, 12, rows); lens[0] = 0; lens[rows - 1] = 3; lens[1234] = 9000
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:]); nnz = int(off[-1])
    col = rng.integers(0, cols, nnz).astype(np.int32); val = rng.uniform(-1, 1, nnz).astype(np.float32); x = rng.uniform(-1, 1, cols).astype(np.float32)
    y = np.zeros(rows, np.float32)
    prod = val.astype(np.float64) * x.astype(np.float64)[col]
    np.add.at(y, np.repeat(np.arange(rows), lens), 0)       # (shape only)
    y[:] = np.add.reduceat(np.concatenate([prod, [0.0]]), np.minimum(off[:-1], nnz)).astype(np.float32) * (lens > 0)
    A = SimpleNamespace(rows=rows, cols=cols, nnz=nnz, row_offsets=torch.from_numpy(off.astype(np.int32)), column_indices=torch.from_numpy(col),
                        values=torch.from_numpy(val))
    ok = M.sampled_check(A, torch.from_numpy(x), torch.from_numpy(y), depth=12)
    assert ok["violations"] == 0 and ok["worst_ratio"] < 1 and ok["longest_row"] == 9000 and ok["rows_checked"] > 3000
    for row, bad in ((1234, 1.0), (0, 1e-20), (rows - 1, float("nan"))):
        y2 = y.copy(); y2[row] = y2[row] + bad if bad == bad else bad
        res = M.sampled_check(A, torch.from_numpy(x), torch.from_numpy(y2), depth=12)
        assert res["violations"] >= 1, (row, res)
