"""The synthetic workloads of SURVEY.md 8(d) (merge_spmv_amd/generators.py, torch ops so
they can be built in HBM) against an independent numpy splitmix64: same integers, same
values, valid CSR.  CPU only."""
import numpy as np
import pytest
import torch

from merge_spmv_amd import generators as G

M64 = (1 << 64) - 1


def splitmix64_np(seed, idx):
    z = (np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def u01_np(seed, idx):
    return (splitmix64_np(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def test_splitmix_and_uniform_streams():
    idx = np.arange(0, 5000, dtype=np.int64)
    with np.errstate(over="ignore"):
        want = splitmix64_np(G.SEED_C2, idx)
    got = G.splitmix64(G.SEED_C2, torch.from_numpy(idx)).numpy().view(np.uint64)
    assert np.array_equal(got, want)
    with np.errstate(over="ignore"):
        u = u01_np(G.SEED_C2 + 1, idx)
    v = G.uniform_pm1(G.SEED_C2 + 1, 5000, torch.float32, "cpu").numpy()
    assert np.array_equal(v, (u * 2.0 - 1.0).astype(np.float32))
    assert v.min() >= -1.0 and v.max() < 1.0


def test_uniform_csr_is_the_c2_definition():
    rows, cols, npr = 300, 1000, 32
    A = G.uniform_csr(rows, cols, npr, dtype=torch.float32, device="cpu")
    assert A.row_offsets.tolist() == [r * npr for r in range(rows + 1)]
    with np.errstate(over="ignore"):
        c = np.minimum((u01_np(G.SEED_C2, np.arange(rows * npr)) * cols).astype(np.int64), cols - 1).reshape(rows, npr)
    c.sort(axis=1)
    assert np.array_equal(A.column_indices.numpy().reshape(rows, npr), c)
    # a row range of the same matrix
    B = G.uniform_csr(rows, cols, npr, dtype=torch.float32, device="cpu", row_lo=100, row_hi=180)
    assert torch.equal(B.column_indices, A.column_indices[100 * npr: 180 * npr])
    assert torch.equal(B.values, A.values[100 * npr: 180 * npr])


def test_degenerate_csr_shape():
    A = G.degenerate_csr(rows=1 << 12, giant_nnz=1 << 14, every=64, dtype=torch.float32, device="cpu")
    lens = np.diff(A.row_offsets.numpy())
    assert lens[2048] == 1 << 14 and lens.sum() == A.nnz
    others = np.delete(lens, 2048)
    assert set(np.unique(others)) <= {0, 1} and others.sum() == (1 << 12) // 64 - 1
    cols = A.column_indices.numpy()
    g0 = A.row_offsets[2048].item()
    assert np.all(np.diff(cols[g0: g0 + (1 << 14)]) >= 0) and cols.max() < (1 << 12)


def test_rmat_csr_valid_and_range_consistent():
    scale, edges = 10, 20000
    A, e = G.rmat_csr(scale, edges, dtype=torch.float64, device="cpu", return_edge_ids=True)
    off = A.row_offsets.numpy().astype(np.int64); col = A.column_indices.numpy()
    assert off[0] == 0 and off[-1] == edges and np.all(np.diff(off) >= 0)
    rowid = np.repeat(np.arange(1 << scale), np.diff(off))
    key = rowid * (1 << scale) + col
    assert np.all(np.diff(key) >= 0)                       # sorted by (row, col), duplicates kept
    assert len(np.unique(e.numpy())) == edges
    lens = np.diff(off)
    assert lens.max() > 20 * lens.mean()                   # skewed, as R-MAT should be
    B = G.rmat_csr(scale, edges, dtype=torch.float64, device="cpu", row_lo=100, row_hi=400)
    a, b = off[100], off[400]
    assert torch.equal(B.column_indices, A.column_indices[a:b]) and torch.equal(B.values, A.values[a:b])


def test_tools_and_entry_points_compile():
    """the development scripts under tools/, bench.py and __graft_entry__.py at least parse"""
    import glob, os, py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]:
        py_compile.compile(f, doraise=True)


@pytest.mark.parametrize("kind,width", [("grid2d", 1), ("grid2d", 2), ("grid2d", 9), ("grid2d", 40), ("grid3d", 1), ("grid3d", 2), ("grid3d", 7)])
def test_lattice_generators_are_the_reference_inputs(kind, width):
    """grid2d_csr / grid3d_csr (what the measuring tools feed the GPU) are array for array the CSR the oracle's restatement of
    InitGrid2d / InitGrid3d + the COO -> CSR conversion gives (sparse_matrix.h:461-617,666-728)."""
    from oracle import oracle as O
    c = O.make(kind, width, dtype=np.float64)
    d = (G.grid2d_csr if kind == "grid2d" else G.grid3d_csr)(width, torch.float64, "cpu")
    assert (c.rows, c.cols) == (d.rows, d.cols)
    assert np.array_equal(c.row_offsets, d.row_offsets.numpy())
    assert np.array_equal(c.column_indices, d.column_indices.numpy())
    assert np.array_equal(c.values, d.values.numpy())


def test_circuit_shaped_stand_in():
    """circuit_csr: the circuit5M-shaped input of bench.py's `configs` (the reference's one published number is on circuit5M,
    README.md:116,137-138) -- exact nonzero count, valid sorted CSR, the stated row-length classes, a diagonal in every row,
    four giant rows, and the same matrix every time."""
    rows, nnz = 55_583, 595_243                        # 1/100 of circuit5M
    A = G.circuit_csr(rows, nnz, dtype=torch.float64, device="cpu")
    off = A.row_offsets.numpy().astype(np.int64); col = A.column_indices.numpy().astype(np.int64)
    assert A.rows == A.cols == rows and A.nnz == nnz and off[0] == 0 and off[-1] == nnz
    lens = np.diff(off)
    assert lens.min() >= 1 and abs(lens.mean() - nnz / rows) < 1e-9
    row_of = np.repeat(np.arange(rows), lens)
    assert np.all(np.diff(row_of * rows + col) >= 0) and col.min() >= 0 and col.max() < rows      # sorted by (row, column)
    giants = np.sort(lens)[-4:]
    assert np.allclose(giants / nnz, [160_000 / 59_524_291, 320_000 / 59_524_291, 645_000 / 59_524_291, 1_290_501 / 59_524_291], rtol=2e-2)
    rest = np.sort(lens)[:-4]
    assert 0.68 < (rest <= 8).mean() < 0.73 and rest.max() <= 91
    # the diagonal is there; most other entries are near it
    has_diag = np.zeros(rows, bool); has_diag[row_of[col == row_of]] = True
    assert has_diag.all()
    ordinary = lens[row_of] < 1000
    assert 0.75 < (np.abs(col - row_of)[ordinary] <= 2000).mean() < 0.88
    v = A.values.numpy()
    assert v.min() >= -1 and v.max() < 1 and abs(v.mean()) < 0.01
    B = G.circuit_csr(rows, nnz, dtype=torch.float64, device="cpu")
    assert torch.equal(A.column_indices, B.column_indices) and torch.equal(A.values, B.values) and torch.equal(A.row_offsets, B.row_offsets)
    # the stated full size is what the defaults give (no tensor built here)
    assert (G.CIRCUIT5M_ROWS, G.CIRCUIT5M_NNZ) == (5_558_326, 59_524_291)
