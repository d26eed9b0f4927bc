"""GPU parity tests (-m gpu): the HIP merge-path CsrMV, called through the
C ABI (include/mspmv.h via merge_spmv_amd.DeviceSpmv.CsrMV), against the
oracle on the same inputs.

Bars: tile coordinates and carry keys are integers -> bit-exact vs the
reference-pinned MergePathSearch; y is exact where the arithmetic is exact
(the reference's own generators: values = x = 1.0 -> y = row length) and
within the stated tolerance otherwise:
    |y[r] - g[r]| <= c * eps * s[r],  g = fp64-accumulated gold, s = sum|val*x|,
    c = 2*(ceil(log2(len_r+1)) + items_per_thread + 8), eps = 2^-24 | 2^-53,
    empty rows exactly 0                      (SURVEY.md 8d / BASELINE.md 2).
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

CASES = load_golden("matrices.json")["cases"]
DT = {"f32": (np.float32, 4), "f64": (np.float64, 8)}


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    import merge_spmv_amd as M_
    M_.load_library()          # raises if the HIP extension is missing: no fallback
    M_.set_tuning(4); M_.set_tuning(8)
    return M_


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_gpu(M, csr, x, sentinel=True, **kw):
    """One CsrMV through the C ABI; y pre-filled with NaN so unwritten rows show."""
    vb = csr.values.dtype.itemsize
    ws = M.CsrMVWorkspace(csr.rows, csr.nnz, torch.float32 if vb == 4 else torch.float64)
    y = torch.full((csr.rows,), float("nan"), dtype=ws.dtype, device="cuda")
    M.csrmv(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x.astype(csr.values.dtype)),
            y=y, num_cols=csr.cols, workspace=ws, **kw)
    torch.cuda.synchronize()
    return y.cpu().numpy(), ws


def check_strict(M, csr, x, y):
    g, s = O.spmv_gold_acc64(csr, x.astype(csr.values.dtype))
    ipt = M.launch_info(csr.rows, csr.nnz, csr.values.dtype.itemsize)["items_per_thread"]
    ok, worst = O.strict_check(csr, y, g, s, items_per_thread=ipt)
    assert ok, f"strict tolerance violated, worst ratio {worst}"
    return worst


def bound_ipt(M, rows, cols, nnz, vb):
    """The `items_per_thread` term of the strict bound for a call of these sizes: the nonzeros a thread sums serially
    (the compiled tile's items per thread, rounded up to whole 4-element chunks) plus one re-association per column-band
    pass the call may run (include/mspmv.h: mspmv_get_band_passes)."""
    return M.serial_sum_depth(rows, cols, nnz, vb)


def snapped_carries(csr, x, coords, tile_items, head_max):
    """What tile_kernel_snap leaves as the per-tile carry values (mspmv.h: snap_head_max): a boundary that falls <= head_max
    nonzeros into its row is moved back to the row's first nonzero -- the tile before it then has no open row (carry 0) --
    and a tile publishes the sum of the nonzeros of its open row that IT multiplied (its adopted head included)."""
    off = csr.row_offsets.astype(np.int64)
    prod = csr.values.astype(np.float64) * x.astype(np.float64)[csr.column_indices]
    n = coords.shape[0] - 1
    out = np.zeros(n)
    for t in range(n):
        (x0, y0), (x1, y1) = coords[t], coords[t + 1]
        rs0, rs1 = off[x0], off[x1]
        s0 = rs0 if y0 - rs0 <= head_max else y0
        if y1 - rs1 > head_max:
            lo = max(rs1, s0)
            out[t] = float(np.sum(prod[lo:y1]))
    return out


def check_tiles(M, csr, x, ws):
    """coords / carry keys bit-exact vs the oracle's tile emulation; carry values within tolerance."""
    vb = csr.values.dtype.itemsize
    info = M.launch_info(csr.rows, csr.nnz, vb)
    coords, keys, vals = M.debug_read_tiles(ws.buffer, csr.rows, csr.nnz, vb)
    want = O.tile_coords(csr, info["tile_items"])
    assert np.array_equal(coords, want[: info["num_tiles"] + 1])
    _, ck, cv = O.tiled_csrmv(csr, x.astype(csr.values.dtype), info["tile_items"])
    assert np.array_equal(keys, ck)
    if info["snap_head_max"] > 0 and info["num_tiles"] > 1:
        cv = snapped_carries(csr, x.astype(csr.values.dtype), coords.reshape(-1, 2).astype(np.int64), info["tile_items"], info["snap_head_max"])
        never_open = np.array([coords.reshape(-1, 2)[t + 1][1] - csr.row_offsets[coords.reshape(-1, 2)[t + 1][0]] <= info["snap_head_max"]
                               for t in range(info["num_tiles"])])
        assert np.all(vals[never_open] == 0)              # a snapped boundary leaves nothing open: exactly zero
    # a carry is a sum of up to tile_items products of magnitude <= max|val*x|, summed in a
    # different association order than the sequential emulation: bound the difference by
    # eps * tile_items * max|val|*max|x| (cancellation makes a relative bound meaningless)
    eps = 2.0 ** -23 if vb == 4 else 2.0 ** -52
    scale = float(np.abs(csr.values).max(initial=0)) * float(np.abs(x).max(initial=0)) * (info["tile_items"] + info["snap_head_max"])
    assert np.all(np.abs(vals.astype(np.float64) - np.asarray(cv, np.float64)) <= eps * scale + 1e-300)


def test_known_answer_device_spmv(M, golden_kat):
    """cub/device/device_spmv.cuh:90-123."""
    k = golden_kat
    for dtype in (np.float32, np.float64):
        csr = O.Csr(k["rows"], k["cols"], np.asarray(k["row_offsets"], np.int32),
                    np.asarray(k["column_indices"], np.int32), np.asarray(k["values"], dtype))
        y, _ = run_gpu(M, csr, np.asarray(k["x"], dtype))
        assert np.array_equal(y, np.asarray(k["y"], dtype))


@pytest.mark.parametrize("case", CASES, ids=[c["label"] for c in CASES])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_golden_matrices(M, case, prec):
    dtype, vb = DT[prec]
    args = [os.path.join(ROOT, case["args"][0])] if case["kind"] == "mtx" else case["args"]
    csr = O.make(case["kind"], *args, dtype=dtype)
    # the reference's protocol: x = 1 (gpu_spmv.cu:521-525) -> compare with SpmvGold
    x = np.ones(csr.cols, dtype)
    y, ws = run_gpu(M, csr, x)
    gold = O.spmv_gold(csr, x)
    assert O.compare_results(y, gold) == 0          # the reference's own PASS rule
    check_strict(M, csr, x, y)
    if case["kind"] != "mtx":
        assert np.array_equal(y, gold)              # exact arithmetic: y = row length
    check_tiles(M, csr, x, ws)
    # a non-trivial x catches a wrong column gather (all-ones cannot)
    x2 = np.arange(1, csr.cols + 1, dtype=dtype) * 0.5
    y2, _ = run_gpu(M, csr, x2)
    check_strict(M, csr, x2, y2)


def random_csr(rng, rows, cols, lens, dtype):
    off = np.zeros(rows + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    col = rng.integers(0, cols, size=nnz).astype(np.int32)
    # sort columns within rows like CsrMatrix::Init (sparse_matrix.h:676)
    rowid = np.repeat(np.arange(rows), lens)
    order = np.lexsort((col, rowid))
    col = col[order]
    val = rng.uniform(-1, 1, size=nnz).astype(dtype)
    return O.Csr(rows, cols, off.astype(np.int32), col, val)


SHAPES = {
    "uniform_short": lambda rng: (3000, 3000, rng.integers(0, 6, 3000)),
    "uniform_32": lambda rng: (2000, 50000, np.full(2000, 32)),
    "power_law": lambda rng: (5000, 5000, np.minimum((rng.pareto(1.2, 5000) * 3).astype(np.int64), 20000)),
    "all_empty": lambda rng: (10000, 50, np.zeros(10000, np.int64)),
    "leading_trailing_empty": lambda rng: (9000, 100, np.concatenate([np.zeros(4000, np.int64), rng.integers(1, 9, 1000), np.zeros(4000, np.int64)])),
    "one_giant_row": lambda rng: (1, 777, np.array([200000])),
    "giant_row_between_empties": lambda rng: (6001, 1000, np.concatenate([np.zeros(3000, np.int64), [150000], np.zeros(3000, np.int64)])),
    "giant_plus_sprinkle": lambda rng: (40000, 40000, np.where(np.arange(40000) == 20000, 120000, (np.arange(40000) % 512 == 0).astype(np.int64))),
    "single_col": lambda rng: (5000, 1, rng.integers(0, 3, 5000)),
    "single_tile": lambda rng: (10, 10, rng.integers(0, 4, 10)),
    "one_row_one_nnz": lambda rng: (1, 1, np.array([1])),
    "exact_tile_multiple": lambda rng: (1792, 64, np.full(1792, 3)),
}


PATHS = {"one_launch_small_shape": 0, "classic_small_shape": 0x40000000, "one_launch_large_shape": 16, "one_launch_runs_of_8": 0x3000010,
         "one_launch_round_robin": 0xF000010, "classic_three_launch": 0x40000010, "classic_interp_coords": 0x60000010, "classic_search_kernel": 0x40000018,
         "classic_contiguous_map": 0x4E000010, "classic_atomic_fix": 0x12, "classic_multilevel_fix": 0x90, "reference_walk": 4 | 16,
         # non-temporal streams whatever the size (the default beyond 256 MB): in fp64 the values are fetched line by line over the
         # wave and staged in wire layout (ld_stream4_linewise, wire_cols) -- here on every shape family, ragged array tails included
         "one_launch_nontemporal": 32 | 16, "one_launch_nontemporal_small_shape": 32, "classic_nontemporal": 0x40000030,
         # closed tiles of short rows through the general flag / segmented-scan reduction as well (MSPMV_TUNE_NO_LEAN)
         "one_launch_general_reduction": -0x80000000 | 16, "one_launch_general_reduction_nontemporal": -0x80000000 | 32}


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("path", sorted(PATHS))
def test_random_and_degenerate_shapes(M, shape, prec, path):
    """Every dispatch path on the same inputs.  The default is ONE launch of tile_kernel_snap -- row-snapped tiles on coordinate
    hints that are garbage here (fresh temp storage), so every tile searches its boundaries -- with the tile shape the sizes
    select, or with MSPMV_TUNE_NO_FUSED the large-problem shape; MSPMV_TUNE_TWO_LAUNCH (and the fix-up / search options) run
    the classic three launches instead (coordinate pass by scatter, interpolation search or the 64-ary wave search;
    tile_kernel_vec with one carry per tile; one-pass, atomic or multi-level fix-up; "classic_contiguous_map" = runs of 2^14
    tiles per XCD, the mapping family the prepared plan uses); "reference_walk" = the dword-per-lane kernel with the
    reference's per-thread merge-path search + walk inside the tile (MSPMV_TUNE_NO_VEC)."""
    dtype, vb = DT[prec]
    rng = np.random.default_rng(sum(map(ord, shape)))
    rows, cols, lens = SHAPES[shape](rng)
    csr = random_csr(rng, rows, cols, np.asarray(lens, np.int64), dtype)
    x = rng.uniform(-1, 1, size=cols).astype(dtype)
    try:
        M.set_tuning(vb, 0, 0, PATHS[path])
        y, ws = run_gpu(M, csr, x)
        assert not np.isnan(y).any(), "a row was never written"
        check_strict(M, csr, x, y)
        check_tiles(M, csr, x, ws)
        # bitwise reproducible (deterministic fix-up / fixed order of the taken carries; the atomic fix-up is the one exception)
        for _ in range(3):
            y_again, _ = run_gpu(M, csr, x)
            assert path == "classic_atomic_fix" or np.array_equal(y, y_again)
    finally:
        M.set_tuning(vb)


COMPACT_SHAPES = dict(SHAPES)
COMPACT_SHAPES.update({
    # what the compact front end's fast lane is made for, and its edges: closed lean tiles of short rows, rows of 9..16 and > 16
    # nonzeros inside lean tiles (second batch, 16-lane groups), tiles of > 511 rows (further row-offset rounds), every residue of
    # nnz and rows + 1 modulo 4 (the ragged last chunk of the nonzero arrays), empty rows around, a last tile of a few items
    "five_point": lambda rng: (20000, 20000, np.full(20000, 5)),
    "short_0_to_8": lambda rng: (30011, 9000, rng.integers(0, 9, 30011)),
    "short_with_16s": lambda rng: (12000, 5000, np.where(rng.random(12000) < 0.05, rng.integers(9, 17, 12000), rng.integers(0, 5, 12000))),
    "short_with_long": lambda rng: (15000, 5000, np.where(rng.random(15000) < 0.004, rng.integers(17, 1500, 15000), rng.integers(0, 4, 15000))),
    "mostly_empty": lambda rng: (60000, 300, (rng.random(60000) < 0.2).astype(np.int64) * rng.integers(1, 4, 60000)),
    "ragged_1": lambda rng: (7001, 700, np.concatenate([np.full(7000, 3), [2]])),
    "ragged_2": lambda rng: (7002, 700, np.concatenate([np.full(7001, 3), [3]])),
    "ragged_3": lambda rng: (7003, 700, np.concatenate([np.full(7002, 3), [1]])),
    "ragged_tail_tile_of_3_items": lambda rng: (3585, 100, np.concatenate([np.full(3583, 1), [0, 2]])),
})


@pytest.mark.parametrize("shape", sorted(COMPACT_SHAPES))
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_compact_front_end_is_bitwise_the_general_kernel(M, shape, prec):
    """Small problems (up to 2304 tiles) run tile_kernel_snap behind its compact front end (mspmv_kernels.hpp: compact_front;
    reference: the small-problem special case of dispatch_spmv_orig.cuh:674-679 / agent_spmv_orig.cuh:867-891): closed lean tiles
    on verified hints take a fast lane of ~200 instructions, every other tile the general body of the same kernel.  Same y, bit
    for bit, as the general kernel (mspmv_set_compact_tiles(-1)) -- on the first call (garbage hints: every tile falls through
    to the general body), on the second (right hints: the fast lane) and with x changed; right coordinates and carries afterwards;
    strict tolerance against the oracle."""
    dtype, vb = DT[prec]
    tdt = torch.float32 if vb == 4 else torch.float64
    rng = np.random.default_rng(sum(map(ord, shape)) + 5)
    rows, cols, lens = COMPACT_SHAPES[shape](rng)
    csr = random_csr(rng, rows, cols, np.asarray(lens, np.int64), dtype)
    xs = [rng.uniform(-1, 1, size=cols).astype(dtype) for _ in range(2)]
    dv, do, dc = dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices)

    def run(ws, x):
        y = torch.full((csr.rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(dv, do, dc, dev(x), y=y, num_cols=csr.cols, workspace=ws)
        torch.cuda.synchronize()
        return y.cpu().numpy()

    try:
        M.set_compact_tiles(-1)
        ws_g = M.CsrMVWorkspace(csr.rows, csr.nnz, tdt)
        want = [run(ws_g, xs[0]), run(ws_g, xs[0]), run(ws_g, xs[1])]
        assert np.array_equal(want[0], want[1])
        M.set_compact_tiles(0)
        ws_c = M.CsrMVWorkspace(csr.rows, csr.nnz, tdt)
        ws_c.buffer.random_(0, 255)                     # garbage hints
        got = [run(ws_c, xs[0]), run(ws_c, xs[0]), run(ws_c, xs[1])]
        for g, w in zip(got, want):
            assert not np.isnan(g).any(), "a row was never written"
            assert np.array_equal(g, w)
        check_strict(M, csr, xs[1], got[2])
        check_tiles(M, csr, xs[1], ws_c)
        # hints of the general kernel serve the compact one and the other way round
        assert np.array_equal(run(ws_g, xs[1]), want[2])
        M.set_compact_tiles(-1)
        assert np.array_equal(run(ws_c, xs[0]), want[0])
    finally:
        M.set_compact_tiles(0)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_one_launch_path_does_not_depend_on_its_coordinate_hints(M, prec):
    """tile_kernel_snap reads its tile boundaries from temp storage as HINTS and verifies them against the row offsets:
    the result must be bit for bit the same whether the hints are garbage (fresh buffer, every byte pattern), right (second
    call on the buffer, or after mspmv_csrmv_prepare), or right for ANOTHER matrix of the same rows / nnz (a buffer reused
    for a different sparsity pattern) -- and the buffer must hold the right coordinates afterwards."""
    dtype, vb = DT[prec]
    tdt = torch.float32 if vb == 4 else torch.float64
    rng = np.random.default_rng(77)
    rows = 250_000
    for family in ("short", "skewed"):
        lens = rng.integers(0, 12, rows) if family == "short" else np.minimum((rng.pareto(1.1, rows) * 2).astype(np.int64), 60000)
        a = random_csr(rng, rows, rows, lens, dtype)
        b = random_csr(rng, rows, rows, rng.permutation(lens), dtype)          # same rows / nnz, another pattern
        x = rng.uniform(-1, 1, rows).astype(dtype)
        try:
            M.set_tuning(vb, 0, 0, 16)
            info = M.launch_info(a.rows, a.nnz, vb)
            assert info["snap_head_max"] > 0 and info["fixup_levels"] == 0
            da = [dev(v) for v in (a.values, a.row_offsets, a.column_indices, x)]
            db = [dev(v) for v in (b.values, b.row_offsets, b.column_indices, x)]
            ws = M.CsrMVWorkspace(a.rows, a.nnz, tdt)
            results = []
            for fill in (0x00, 0xFF, 0x7F, 0x01):                             # garbage of several kinds
                ws.buffer.fill_(fill)
                results.append(M.csrmv(*da, workspace=ws).clone())
            results.append(M.csrmv(*da, workspace=ws).clone())                 # hints now right
            coords, _, _ = M.debug_read_tiles(ws.buffer, a.rows, a.nnz, vb)
            assert np.array_equal(coords, O.tile_coords(a, info["tile_items"])[: info["num_tiles"] + 1])
            yb = M.csrmv(*db, workspace=ws).clone()                            # hints right for the WRONG matrix
            results.append(M.csrmv(*da, workspace=ws).clone())                 # ... and now wrong for this one
            ws2 = M.CsrMVWorkspace(a.rows, a.nnz, tdt).prepare(da[1])
            results.append(M.csrmv(*da, workspace=ws2).clone())                # prepared
            torch.cuda.synchronize()
            for r in results[1:]:
                assert torch.equal(r, results[0])
            check_strict(M, a, x, results[0].cpu().numpy())
            check_strict(M, b, x, yb.cpu().numpy())
            coords, _, _ = M.debug_read_tiles(ws.buffer, a.rows, a.nnz, vb)
            assert np.array_equal(coords, O.tile_coords(a, info["tile_items"])[: info["num_tiles"] + 1])
        finally:
            M.set_tuning(vb)


def test_empty_matrix_and_zero_rows(M):
    for dtype in (np.float32, np.float64):
        csr = O.Csr(0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, dtype))
        y, _ = run_gpu(M, csr, np.ones(5, dtype))
        assert y.size == 0
        csr = O.Csr(7, 0, np.zeros(8, np.int32), np.zeros(0, np.int32), np.zeros(0, dtype))
        y, _ = run_gpu(M, csr, np.ones(0, dtype))
        assert np.array_equal(y, np.zeros(7, dtype))


def test_large_path_launch_log(M, capfd):
    csr = O.make("grid3d", 12, dtype=np.float64)
    for flags, expect, absent in ((16, ("tile_kernel_snap",), ("coords_scatter_kernel", "fixup")),
                                  (16 | 0x40000000, ("coords_scatter_kernel", "tile_kernel_vec", "fixup_onepass_kernel"), ("tile_kernel_snap",))):
        try:
            M.set_tuning(8, 0, 0, flags)
            y, _ = run_gpu(M, csr, np.ones(csr.cols), debug_synchronous=True)
        finally:
            M.set_tuning(8)
        assert np.array_equal(y, O.spmv_gold(csr, np.ones(csr.cols)))
        out = capfd.readouterr().out
        assert all(k in out for k in expect) and not any(k in out for k in absent), out


def test_all_ones_giant_row_is_exact(M):
    """C4-style closed form: one row of 2^22 ones (exactly representable in fp32)."""
    n = 1 << 22
    lens = np.zeros(2049, np.int64); lens[1024] = n; lens[::256] += 1
    rng = np.random.default_rng(3)
    off = np.zeros(lens.size + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    csr = O.Csr(lens.size, 4096, off.astype(np.int32), (np.arange(nnz) % 4096).astype(np.int32), np.ones(nnz, np.float32))
    y, _ = run_gpu(M, csr, np.ones(4096, np.float32))
    assert np.array_equal(y, lens.astype(np.float32))


@pytest.mark.parametrize("vb,block,ipt", [(4, 256, 7), (4, 256, 11), (8, 256, 7), (8, 256, 11)])
# 16 = the large-problem tile shape whatever the size, +2 atomic fix-up (classic pipeline), 4 = dword-per-lane
# kernel with the reference's in-tile walk, +8 binary-search coordinate pass, 32/64 forced stream policy,
# 128 = multi-level fix-up (default: one launch); bits 24-27 = block->tile mapping (0xF: round-robin, 3: runs of 8);
# 0x20000000 = coordinates by the per-boundary interpolation search (the default from 10 M rows up; scatter pass below)
@pytest.mark.parametrize("flags", [0, 2, 4, 16, 18, 20, 24, 48, 80, 128, 144, 0xF000010, 0x3000010, 0x20000010, 0x10000010, 0x40000000, 0x40000010, 0x4F000010, 0x60000010])
def test_every_compiled_tile_shape(M, vb, block, ipt, flags):
    dtype = np.float32 if vb == 4 else np.float64
    rng = np.random.default_rng(block * 100 + ipt)
    lens = np.minimum((rng.pareto(1.1, 20000) * 2).astype(np.int64), 50000)
    lens[7777] = 90000
    csr = random_csr(rng, 20000, 20000, lens, dtype)
    x = rng.uniform(-1, 1, size=csr.cols).astype(dtype)
    try:
        M.set_tuning(vb, block, ipt, flags)
        info = M.launch_info(csr.rows, csr.nnz, vb)
        assert (info["block_threads"], info["items_per_thread"], info["flags"]) == (block, ipt, flags)
        y, ws = run_gpu(M, csr, x)
        check_strict(M, csr, x, y)
        check_tiles(M, csr, x, ws)
    finally:
        M.set_tuning(vb)


def test_axpby_extension(M):
    rng = np.random.default_rng(11)
    for dtype in (np.float32, np.float64):
        csr = random_csr(rng, 4000, 4000, rng.integers(0, 40, 4000), dtype)
        x = rng.uniform(-1, 1, 4000).astype(dtype)
        y0 = rng.uniform(-1, 1, 4000).astype(dtype)
        g, s = O.spmv_gold_acc64(csr, x)
        for flags in (0, 16, 20, 0x40000000, 0x40000010):          # one launch (small / large shape), dword-per-lane fallback, classic three launches
            M.set_tuning(csr.values.dtype.itemsize, 0, 0, flags)
            try:
                for alpha, beta in ((1.0, 0.0), (2.5, 0.0), (1.0, 1.0), (-0.5, 3.0)):
                    y = dev(y0.copy())
                    M.csrmv(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x), y=y, alpha=alpha, beta=beta)
                    want = alpha * g + beta * y0.astype(np.float64)
                    tol = (2.0 ** -20 if dtype == np.float32 else 2.0 ** -48) * (abs(alpha) * s + abs(beta) * np.abs(y0) + 1e-30)
                    assert np.all(np.abs(y.cpu().numpy() - want) <= tol)
            finally:
                M.set_tuning(csr.values.dtype.itemsize)
        # beta == 0 must not read y (NaN there would poison it otherwise)
        y = torch.full((4000,), float("nan"), dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
        M.csrmv(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x), y=y, alpha=2.0, beta=0.0)
        assert not torch.isnan(y).any()


def test_two_phase_temp_storage_on_device(M):
    csr = O.make("grid2d", 40, dtype=np.float32)
    st, size = M.DeviceSpmv.CsrMV(None, 0, dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices),
                                  dev(np.ones(csr.cols, np.float32)), torch.empty(csr.rows, device="cuda"),
                                  csr.rows, csr.cols, csr.nnz)
    assert st == 0 and size > 0
    small = torch.empty(size - 1, dtype=torch.uint8, device="cuda")
    st, _ = M.DeviceSpmv.CsrMV(small, size - 1, dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices),
                               dev(np.ones(csr.cols, np.float32)), torch.empty(csr.rows, device="cuda"),
                               csr.rows, csr.cols, csr.nnz)
    assert st == 1


def test_runs_on_a_side_stream_and_with_debug_sync(M, capfd):
    csr = O.make("grid3d", 12, dtype=np.float64)
    x = np.ones(csr.cols)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y, _ = run_gpu(M, csr, x, stream=s, debug_synchronous=True)
    assert np.array_equal(y, O.spmv_gold(csr, x))
    out = capfd.readouterr().out
    assert "tile_kernel_snap" in out


def test_single_hip_runtime_loaded(M):
    """libmspmv.so must bind to the HIP runtime torch already loaded."""
    maps = open("/proc/self/maps").read()
    libs = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(libs) == 1, libs
    assert any("libmspmv.so" in line for line in maps.splitlines())


def test_full_size_c2_properties(M):
    """BASELINE config 2 at full size (3.125M rows x 32 nnz/row = 100M nnz,
    fp32): the oracle gold is O(seconds) in C/OpenMP, so check the strict
    tolerance on every row, plus linearity A(ax+by) = aAx + bAy."""
    from merge_spmv_amd import generators as G
    rows = cols = 3_125_000
    A = G.uniform_csr(rows, cols, 32, dtype=torch.float32, device="cuda")
    x = G.uniform_pm1(G.SEED_C2 + 2, cols, torch.float32, "cuda")
    ws = M.CsrMVWorkspace(rows, A.nnz, torch.float32)
    y = M.csrmv(A.values, A.row_offsets, A.column_indices, x, workspace=ws)
    torch.cuda.synchronize()
    csr = O.Csr(rows, cols, A.row_offsets.cpu().numpy(), A.column_indices.cpu().numpy(), A.values.cpu().numpy())
    xh = x.cpu().numpy()
    g, s = O.spmv_gold_acc64(csr, xh)
    ok, worst = O.strict_check(csr, y.cpu().numpy(), g, s, items_per_thread=bound_ipt(M, rows, cols, A.nnz, 4))
    assert ok, worst
    x2 = G.uniform_pm1(99, cols, torch.float32, "cuda")
    y2 = M.csrmv(A.values, A.row_offsets, A.column_indices, x2, workspace=ws).clone()
    y3 = M.csrmv(A.values, A.row_offsets, A.column_indices, 2.0 * x - 0.5 * x2, workspace=ws)
    lin = (2.0 * y.double() - 0.5 * y2.double())
    err = (y3.double() - lin).abs()
    assert float((err / (torch.from_numpy(s).cuda() * 2.0 ** -18 + 1e-12)).max()) <= 4.0


@pytest.mark.parametrize("parts", [1, 2, 4, 8])
def test_sharded_merge_partition_on_one_gpu(M, parts):
    """every part's swath through the HIP CsrMV as a local CSR (its extra
    last row is the carry), the carries exchanged (here: copied, all parts live on one
    GPU) and applied by mspmv_mg_apply_carries; equals the unsharded result."""
    from merge_spmv_amd import multi_gpu as MG
    rng = np.random.default_rng(parts)
    lens = np.minimum((rng.pareto(1.1, 30000) * 2).astype(np.int64), 40000)
    lens[12345] = 400000                                    # spans several parts
    csr = random_csr(rng, 30000, 30000, lens, np.float64)
    x = rng.uniform(-1, 1, csr.cols)
    xd = dev(x)
    ops = [MG.ShardedCsrMV(MG.shard_from_host_csr(csr.row_offsets, csr.column_indices, csr.values, csr.cols, g, parts))
           for g in range(parts)]
    carries = torch.zeros(parts, dtype=torch.float64, device="cuda")
    for g, op in enumerate(ops):
        s = op.shard
        st, _ = M.DeviceSpmv.CsrMV(op.workspace.buffer, op.workspace.bytes, s.values, s.row_offsets, s.column_indices, xd,
                                   op.y_local, s.local_rows, s.num_cols, s.local_nnz)
        assert st == 0
        carries[g] = op.y_local[s.local_rows - 1]
    y = np.full(csr.rows, np.nan)
    for g, op in enumerate(ops):
        op.carries.copy_(carries)
        op._apply_carries()
        torch.cuda.synchronize()
        s = op.shard
        y[int(s.row_split[g]): int(s.row_split[g + 1])] = op.y_local[: s.owned_rows].cpu().numpy()
    check_strict(M, csr, x, y)


def _host_csr(A):
    return O.Csr(A.rows, A.cols, A.row_offsets.cpu().numpy(), A.column_indices.cpu().numpy(), A.values.cpu().numpy())


def test_full_size_c4_degenerate(M):
    """BASELINE config 4 at full size: fp32, 16 777 216 rows, one row of 67 108 864 nonzeros
    (spanning ~24 000 tiles: long equal-key carry runs through both fix-up levels), one nonzero
    in every 4096-th other row, everything else empty.  All-ones: closed form, exact in fp32
    (2^26 is representable).  Random values: strict tolerance on every row."""
    from merge_spmv_amd import generators as G
    A = G.degenerate_csr(dtype=torch.float32, device="cuda", ones=True)
    x = torch.ones(A.cols, dtype=torch.float32, device="cuda")
    y = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
    lens = torch.diff(A.row_offsets.to(torch.int64)).to(torch.float32)
    assert torch.equal(y, lens)
    assert int((lens == 0).sum()) > 16_000_000 and float(lens.max()) == float(1 << 26)
    B = G.degenerate_csr(dtype=torch.float32, device="cuda", ones=False)
    xr = G.uniform_pm1(G.SEED_C4 + 2, B.cols, torch.float32, "cuda")
    yr = M.csrmv(B.values, B.row_offsets, B.column_indices, xr, num_cols=B.cols)
    torch.cuda.synchronize()
    csr = _host_csr(B)
    g, s = O.spmv_gold_acc64(csr, xr.cpu().numpy())
    ok, worst = O.strict_check(csr, yr.cpu().numpy(), g, s, items_per_thread=bound_ipt(M, B.rows, B.cols, B.nnz, 4))
    assert ok, worst


def _c3_check(M, A, x, label):
    """every row of a config-3 matrix against the oracle's fp64-accumulated gold (strict bound), bitwise repeatable"""
    y = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
    y2 = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    csr = _host_csr(A)
    g, s = O.spmv_gold_acc64(csr, x.cpu().numpy())
    ok, worst = O.strict_check(csr, y.cpu().numpy(), g, s, items_per_thread=bound_ipt(M, A.rows, A.cols, A.nnz, 8))
    assert ok, (label, worst)
    return csr, worst


def _load_real_c3(name, dtype=np.float64):
    """MSPMV_C3_DIR=<dir holding webbase-1M.mtx / com-Orkut.mtx> (SuiteSparse, ufl_matrices.txt:2379): the real
    matrices through the product's own Matrix Market reader (host/sparse_matrix.hpp via libmspmv_host.so)."""
    d = os.environ.get("MSPMV_C3_DIR")
    path = os.path.join(d, name) if d else None
    if not path or not os.path.exists(path):
        return None
    import ctypes
    H = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
    H.mspmv_host_matrix_create.restype = ctypes.c_void_p
    H.mspmv_host_matrix_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    H.mspmv_host_matrix_destroy.argtypes = [ctypes.c_void_p]
    H.mspmv_host_matrix_shape.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    H.mspmv_host_matrix_copy.argtypes = [ctypes.c_void_p] * 4
    st = ctypes.c_int()
    h = H.mspmv_host_matrix_create(b"mtx", 0, 0, path.encode(), int(dtype == np.float32), ctypes.byref(st))
    try:
        assert st.value == 0, f"the product's Matrix Market reader refused {path}"
        r, c, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        H.mspmv_host_matrix_shape(h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(n))
        off = np.zeros(r.value + 1, np.int32); col = np.zeros(max(n.value, 1), np.int32); val = np.zeros(max(n.value, 1), dtype)
        H.mspmv_host_matrix_copy(h, off.ctypes.data, col.ctypes.data, val.ctypes.data)
        return O.Csr(r.value, c.value, off, col[: n.value], val[: n.value])
    finally:
        H.mspmv_host_matrix_destroy(h)


def test_config3_webbase_sized_fp64(M):
    """BASELINE config 3, webbase-1M (1 000 005^2, 3 105 536 nonzeros, ufl_matrices.txt:2379): the real file when
    MSPMV_C3_DIR holds it, else its size-matched stand-in -- R-MAT scale 20 with exactly 3 105 536 edges, duplicates kept,
    fp64 -- every row against the oracle."""
    from merge_spmv_amd import generators as G
    real = _load_real_c3("webbase-1M.mtx")
    if real is not None:
        A = G.DeviceCsr(real.rows, real.cols, dev(real.row_offsets), dev(real.column_indices), dev(real.values))
    else:
        A = G.rmat_csr(G.C3_WEBBASE_SCALE, G.C3_WEBBASE_EDGES, dtype=torch.float64, device="cuda", seed=G.SEED_C3)
        assert A.nnz == 3_105_536 and A.rows == 1 << 20
    x = G.uniform_pm1(G.SEED_C3 + 2, A.cols, torch.float64, "cuda")
    csr, _ = _c3_check(M, A, x, "webbase-sized")
    lens = np.diff(csr.row_offsets.astype(np.int64))
    assert lens.max() > 100 * max(lens.mean(), 1)           # power-law: the load-balance stress the config is there for


def test_config3_orkut_sized_fp64(M):
    """BASELINE config 3, com-Orkut (3 072 441^2, 117 185 083 stored entries of a symmetric pattern matrix ->
    234 370 166 nonzeros): the real file when MSPMV_C3_DIR holds it, else the size-matched stand-in -- R-MAT scale 22,
    117 185 083 stored entries mirrored the way InitMarket mirrors a `symmetric` file (sparse_matrix.h:362-368), fp64 --
    every one of the 4 194 304 rows against the oracle."""
    from merge_spmv_amd import generators as G
    real = _load_real_c3("com-Orkut.mtx")
    if real is not None:
        A = G.DeviceCsr(real.rows, real.cols, dev(real.row_offsets), dev(real.column_indices), dev(real.values))
    else:
        A = G.rmat_symmetric_csr(G.C3_ORKUT_SCALE, G.C3_ORKUT_EDGES, dtype=torch.float64, device="cuda", seed=G.SEED_C3)
        assert 2 * G.C3_ORKUT_EDGES - (1 << 16) < A.nnz <= 2 * G.C3_ORKUT_EDGES and A.rows == 1 << 22
        # symmetric by construction: the row and the column histograms agree
        assert torch.equal(torch.bincount(A.column_indices.long(), minlength=A.rows), torch.diff(A.row_offsets.long()))
    x = G.uniform_pm1(G.SEED_C3 + 2, A.cols, torch.float64, "cuda")
    csr, _ = _c3_check(M, A, x, "Orkut-sized")
    lens = np.diff(csr.row_offsets.astype(np.int64))
    assert lens.max() > 1000 * max(lens.mean(), 1)          # genuinely skewed


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("kind", ["dense5", "grid2d", "ragged_0_8", "upto16"])
def test_short_row_matrices_are_bitwise_the_sequential_definition(M, kind, prec):
    """Closed tiles of short rows take the lean reduction (consume_tile_rows): every row of at most 16 nonzeros is added up
    left to right from +0.0, product by product -- the very order of the reference's SpmvGold (gpu_spmv.cu:262-278,
    cpu_spmv.cpp:257-277) -- so on a matrix whose tiles are all such tiles y is BIT FOR BIT the oracle's sequential gold in
    the matrix's own precision, at a size of the small tile shape and at one of the large shape; with the general
    reduction forced (MSPMV_TUNE_NO_LEAN) it is the same sums in another association (strict bound only)."""
    dtype, vb = DT[prec]
    rng = np.random.default_rng(len(kind) * 3 + vb)
    for rows in (3000, 1_300_000):
        if kind == "dense5":
            lens = np.full(rows, 5); cols = 5
        elif kind == "grid2d":
            w = int(np.sqrt(rows)); rows = w * w
            lens = None
        elif kind == "ragged_0_8":
            lens = rng.integers(0, 9, rows); cols = rows
        else:       # mostly short, some of 9..16: the average stays <= 8 in every tile
            lens = np.where(rng.random(rows) < 0.2, rng.integers(9, 17, rows), rng.integers(0, 5, rows)); cols = rows
        if lens is None:
            g = O.make("grid2d", w, dtype=dtype)
            csr = O.Csr(g.rows, g.cols, g.row_offsets, g.column_indices, rng.uniform(-1, 1, g.nnz).astype(dtype))
        else:
            csr = random_csr(rng, rows, cols, np.asarray(lens, np.int64), dtype)
        x = rng.uniform(-1, 1, csr.cols).astype(dtype)
        y, _ = run_gpu(M, csr, x)
        gold = O.spmv_gold(csr, x)
        assert np.array_equal(y.view(np.uint32 if vb == 4 else np.uint64), gold.view(np.uint32 if vb == 4 else np.uint64)), \
            (kind, prec, rows, int((y != gold).sum()))
        check_strict(M, csr, x, y)
        try:
            M.set_tuning(vb, 0, 0, -0x80000000)             # MSPMV_TUNE_NO_LEAN
            y2, _ = run_gpu(M, csr, x)
        finally:
            M.set_tuning(vb)
        check_strict(M, csr, x, y2)


def test_large_fp64_matrix_of_short_rows_over_a_tiny_x_takes_the_small_shape(M):
    """mspmv_api.hip: skinny_rule -- the reference's --dense=<cols> family (cpu_spmv.cpp:581-587; --dense=5 is BASELINE config 1's
    matrix) beyond 8 M path items runs the small tile shape behind the compact front end.  launch_info with the column count reports
    that layout, without it the default one with temp_bytes large enough for either; y is BIT FOR BIT the oracle's sequential gold
    (every tile is a closed lean tile) and bit for bit what the default shape gives (compact launches off: the rule does not apply)."""
    rng = np.random.default_rng(55)
    rows, cols = 1_700_000, 5                                     # 10.2 M path items
    csr = random_csr(rng, rows, cols, np.full(rows, 5, np.int64), np.float64)
    x = rng.uniform(-1, 1, cols)
    plain = M.launch_info(csr.rows, csr.nnz, 8)
    exact = M.launch_info(csr.rows, csr.nnz, 8, num_cols=cols)
    assert plain["items_per_thread"] == 11 and exact["items_per_thread"] == 7 and exact["num_tiles"] > plain["num_tiles"]
    assert plain["temp_bytes"] == exact["temp_bytes"] >= 16 * exact["num_tiles"]
    assert M.launch_info(csr.rows, csr.nnz, 8, num_cols=csr.rows)["items_per_thread"] == 11        # (x is not tiny: the default shape)
    assert M.launch_info(csr.rows, csr.nnz, 4, num_cols=cols)["items_per_thread"] == 11            # (fp32: the default shape)
    size = ctypes.c_size_t(0)                                     # the C size query knows the column count
    assert M.load_library().mspmv_csrmv_f64(None, ctypes.byref(size), None, None, None, None, None, rows, cols, csr.nnz, None, 0) == 0
    assert size.value == exact["temp_bytes"]
    y, ws = run_gpu(M, csr, x)
    y2t = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")       # the second call on the workspace: on hints
    M.csrmv(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x), y=y2t, num_cols=cols, workspace=ws)
    torch.cuda.synchronize(); y2 = y2t.cpu().numpy()
    gold = O.spmv_gold(csr, x)
    assert np.array_equal(y.view(np.uint64), gold.view(np.uint64)) and np.array_equal(y2.view(np.uint64), gold.view(np.uint64))
    check_strict(M, csr, x, y)
    try:
        M.set_compact_tiles(-1)
        assert M.launch_info(csr.rows, csr.nnz, 8, num_cols=cols)["items_per_thread"] == 11
        y3, _ = run_gpu(M, csr, x)
    finally:
        M.set_compact_tiles(0)
    assert np.array_equal(y3.view(np.uint64), y.view(np.uint64))
    # y = alpha A x + beta y through the same rule (mspmv_csrmv_axpby_f64): bit for bit what the default shape gives
    y0 = torch.from_numpy(rng.uniform(-1, 1, rows)).cuda()
    outs = []
    for ct in (0, -1):
        try:
            M.set_compact_tiles(ct)
            ya = y0.clone()
            M.csrmv(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x), y=ya, num_cols=cols, alpha=0.5, beta=-1.0)
            torch.cuda.synchronize(); outs.append(ya.cpu().numpy())
        finally:
            M.set_compact_tiles(0)
    assert np.array_equal(outs[0].view(np.uint64), outs[1].view(np.uint64))
    assert np.allclose(outs[0], 0.5 * gold - y0.cpu().numpy(), rtol=0, atol=1e-13)
    # a workspace that is too small for the small shape's layout runs the default shape (no error)
    dv = torch.from_numpy(csr.values).cuda(); do = torch.from_numpy(csr.row_offsets).cuda(); dc = torch.from_numpy(csr.column_indices).cuda()
    dx = torch.from_numpy(x).cuda(); dy = torch.empty(rows, dtype=torch.float64, device="cuda")
    small = ctypes.c_size_t(0)
    try:
        M.set_compact_tiles(-1)
        assert M.load_library().mspmv_csrmv_f64(None, ctypes.byref(small), None, None, None, None, None, rows, cols, csr.nnz, None, 0) == 0
    finally:
        M.set_compact_tiles(0)
    assert small.value < size.value
    buf = torch.empty(small.value, dtype=torch.uint8, device="cuda")
    st = M.load_library().mspmv_csrmv_f64(ctypes.c_void_p(buf.data_ptr()), ctypes.byref(small), ctypes.c_void_p(dv.data_ptr()), ctypes.c_void_p(do.data_ptr()),
                                          ctypes.c_void_p(dc.data_ptr()), ctypes.c_void_p(dx.data_ptr()), ctypes.c_void_p(dy.data_ptr()), rows, cols, csr.nnz, None, 0)
    torch.cuda.synchronize()
    assert st == 0 and np.array_equal(dy.cpu().numpy().view(np.uint64), gold.view(np.uint64))


def test_circuit5m_shaped_stand_in_full_size_fp64(M):
    """The reference's one published number is on circuit5M (README.md:116,137-138: 5 558 326^2, 59 524 291 nonzeros, fp64);
    the file cannot be fetched offline, so bench.py's `configs` run a seeded stand-in of exactly those sizes with a circuit
    matrix's row-length spread (generators.circuit_csr): every row against the oracle, fp64 and fp32, and the giant rows
    (1.29 M nonzeros: ~460 tiles) through the published-record path."""
    from merge_spmv_amd import generators as G
    A = G.circuit_csr(dtype=torch.float64, device="cuda")
    assert (A.rows, A.cols, A.nnz) == (5_558_326, 5_558_326, 59_524_291)
    x = G.uniform_pm1(G.SEED_CIRCUIT + 9, A.cols, torch.float64, "cuda")
    csr, worst = _c3_check(M, A, x, "circuit5M-shaped")
    lens = np.diff(csr.row_offsets.astype(np.int64))
    assert lens.max() == 1_290_501 and 10.70 < lens.mean() < 10.72
    A32 = G.DeviceCsr(A.rows, A.cols, A.row_offsets, A.column_indices, A.values.float())
    y = M.csrmv(A32.values, A32.row_offsets, A32.column_indices, x.float(), num_cols=A.cols)
    torch.cuda.synchronize()
    csr32 = O.Csr(csr.rows, csr.cols, csr.row_offsets, csr.column_indices, A32.values.cpu().numpy())
    g, s = O.spmv_gold_acc64(csr32, x.float().cpu().numpy())
    ok, w32 = O.strict_check(csr32, y.cpu().numpy(), g, s, items_per_thread=bound_ipt(M, A.rows, A.cols, A.nnz, 4))
    assert ok, w32


def test_capturable_into_a_hip_graph(M):
    """The C-ABI call is only kernel launches on the caller's stream (device attributes and
    residency are queried once, outside capture), so it can be captured into a hipGraph and
    replayed -- e.g. to amortise launch overhead in a solver loop over a small matrix."""
    rng = np.random.default_rng(21)
    for rows, dtype in ((3000, np.float32), (400000, np.float64)):
        lens = rng.integers(0, 12, rows)
        csr = random_csr(rng, rows, rows, lens, dtype)
        x1 = rng.uniform(-1, 1, rows).astype(dtype); x2 = rng.uniform(-1, 1, rows).astype(dtype)
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        val, off, col = dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices)
        x = dev(x1.copy()); y = torch.zeros(rows, dtype=tdt, device="cuda")
        ws = M.CsrMVWorkspace(rows, csr.nnz, tdt)
        M.csrmv(val, off, col, x, y=y, workspace=ws)                 # warm-up outside capture (one-time queries)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            M.csrmv(val, off, col, x, y=y, workspace=ws)
        for xin in (x1, x2):
            x.copy_(dev(xin)); y.fill_(float("nan"))
            g.replay(); torch.cuda.synchronize()
            check_strict(M, csr, xin, y.cpu().numpy())


def test_prepared_calls_skip_the_coordinate_pass_and_match_bitwise(M, capfd):
    """mspmv_csrmv_prepare + mspmv_csrmv_prepared_*: the tile coordinates are computed once; results are
    bitwise those of the stateless call, for plain and alpha/beta forms, large (ONE launch either way: the prepared one has
    no search) and small problems."""
    rng = np.random.default_rng(21)
    for dtype, rows, hi in ((np.float32, 300000, 60), (np.float64, 300000, 40), (np.float32, 3000, 30)):
        csr = random_csr(rng, rows, rows, rng.integers(0, hi, rows), dtype)
        x = rng.uniform(-1, 1, rows).astype(dtype)
        d = [dev(a) for a in (csr.values, csr.row_offsets, csr.column_indices, x)]
        y_ref = M.csrmv(*d)
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        ws = M.CsrMVWorkspace(csr.rows, csr.nnz, tdt).prepare(d[1])
        capfd.readouterr()
        y = M.csrmv(*d, workspace=ws, debug_synchronous=True)
        log = capfd.readouterr().out
        assert "coords_scatter_kernel" not in log and "search_kernel" not in log and "fixup" not in log, log
        assert torch.equal(y, y_ref)
        for _ in range(3):                                  # the coordinates survive the calls
            assert torch.equal(M.csrmv(*d, workspace=ws), y_ref)
        assert torch.equal(M.csrmv(*d, workspace=M.CsrMVWorkspace(csr.rows, csr.nnz, tdt)), y_ref)   # stateless path still fine
        y0 = dev(rng.uniform(-1, 1, rows).astype(dtype))
        a = M.csrmv(*d, y=y0.clone(), alpha=-0.5, beta=2.0)
        b = M.csrmv(*d, y=y0.clone(), alpha=-0.5, beta=2.0, workspace=ws)
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------
# SpMM extension: every column of Y must satisfy the CsrMV tolerance against the
# oracle's sequential fp64-accumulated sum of that column.
# ---------------------------------------------------------------------------
def _check_csrmm(M, csr, X, Y, alpha=1.0, beta=0.0, Y0=None):
    for c in range(X.shape[1]):
        g, s = O.spmv_gold_acc64(csr, np.ascontiguousarray(X[:, c]))
        want = alpha * g + (beta * Y0[:, c].astype(np.float64) if beta != 0.0 else 0.0)
        eps = 2.0 ** -24 if csr.values.dtype == np.float32 else 2.0 ** -53
        lens = np.diff(csr.row_offsets.astype(np.int64))
        # 8 products per thread in the pack kernel's tile; a slot of the slot form (groups of 8 / 16 columns, large matrices) adds up to
        # 2816 / 32 = 88 products one after the other, then the pieces of a row meet slot by slot (up to 64 per tile)
        cc = 2.0 * (np.ceil(np.log2(lens + 1)) + 8 + 8) + np.minimum(lens, 160)
        tol = cc * eps * (abs(alpha) * s + (abs(beta) * np.abs(Y0[:, c]) if beta != 0.0 else 0.0)) + (0 if alpha == 1.0 and beta == 0.0 else 4 * eps * np.abs(want))
        got = Y[:, c].astype(np.float64)
        bad = np.abs(got - want) > tol
        assert not bad.any(), (c, int(bad.sum()), float(np.abs(got - want).max()))
        if alpha == 1.0 and beta == 0.0:
            assert np.all(got[lens == 0] == 0.0)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 13, 16, 23, 33])
def test_csrmm_matches_the_oracle_column_by_column(M, prec, k):
    dtype, vb = DT[prec]
    rng = np.random.default_rng(100 + k)
    lens = np.minimum((rng.pareto(1.2, 30000) * 3).astype(np.int64), 20000)
    lens[123] = 60000; lens[5000:5600] = 0
    csr = random_csr(rng, 30000, 20000, lens, dtype)
    X = rng.uniform(-1, 1, size=(csr.cols, k)).astype(dtype)
    Y = M.csrmm(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(X))
    _check_csrmm(M, csr, X, Y.cpu().numpy())
    # reproducible bit for bit
    Y2 = M.csrmm(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(X))
    assert torch.equal(Y, Y2)
    # alpha/beta, and a padded leading dimension (views into wider buffers: unaligned packs)
    Xw = torch.zeros(csr.cols, k + 3, dtype=Y.dtype, device="cuda"); Xv = Xw[:, 1:1 + k]; Xv.copy_(dev(X))
    Y0 = rng.uniform(-1, 1, size=(csr.rows, k)).astype(dtype)
    Yw = torch.zeros(csr.rows, k + 2, dtype=Y.dtype, device="cuda"); Yv = Yw[:, 2:2 + k]; Yv.copy_(dev(Y0))
    M.csrmm(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), Xv, Y=Yv, alpha=-1.5, beta=0.5)
    _check_csrmm(M, csr, X, Yv.cpu().numpy(), alpha=-1.5, beta=0.5, Y0=Y0)
    assert float(Yw[:, :2].abs().max()) == 0.0                      # nothing outside the view was written


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_csrmm_slot_form_on_large_matrices(M, prec):
    """Groups of 8 / 16 right-hand sides of matrices with >= 8 M path items run the slot form (spmm_lane_kernel: four or eight lanes
    per nonzero share, rows written straight from registers, a share's first / last row pieces meeting through LDS).  A matrix with
    everything that form distinguishes -- runs of empty rows, rows of one nonzero, rows longer than a share, longer than a tile,
    a giant row over hundreds of tiles, row ends on share and tile boundaries -- for k = 8, 16, 24 and 13 (8 + 4 + 1: the packs
    beside it), unaligned views, alpha / beta; every column against the oracle, bit for bit reproducible."""
    dtype, vb = DT[prec]
    rng = np.random.default_rng(2024)
    rows = 260000
    lens = np.minimum((rng.pareto(1.1, rows) * 6).astype(np.int64), 30000)
    lens[1000:9000] = 0                                  # a run of empty rows across several tiles
    lens[20000:20400] = 44                               # rows of exactly one share (fp32 k = 16), back to back
    lens[30000:30100] = 2816                             # rows of exactly one tile's items
    lens[40000] = 1_500_000                              # a giant row: ~530 tiles
    lens[40001:40050] = 0
    lens[50000:90000:7] = 1
    csr = random_csr(rng, rows, 70000, lens, dtype)
    assert csr.rows + csr.nnz >= (8 << 20)
    vals, offs, cols = dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices)
    for k in (8, 16, 24, 13):
        X = rng.uniform(-1, 1, size=(csr.cols, k)).astype(dtype)
        Y = M.csrmm(vals, offs, cols, dev(X))
        _check_csrmm(M, csr, X, Y.cpu().numpy())
        assert torch.equal(Y, M.csrmm(vals, offs, cols, dev(X)))
    k = 16
    X = rng.uniform(-1, 1, size=(csr.cols, k)).astype(dtype)
    Xw = torch.zeros(csr.cols, k + 3, dtype=vals.dtype, device="cuda"); Xv = Xw[:, 1:1 + k]; Xv.copy_(dev(X))
    Y0 = rng.uniform(-1, 1, size=(csr.rows, k)).astype(dtype)
    Yw = torch.zeros(csr.rows, k + 2, dtype=vals.dtype, device="cuda"); Yv = Yw[:, 2:2 + k]; Yv.copy_(dev(Y0))
    M.csrmm(vals, offs, cols, Xv, Y=Yv, alpha=-1.5, beta=0.5)
    _check_csrmm(M, csr, X, Yv.cpu().numpy(), alpha=-1.5, beta=0.5, Y0=Y0)
    assert float(Yw[:, :2].abs().max()) == 0.0                      # nothing outside the view was written


@pytest.mark.parametrize("shape", ["all_empty", "leading_trailing_empty", "one_giant_row", "giant_row_between_empties", "single_col", "single_tile", "one_row_one_nnz", "exact_tile_multiple"])
def test_csrmm_degenerate_shapes(M, shape):
    rng = np.random.default_rng(7)
    rows, cols, lens = SHAPES[shape](rng)
    for dtype in (np.float32, np.float64):
        csr = random_csr(rng, rows, cols, np.asarray(lens, np.int64), dtype)
        X = rng.uniform(-1, 1, size=(cols, 4)).astype(dtype)
        Y = M.csrmm(dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(X))
        assert not torch.isnan(Y).any()
        _check_csrmm(M, csr, X, Y.cpu().numpy())


def test_randomized_differential_smoke(M):
    """A few seconds of tools/fuzz.py (random shapes x precisions x alignments x tuning flags x
    CsrMV / axpby / prepared / SpMM) -- the long runs are done by hand and noted in DESIGN.md."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), "6", "11"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all within tolerance" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---------------------------------------------------------------------------
# The GPU driver binary with the reference's command line: same report
# lines, same --quiet CSV, PASS under the reference's rule and under the strict check.
# ---------------------------------------------------------------------------
def _gpu_spmv(*args, timeout=300):
    import subprocess
    exe = os.path.join(ROOT, "merge_spmv_amd", "gpu_spmv")
    if not os.path.exists(exe):                  # normally built by __graft_entry__.build(); hipcc is on the GPU box too
        subprocess.run(["make", "-C", os.path.join(ROOT, "merge_spmv_amd"), "gpu_spmv"], check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_gpu_driver_reports_like_the_reference(M):
    out = _gpu_spmv("--grid2d=300", "--i=20")
    assert "grid2d_300, " in out and "20 timing iterations" in out and "num_rows: 90000" in out
    for method in ("Merge-based CsrMV, ", "rocSPARSE CsrMV, ", "rocSPARSE HybMV, "):
        assert method in out
    assert out.count("\tPASS") == 3 and "FAIL" not in out
    assert out.count("strict check: PASS") == 3
    import re
    assert re.search(r"fp64: [0-9.]+ setup ms, [0-9.]+ avg ms, [0-9.]+ gflops, [0-9.]+ effective GB/s \([0-9.]+% peak\)", out)
    # fp32, alpha/beta honoured by the merge kernel and by the gold (the reference parses them but computes with 1/0)
    out = _gpu_spmv("--grid3d=40", "--fp32", "--alpha=2.5", "--beta=-0.5", "--i=5", "--no-vendor", "--prepared")
    assert "fp32:" in out and out.count("\tPASS") == 2 and "FAIL" not in out and "prepared: coordinates found once" in out


def test_gpu_driver_quiet_csv_and_matrix_market_input(M):
    mtx = os.path.join(ROOT, "tests/golden/mtx/giant_row.mtx")
    line = _gpu_spmv("--quiet", "--mtx=" + mtx, "--i=3").strip()
    f = [s.strip() for s in line.split(",")]
    # file, 7 statistics, device, precision, then (method, setup_ms, avg_ms, gflops, GB/s) x 3 (eval_csrmv.sh:8, gpu_spmv.cu:467-471,532-534)
    assert f[0] == mtx and f[9] == "fp64" and len(f) >= 10 + 3 * 5
    assert f[10] == "Merge-based CsrMV" and f[15] == "rocSPARSE CsrMV" and f[20] == "rocSPARSE HybMV"
    for i in (11, 12, 13, 14, 16, 17, 18, 19, 21, 22, 23, 24):
        float(f[i])
    # symmetric file: mirrored entries (sparse_matrix.h:362-368) -> verified against the gold inside the driver
    out = _gpu_spmv("--mtx=" + os.path.join(ROOT, "tests/golden/mtx/symmetric.mtx"), "--i=2", "--no-vendor")
    assert "\tPASS" in out and "FAIL" not in out


def test_gpu_driver_plan_and_multi_gpu_method_lines(M):
    """extra method lines of this project (non-quiet only): --plan = the prepared band-major plan with its set-up in the
    `setup ms` column; --gpus=G = the matrix merge-partitioned over G parts through the C multi-GPU operator, here with
    all parts on one device, peer exchange and the RCCL backend (one rank); --timing prints the ingest phases"""
    out = _gpu_spmv("--grid2d=400", "--fp32", "--i=10", "--no-vendor", "--plan=8", "--gpus=1,3", "--mg-one-device")
    assert "Merge-based CsrMV (prepared band-major plan), " in out and "\t8 column band(s)" in out
    assert "Merge-based CsrMV (1 GPU, all parts on one device), " in out and "Merge-based CsrMV (3 GPUs, all parts on one device), " in out
    assert "carry exchange: peer reads, 12 bytes per step" in out
    assert out.count("\tPASS") == 4 and "FAIL" not in out and out.count("strict check: PASS") == 4
    out = _gpu_spmv("--wheel=20000", "--i=5", "--no-vendor", "--gpus=1", "--mg-exchange=rccl")
    assert "carry exchange: RCCL all-gather, 8 bytes per step" in out and out.count("\tPASS") == 2 and "FAIL" not in out
    out = _gpu_spmv("--mtx=" + os.path.join(ROOT, "tests/golden/mtx/symmetric.mtx"), "--i=2", "--no-vendor", "--timing")
    assert "ingest seconds: read " in out
    # the CSV keeps the reference's columns whatever extras are asked for
    line = _gpu_spmv("--quiet", "--grid2d=100", "--i=3", "--plan", "--gpus=2", "--mg-one-device", "--timing").strip()
    assert "plan" not in line and "GPU" not in line and "ingest" not in line and line.count("Merge-based CsrMV") == 1


def _guarded(t, fill, guard=64):
    """t inside a larger buffer whose surroundings hold `fill`; returns (view, whole buffer)"""
    buf = torch.full((t.numel() + 2 * guard,), fill, dtype=t.dtype, device="cuda")
    buf[guard:guard + t.numel()] = t
    return buf[guard:guard + t.numel()], buf


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("flags", [0, 16, 4, 0xF000010])
def test_nothing_outside_the_arrays_is_used_or_written(M, prec, flags):
    """Every array sits between guard regions: NaN around values and x (a product with anything read from
    there would poison y), a pattern around y and around the temp storage that must survive the call.
    Shapes with ragged array tails (nnz and rows not multiples of 4) on all dispatch paths, CsrMV and SpMM."""
    dtype, vb = DT[prec]
    tdt = torch.float32 if vb == 4 else torch.float64
    rng = np.random.default_rng(5 + flags)
    for rows, cols, hi in ((1001, 333, 9), (40003, 1777, 40), (7, 5, 3)):
        csr = random_csr(rng, rows, cols, rng.integers(0, hi, rows), dtype)
        if csr.nnz == 0:
            continue
        x = rng.uniform(-1, 1, cols).astype(dtype)
        val, _ = _guarded(dev(csr.values), float("nan"))
        col, _ = _guarded(dev(csr.column_indices), 0)
        off, _ = _guarded(dev(csr.row_offsets), 0)
        xv, _ = _guarded(dev(x), float("nan"))
        y, ybuf = _guarded(torch.zeros(rows, dtype=tdt, device="cuda"), 12345.0)
        info = M.launch_info(rows, csr.nnz, vb)
        ws = M.CsrMVWorkspace(rows, csr.nnz, tdt)
        tbuf = torch.full((ws.bytes + 512,), 0xAB, dtype=torch.uint8, device="cuda")
        ws.buffer = tbuf[256:256 + ws.bytes]
        try:
            M.set_tuning(vb, 0, 0, flags)
            M.csrmv(val, off, col, xv, y=y, num_cols=cols, workspace=ws)
        finally:
            M.set_tuning(vb)
        torch.cuda.synchronize()
        assert not torch.isnan(y).any()
        check_strict(M, csr, x, y.cpu().numpy())
        assert bool((ybuf[:64] == 12345.0).all()) and bool((ybuf[-64:] == 12345.0).all())
        assert bool((tbuf[:256] == 0xAB).all()) and bool((tbuf[-256:] == 0xAB).all())
        # SpMM on the same guarded arrays
        k = 5
        X = rng.uniform(-1, 1, (cols, k)).astype(dtype)
        Xv, _ = _guarded(dev(X).reshape(-1), float("nan"))
        Yv, Ybuf = _guarded(torch.zeros(rows * k, dtype=tdt, device="cuda"), 12345.0)
        M.csrmm(val, off, col, Xv.view(cols, k), Y=Yv.view(rows, k))
        torch.cuda.synchronize()
        assert not torch.isnan(Yv).any()
        _check_csrmm(M, csr, X, Yv.view(rows, k).cpu().numpy())
        assert bool((Ybuf[:64] == 12345.0).all()) and bool((Ybuf[-64:] == 12345.0).all())


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("flags", [0, 16, 4])
def test_nan_and_inf_stay_in_their_rows(M, prec, flags):
    """Non-finite products must poison exactly the rows they belong to: every cross-thread / cross-tile
    combination selects instead of multiplying by zero (a NaN * 0 would leak into the neighbours)."""
    dtype, vb = DT[prec]
    rng = np.random.default_rng(17)
    rows = 30000
    lens = rng.integers(0, 12, rows); lens[500] = 9000; lens[20000] = 40000
    csr = random_csr(rng, rows, 5000, lens, dtype)
    x = rng.uniform(-1, 1, 5000).astype(dtype)
    bad_rows = [3, 500, 777, 20000, rows - 1]
    vals = csr.values.copy()
    off = csr.row_offsets
    for i, r in enumerate(bad_rows):
        if off[r + 1] > off[r]:
            vals[off[r] + (off[r + 1] - off[r]) // 2] = np.nan if i % 2 == 0 else np.inf
    poisoned = O.Csr(csr.rows, csr.cols, csr.row_offsets, csr.column_indices, vals)
    try:
        M.set_tuning(vb, 0, 0, flags)
        y_clean, _ = run_gpu(M, csr, x)
        y_bad, _ = run_gpu(M, poisoned, x)
    finally:
        M.set_tuning(vb)
    touched = np.zeros(rows, bool)
    for r in bad_rows:
        if off[r + 1] > off[r]:
            touched[r] = True
            assert not np.isfinite(y_bad[r]), r
    assert np.array_equal(y_bad[~touched], y_clean[~touched])          # bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("cols", [1, 5, 32, 511, 512, 1024, 1025])
def test_tiny_x_is_gathered_from_lds(M, prec, cols):
    """x of at most 4 KB (the reference's --dense=<cols> inputs) is copied to LDS once per block and gathered
    there; the result must be BITWISE what the global-memory gather gives (MSPMV_TUNE_NO_XLDS = 0x80000), on the
    small-problem kernel, the large-problem kernel and with alpha/beta, incl. ragged array tails."""
    dtype, vb = DT[prec]
    rng = np.random.default_rng(cols)
    for rows, hi in ((3001, 9), (50003, 40)):
        csr = random_csr(rng, rows, cols, rng.integers(0, hi, rows), dtype)
        x = rng.uniform(-1, 1, cols).astype(dtype)
        got = {}
        # ("compact": the default for a small problem -- the compact front end gathers even a tiny x from memory;
        #  every other entry runs with it switched off, so that the small-problem kernel's LDS copy is what is compared)
        for flags in ("compact", 0, 0x80000, 16, 16 | 0x80000):
            try:
                M.set_compact_tiles(0 if flags == "compact" else -1)
                M.set_tuning(vb, 0, 0, 0 if flags == "compact" else flags)
                y, ws = run_gpu(M, csr, x)
                y2, _ = run_gpu(M, csr, x, alpha=1.5, beta=0.0)
            finally:
                M.set_tuning(vb); M.set_compact_tiles(0)
            check_strict(M, csr, x, y)
            got[flags] = (y, y2)
        assert np.array_equal(got[0][0], got[0x80000][0]) and np.array_equal(got[0][1], got[0x80000][1])
        assert np.array_equal(got["compact"][0], got[0][0]) and np.array_equal(got["compact"][1], got[0][1])
        assert np.array_equal(got[16][0], got[16 | 0x80000][0]) and np.array_equal(got[16][1], got[16 | 0x80000][1])


@pytest.mark.gpu
def test_config3_through_the_matrix_market_path(M, tmp_path):
    """BASELINE config 3 through the REAL ingest path (tools/c3_ingest.py does this at com-Orkut size: 117 M lines, see
    profiles/r02_c3_ingest.txt): a power-law graph written as a `coordinate pattern symmetric` Matrix Market file,
    `gpu_spmv --mtx=... --cache` run twice (parse + mirror + COO->CSR + image, then the image alone), PASS + strict PASS
    both times, and the CSR the driver built equals the generator's symmetrised CSR array for array."""
    import ctypes
    from merge_spmv_amd import generators as G
    scale, edges = 18, 3_000_000
    n = 1 << scale
    r, c = G.rmat_edges(scale, 0, edges, "cuda", G.SEED_C3)
    off_diag = r != c
    rr = torch.cat([r, c[off_diag]]); cc = torch.cat([c, r[off_diag]])
    order = torch.sort(rr * n + cc, stable=True).indices
    exp_cols = cc[order].to(torch.int32).cpu().numpy()
    exp_off = np.zeros(n + 1, np.int64); np.cumsum(torch.bincount(rr, minlength=n).cpu().numpy(), out=exp_off[1:])
    H = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
    H.mspmv_host_write_pattern_mtx.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_int]
    rh = r.to(torch.int32).cpu().numpy(); ch = c.to(torch.int32).cpu().numpy()
    path = str(tmp_path / "c3.mtx")
    assert H.mspmv_host_write_pattern_mtx(path.encode(), n, n, edges, rh.ctypes.data, ch.ctypes.data, 1) == 0
    first = _gpu_spmv("--mtx=" + path, "--cache", "--timing", "--i=5", "--no-vendor")
    second = _gpu_spmv("--mtx=" + path, "--cache", "--timing", "--i=5", "--no-vendor")
    for out in (first, second):
        assert "\tPASS" in out and "FAIL" not in out and "strict check: PASS" in out and f"num_nonzeros: {exp_cols.size}" in out
    assert "Reading binary CSR image" in second and "Reading binary CSR image" not in first
    with open(path + ".fp64.csrbin", "rb") as f:
        head = f.read(28)
        rows, cols, nnz = np.frombuffer(head[16:28], np.int32)
        off = np.fromfile(f, np.int32, rows + 1); col = np.fromfile(f, np.int32, nnz)
    assert (rows, cols, nnz) == (n, n, exp_cols.size)
    assert np.array_equal(off, exp_off.astype(np.int32)) and np.array_equal(col, exp_cols)


@pytest.mark.gpu
def test_calls_are_capturable_in_a_hip_graph(M):
    """A stateless call is two or three kernel launches and nothing else (no allocation, no synchronisation, no runtime
    query), and so is a plan SpMV: both can be captured into a hipGraph on the caller's stream and replayed with new x."""
    rng = np.random.default_rng(3)
    csr = random_csr(rng, 200000, 50000, rng.integers(0, 12, 200000), np.float32)
    val, off, col = dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices)
    x = torch.zeros(csr.cols, dtype=torch.float32, device="cuda")
    y = torch.zeros(csr.rows, dtype=torch.float32, device="cuda"); yp = torch.zeros_like(y)
    ws = M.CsrMVWorkspace(csr.rows, csr.nnz, torch.float32)
    plan = M.CsrMVPlan(val, off, col, csr.cols, bands=8)
    M.csrmv(val, off, col, x, y=y, num_cols=csr.cols, workspace=ws); plan(x, yp)          # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        M.csrmv(val, off, col, x, y=y, num_cols=csr.cols, workspace=ws)
        plan(x, yp)
    for seed in (1, 2):
        xh = np.random.default_rng(seed).uniform(-1, 1, csr.cols).astype(np.float32)
        x.copy_(torch.from_numpy(xh))
        g.replay(); torch.cuda.synchronize()
        check_strict(M, csr, xh, y.cpu().numpy())
        gold, s = O.spmv_gold_acc64(csr, xh)
        ok, worst = O.strict_check(csr, yp.cpu().numpy(), gold, s, items_per_thread=24)
        assert ok, worst


@pytest.mark.gpu
def test_concurrent_host_threads_on_their_own_streams(M):
    """The library keeps no per-call state: host threads calling it concurrently, each with its own temp storage and
    stream, get bitwise the results of serial calls (ctypes releases the GIL for the duration of a call)."""
    import threading
    rng = np.random.default_rng(9)
    jobs = []
    for i in range(4):
        dtype = np.float32 if i % 2 == 0 else np.float64
        csr = random_csr(rng, 30000 + 7000 * i, 20000, rng.integers(0, 30, 30000 + 7000 * i), dtype)
        xh = rng.uniform(-1, 1, csr.cols).astype(dtype)
        t = (dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(xh))
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        serial = M.csrmv(*t, num_cols=csr.cols).clone()
        jobs.append((csr, t, tdt, serial))
    torch.cuda.synchronize()
    results = [None] * len(jobs)
    def work(i):
        csr, t, tdt, _ = jobs[i]
        stream = torch.cuda.Stream()
        ws = M.CsrMVWorkspace(csr.rows, csr.nnz, tdt)
        y = torch.empty(csr.rows, dtype=tdt, device="cuda")
        for _ in range(50):
            M.csrmv(*t, y=y, num_cols=csr.cols, workspace=ws, stream=stream)
        stream.synchronize()
        results[i] = y
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for th in threads: th.start()
    for th in threads: th.join()
    for (csr, t, tdt, serial), y in zip(jobs, results):
        assert y is not None and torch.equal(y, serial)


def test_a_stateless_call_of_the_small_shape_family_leaves_prepared_coordinates_alone(M):
    """Advisor (round 5, medium): a stateless fp64 call on a --dense=<cols> matrix takes the SMALL tile shape by its column count
    (skinny_rule) while mspmv_csrmv_prepare stores coordinates of the DEFAULT shape -- and the classic pipeline of a prepared call
    (arrays not 16-byte aligned: tile_kernel, PHASE_SKIP_COORDS) trusts what it finds.  Since round 6 the small shape's layout sits
    behind the default one in temp storage and is taken only with aligned arrays, so: prepare -> stateless call (small shape) ->
    prepared call on UNALIGNED copies of the arrays must still read intact coordinates and give the right y; and a prepared call on
    the aligned arrays picks the same shape as the stateless one: bit for bit the same y."""
    rng = np.random.default_rng(77)
    rows, cols = 1_650_000, 5                                     # 9.9 M path items: the family of the rule
    csr = random_csr(rng, rows, cols, np.full(rows, 5, np.int64), np.float64)
    x = rng.uniform(-1, 1, cols)
    assert M.launch_info(rows, csr.nnz, 8, num_cols=cols)["items_per_thread"] == 7 and M.launch_info(rows, csr.nnz, 8)["items_per_thread"] == 11
    val, off, col, xd = dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x)
    ws = M.CsrMVWorkspace(rows, csr.nnz, torch.float64)
    ws.buffer.fill_(0xAB)
    ws.prepare(off)
    coords_before, _, _ = M.debug_read_tiles(ws.buffer, rows, csr.nnz, 8)
    # 1. the stateless call (aligned arrays: the small shape), straight through the C ABI so that it is NOT routed to the prepared entry
    y1 = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    st, _ = M.DeviceSpmv.CsrMV(ws.buffer, ws.bytes, val, off, col, xd, y1, rows, cols, csr.nnz)
    assert st == 0
    torch.cuda.synchronize()
    coords_after, _, _ = M.debug_read_tiles(ws.buffer, rows, csr.nnz, 8)
    assert np.array_equal(coords_before, coords_after)                        # the default layout's coordinates are untouched
    # 2. a prepared call on the aligned arrays: the same rule, the same shape, the same bits
    y2 = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    M.csrmv(val, off, col, xd, y=y2, num_cols=cols, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    # 3. a prepared call on arrays that are NOT 16-byte aligned (values and columns one element into their allocations): the classic
    #    pipeline on the stored coordinates
    val_u = torch.empty(csr.nnz + 1, dtype=torch.float64, device="cuda")[1:]; val_u.copy_(val)
    col_u = torch.empty(csr.nnz + 1, dtype=torch.int32, device="cuda")[1:]; col_u.copy_(col)
    assert val_u.data_ptr() % 16 == 8 and col_u.data_ptr() % 16 == 4
    y3 = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    lib = M.load_library(); size = ctypes.c_size_t(ws.bytes)
    st = lib.mspmv_csrmv_prepared_f64(ctypes.c_void_p(ws.buffer.data_ptr()), ctypes.byref(size), ctypes.c_void_p(val_u.data_ptr()), ctypes.c_void_p(off.data_ptr()),
                                      ctypes.c_void_p(col_u.data_ptr()), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(y3.data_ptr()), rows, cols, csr.nnz,
                                      ctypes.c_double(1.0), ctypes.c_double(0.0), None, 0)
    assert st == 0
    torch.cuda.synchronize()
    gold = O.spmv_gold(csr, x)
    check_strict(M, csr, x, y3.cpu().numpy())
    assert np.allclose(y3.cpu().numpy(), gold, rtol=0, atol=1e-13) and np.array_equal(y1.cpu().numpy().view(np.uint64), gold.view(np.uint64))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_matrices_whose_rows_are_all_long_run_the_classic_launches(M, prec):
    """mspmv_api.hip: long_rows_rule -- from an average of 384 nonzeros per row at >= 8 M path items (240 at >= 16 M) nearly every tile
    would publish and take a tagged record, each a round trip through the memory side at the end of a block's life; such calls run
    the coordinate pass + tiles with one carry each + one fix-up launch instead (tools/long_rows_probe.py: 25-40 % faster).  The host
    decides from rows and nonzeros alone; both sides of the threshold give every row within the strict bound, and a matrix with a FEW
    long rows among short ones keeps the one launch."""
    dtype, vb = DT[prec]
    rng = np.random.default_rng(5 + vb)
    cases = []
    for rows, k, classic in ((20_000, 400, True), (40_000, 190, False), (30_000, 256, False), (66_000, 250, True)):
        info = M.launch_info(rows, rows * k, vb)
        assert (info["snap_head_max"] == 0) == classic and (info["fixup_levels"] >= 1) == classic, (rows, k, info)
        cases.append((rows, k))
    assert M.launch_info(1 << 24, 67_112_959, vb)["snap_head_max"] > 0                    # BASELINE config 4: average 4, one giant row
    for rows, k in cases[:2]:                                                                # (one of each side, run on the device)
        lens = np.full(rows, k, np.int64); lens[::3] += rng.integers(0, 40, lens[::3].size)
        csr = random_csr(rng, rows, 30_000, lens, dtype)
        x = rng.uniform(-1, 1, csr.cols).astype(dtype)
        y, ws = run_gpu(M, csr, x)
        check_strict(M, csr, x, y)
        y2, _ = run_gpu(M, csr, x)
        assert np.array_equal(y.view(np.uint8), y2.view(np.uint8))                          # bitwise repeatable either way
