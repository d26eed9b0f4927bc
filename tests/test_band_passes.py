"""Column-band passes of the stateless call (include/mspmv.h: mspmv_set_band_passes; csrc: band_detect_block,
run_band_passes).  CPU: argument checking and the automatic policy's preconditions.  GPU: forced passes against the
oracle on every shape family (the passes must be a pure re-association of the same sums), the device-side detector
on matrices it must accept (uniformly spread columns) and refuse (R-MAT, banded), and that automatic calls stay
bitwise equal to what the chosen path alone computes -- through plain, prepared and graph-captured calls."""
import os

import numpy as np
import pytest

from conftest import ROOT
import merge_spmv_amd as M
from oracle import oracle as O

torch = pytest.importorskip("torch")
gpu = pytest.mark.gpu

NO_FUSED = 16


@pytest.fixture(autouse=True)
def _the_passes_not_the_clocked_form(request):
    """Since round 6 the calls the passes are offered to are served by the clock-scheduled one-pass form (csrc/mspmv_tdm.hpp,
    tests/test_tdm.py) unless that is switched off: the GPU tests of THIS file are about the passes (development library)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    for vb in (4, 8):
        M.set_tdm(vb, -1)
    yield
    for vb in (4, 8):
        M.set_tdm(vb, 0)


def test_set_band_passes_argument_checks():
    """(a setter of the development library, include/mspmv_dev.h)"""
    M.use_library("dev")
    lib = M.load_library()
    for vb in (4, 8):
        for ok in (0, -1, -7, 2, 3, 64):
            assert lib.mspmv_set_band_passes(vb, ok) == 0
        for bad in (1, 65, 1000):
            assert lib.mspmv_set_band_passes(vb, bad) == 1
        assert lib.mspmv_set_band_passes(vb, 0) == 0
    assert lib.mspmv_set_band_passes(2, 0) == 1
    assert lib.mspmv_debug_band_windows(None, 10, 10, 4, None, None) == 1


def test_automatic_policy_table():
    """what the host can decide from sizes alone (csrc/mspmv_api.hip: band_passes_for)"""
    assert M.active_library() == "product"
    C2 = (3_125_000, 3_125_000, 100_000_000)
    assert M.band_passes(*C2, 4) == 3 and M.band_passes(*C2, 8) == 4           # 11.9 / 23.8 MiB of x
    mib = lambda m, vb: m * 2**20 // vb
    rows, nnz = 3_000_000, 96_000_000
    assert [M.band_passes(rows, mib(m, 4), nnz, 4) for m in (4, 6, 8, 12, 16, 24, 32, 48)] == [0, 2, 2, 3, 3, 4, 4, 0]
    assert [M.band_passes(rows, mib(m, 8), nnz, 8) for m in (4, 8, 12, 16, 24, 32, 48)] == [0, 2, 2, 3, 4, 4, 0]
    # a pass must be the CSR stream and little else: >= 160 MiB of it (what asking costs a refused matrix must stay small),
    # >= 8 nonzeros per row
    assert M.band_passes(1_000_000, mib(12, 4), 16_000_000, 4) == 0
    assert M.band_passes(1_000_000, mib(12, 4), 24_000_000, 4) == 3
    assert M.band_passes(20_000_000, mib(12, 4), 100_000_000, 4) == 0
    # dense32 (x of 128 bytes), config 5 (x of 512 MB): never
    assert M.band_passes(3_125_000, 32, 100_000_000, 4) == 0
    assert M.band_passes(1 << 26, 1 << 26, 2_000_000_000, 8) == 0
    try:
        M.set_band_passes(4, -1)
        assert M.band_passes(*C2, 4) == 0
        M.set_band_passes(4, 5)
        assert M.band_passes(*C2, 4) == 5 and M.band_passes(1 << 26, 1 << 26, 2_000_000_000, 4) == 5
        assert M.band_passes(1000, 1000, 5000, 4) == 0          # (a forced count needs as many columns per band ... and the large-problem tile shape)
    finally:
        M.set_band_passes(4, 0)


def test_policy_follows_the_l2_it_is_told_about():
    """the band count is a function of x_bytes / (one XCD's L2) and the stream floor of the
    device's total L2 -- queried from the runtime, here overridden through the environment (read once per process, hence the
    subprocesses): half the L2 means the same x needs more bands and the passes are offered from half the x; a part with
    fewer, larger L2s shifts the other way."""
    import subprocess, sys, json
    code = ("import json, merge_spmv_amd as M\n"
            "M.use_library('dev')      # (the environment overrides exist in the development library only)\n"
            "C2 = (3_125_000, 3_125_000, 100_000_000)\n"
            "mib = lambda m, vb: m * 2**20 // vb\n"
            "print(json.dumps({'caches': M.device_caches(), 'c2': [M.band_passes(*C2, 4), M.band_passes(*C2, 8)],\n"
            "  'f32': [M.band_passes(3_000_000, mib(m, 4), 96_000_000, 4) for m in (2, 3, 4, 6, 8, 12, 16, 24, 32, 48)],\n"
            "  'small_stream': M.band_passes(1_000_000, mib(12, 4), 12_000_000, 4)}))\n")
    def run(env_extra):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    base = run({"MSPMV_FAKE_L2_MIB": "4", "MSPMV_FAKE_XCDS": "8"})          # the MI355X figures, spelled out
    assert base["caches"]["l2_bytes_per_xcd"] == 4 << 20 and base["caches"]["xcds"] == 8
    assert base["c2"] == [3, 4] and base["f32"] == [0, 0, 0, 2, 2, 3, 3, 4, 4, 0] and base["small_stream"] == 0
    half = run({"MSPMV_FAKE_L2_MIB": "2", "MSPMV_FAKE_XCDS": "8"})
    assert half["caches"]["l2_bytes_per_xcd"] == 2 << 20
    assert half["f32"] == [0, 2, 2, 3, 3, 4, 4, 0, 0, 0]                         # every threshold at half the x
    assert half["c2"] == [4, 0]                                                  # 11.9 MiB = 5.96 L2; 23.8 MiB fp64 = 11.9 L2: beyond
    assert half["small_stream"] == 4                                             # 92 MiB of stream >= 5 x 8 x 2 MiB now (x = 6 L2)
    big = run({"MSPMV_FAKE_L2_MIB": "16", "MSPMV_FAKE_XCDS": "2"})
    assert big["c2"] == [0, 0] and big["f32"][-3:] == [2, 2, 3]                  # 24 / 32 / 48 MiB = 1.5 / 2 / 3 L2


def test_temp_storage_covers_the_band_bookkeeping():
    # large-problem shape: 64 verdicts + 8 claim counters + one int per tile on top of coordinates and carries
    info = M.launch_info(3_125_000, 100_000_000, 4)
    assert (info["block_threads"], info["items_per_thread"]) == (256, 11)
    tiles = info["num_tiles"]
    assert info["temp_bytes"] >= (tiles + 1) * 8 + tiles * 8 + (64 + 8 * 64) * 4 + tiles * 4


def _random(rng, rows, cols, lens, dtype):
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    col = rng.integers(0, cols, nnz).astype(np.int32)
    return O.Csr(rows, cols, off.astype(np.int32), col, rng.uniform(-1, 1, nnz).astype(dtype))


SHAPES = {
    "short_rows": lambda rng: (60000, 50000, rng.integers(0, 12, 60000)),
    "power_law": lambda rng: (20000, 30000, np.minimum((rng.pareto(1.1, 20000) * 2).astype(np.int64), 20000)),
    "giant_row": lambda rng: (3000, 100000, np.where(np.arange(3000) == 1500, 300000, rng.integers(0, 3, 3000))),
    "mostly_empty": lambda rng: (400000, 7000, np.where(np.arange(400000) % 97 == 0, 50, 0)),
    "all_empty": lambda rng: (5000, 500, np.zeros(5000, np.int64)),
    "few_cols": lambda rng: (50000, 3, rng.integers(0, 3, 50000)),
    "one_tile_and_a_bit": lambda rng: (100, 4000, np.full(100, 30)),
}


@pytest.fixture
def forced_shape():
    """three-pass pipeline on the 256x11 tile whatever the size; policy and tuning restored afterwards"""
    for vb in (4, 8):
        M.set_tuning(vb, 256, 11, NO_FUSED)
    yield
    for vb in (4, 8):
        M.set_tuning(vb)
        M.set_band_passes(vb, 0)


@gpu
@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("passes", [2, 3, 8])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_forced_passes_match_oracle(forced_shape, shape, passes, prec):
    dtype, tdt = (np.float32, torch.float32) if prec == "f32" else (np.float64, torch.float64)
    vb = 4 if prec == "f32" else 8
    rng = np.random.default_rng(len(shape) * 11 + passes)
    rows, cols, lens = SHAPES[shape](rng)
    csr = _random(rng, rows, cols, np.asarray(lens, np.int64), dtype)
    x = rng.uniform(-1, 1, cols).astype(dtype)
    d = lambda a: torch.from_numpy(a).cuda()
    val, off, col, dx = d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x)
    ws = M.CsrMVWorkspace(rows, csr.nnz, tdt)
    M.set_band_passes(vb, passes)
    y = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
    M.csrmv(val, off, col, dx, y=y, num_cols=cols, workspace=ws)
    torch.cuda.synchronize()
    g, s = O.spmv_gold_acc64(csr, x)
    ok, worst = O.strict_check(csr, y.cpu().numpy(), g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=passes))
    assert ok, (shape, passes, prec, worst)
    # rows without nonzeros are exact zeros, whatever the passes add
    empty = np.diff(csr.row_offsets) == 0
    assert not np.any(y.cpu().numpy()[empty])
    # bitwise reproducible (the blocks share the tiles out differently every time; the sums must not care)
    for _ in range(3):
        y2 = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, dx, y=y2, num_cols=cols, workspace=ws)
        assert torch.equal(y, y2)
    # y = alpha A x + beta y: pass 0 applies beta, the later passes only add
    y0 = rng.uniform(-1, 1, rows).astype(dtype)
    y3 = d(y0.copy())
    M.csrmv(val, off, col, dx, y=y3, num_cols=cols, workspace=ws, alpha=-1.5, beta=0.5)
    torch.cuda.synchronize()
    want = -1.5 * g + 0.5 * y0.astype(np.float64)
    eps = np.finfo(dtype).eps / 2
    c = 2.0 * (np.ceil(np.log2(np.diff(csr.row_offsets.astype(np.int64)) + 1)) + 16 + passes + 8)
    tol = c * eps * (1.5 * s + 0.5 * np.abs(y0)) + 4 * eps * np.abs(want)
    assert np.all(np.abs(y3.cpu().numpy().astype(np.float64) - want) <= tol)
    # one band (the policy's "never") is the ordinary call: the passes only re-associate
    M.set_band_passes(vb, -1)
    y4 = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
    M.csrmv(val, off, col, dx, y=y4, num_cols=cols, workspace=ws)
    torch.cuda.synchronize()
    ok4, _ = O.strict_check(csr, y4.cpu().numpy(), g, s)
    assert ok4
    diff = np.abs(y.cpu().numpy().astype(np.float64) - y4.cpu().numpy().astype(np.float64))
    assert np.all(diff <= 2 * (16 + passes) * eps * s + 1e-300)


def _golden_cases():
    from conftest import load_golden
    return load_golden("matrices.json")["cases"]


@gpu
@pytest.mark.parametrize("case", _golden_cases(), ids=[c["label"] for c in _golden_cases()])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_forced_passes_on_the_reference_golden_matrices(forced_shape, case, prec):
    """The reference's own test inputs (its generators and the Matrix Market fixtures) under its own protocol
    (x = 1, compare with SpmvGold by its PASS rule, gpu_spmv.cu:521-525, utils.h:692-742), with three column-band passes
    forced: exact where the arithmetic is exact (y = row length), the tile coordinates and carry keys bit for bit the
    oracle's, the carries -- accumulated over the passes -- within the carry bound."""
    import os
    from conftest import ROOT
    dtype, tdt = (np.float32, torch.float32) if prec == "f32" else (np.float64, torch.float64)
    vb = 4 if prec == "f32" else 8
    args = [os.path.join(ROOT, case["args"][0])] if case["kind"] == "mtx" else case["args"]
    csr = O.make(case["kind"], *args, dtype=dtype)
    # (inputs with fewer than 3 columns, 4 nonzeros or 3 rows take the ordinary path -- band_passes_for, the dword-per-lane
    # kernel -- and must satisfy the same checks)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    M.set_band_passes(vb, 3)
    ws = M.CsrMVWorkspace(csr.rows, csr.nnz, tdt)
    info = M.launch_info(csr.rows, csr.nnz, vb)
    assert info["num_tiles"] >= 1
    x = np.ones(csr.cols, dtype)
    y = torch.full((csr.rows,), float("nan"), dtype=tdt, device="cuda")
    M.csrmv(d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x), y=y, num_cols=csr.cols, workspace=ws)
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    gold = O.spmv_gold(csr, x)
    assert O.compare_results(yh, gold) == 0
    if case["kind"] != "mtx":
        assert np.array_equal(yh, gold)
    g, s = O.spmv_gold_acc64(csr, x)
    ok, worst = O.strict_check(csr, yh, g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=3))
    assert ok, worst
    coords, keys, vals = M.debug_read_tiles(ws.buffer, csr.rows, csr.nnz, vb)
    want = O.tile_coords(csr, info["tile_items"])
    assert np.array_equal(coords, want[: info["num_tiles"] + 1])
    _, ck, cv = O.tiled_csrmv(csr, x, info["tile_items"])
    assert np.array_equal(keys, ck)
    eps = 2.0 ** -23 if vb == 4 else 2.0 ** -52
    scale = float(np.abs(csr.values).max(initial=0)) * info["tile_items"]
    assert np.all(np.abs(vals.astype(np.float64) - cv.astype(np.float64)) <= eps * scale + 1e-300)


@gpu
@pytest.mark.parametrize("shape", ["short_rows", "power_law", "giant_row", "mostly_empty"])
@pytest.mark.parametrize("passes", [2, 5])
def test_forced_passes_on_the_fp64_mid_size_tile(shape, passes):
    """the fp64 256x7 tile (problems of up to 8 M path items; forced here) carries the passes too"""
    rng = np.random.default_rng(len(shape) * 13 + passes)
    rows, cols, lens = SHAPES[shape](rng)
    csr = _random(rng, rows, cols, np.asarray(lens, np.int64), np.float64)
    x = rng.uniform(-1, 1, cols)
    d = lambda a: torch.from_numpy(a).cuda()
    val, off, col, dx = d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x)
    try:
        M.set_tuning(8, 256, 7, NO_FUSED)
        M.set_band_passes(8, passes)
        assert M.band_passes(rows, cols, csr.nnz, 8) == passes
        ws = M.CsrMVWorkspace(rows, csr.nnz, torch.float64)
        y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        M.csrmv(val, off, col, dx, y=y, num_cols=cols, workspace=ws)
        y2 = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        M.csrmv(val, off, col, dx, y=y2, num_cols=cols, workspace=ws)
        torch.cuda.synchronize()
        g, s = O.spmv_gold_acc64(csr, x)
        ok, worst = O.strict_check(csr, y.cpu().numpy(), g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=passes))
        assert ok, (shape, passes, worst)
        assert torch.equal(y, y2)
        M.set_band_passes(8, -1)
        y1 = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        M.csrmv(val, off, col, dx, y=y1, num_cols=cols, workspace=ws)
        assert O.strict_check(csr, y1.cpu().numpy(), g, s)[0]
    finally:
        M.set_tuning(8); M.set_band_passes(8, 0)


@gpu
def test_forced_passes_more_than_columns_fall_back(forced_shape):
    rng = np.random.default_rng(3)
    csr = _random(rng, 30000, 2, rng.integers(0, 4, 30000), np.float32)
    x = rng.uniform(-1, 1, 2).astype(np.float32)
    d = lambda a: torch.from_numpy(a).cuda()
    M.set_band_passes(4, 8)                    # 8 passes over 2 columns: not taken
    y = M.csrmv(d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x), num_cols=2)
    M.set_band_passes(4, -1)
    y2 = M.csrmv(d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x), num_cols=2)
    assert torch.equal(y, y2)


def _uniform(rows, cols, per_row, tdt, seed=5):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    col = torch.randint(0, cols, (rows, per_row), generator=g, device="cuda", dtype=torch.int32).sort(dim=1).values.reshape(-1).contiguous()
    val = (torch.rand(rows * per_row, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    off = (torch.arange(rows + 1, device="cuda", dtype=torch.int64) * per_row).to(torch.int32)
    x = (torch.rand(cols, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    return val, off, col, x


def _banded(n, w, tdt):
    rows = torch.arange(n, device="cuda", dtype=torch.int64)
    start = (rows - w // 2).clamp(0, n - w)
    col = (start[:, None] + torch.arange(w, device="cuda")[None, :]).reshape(-1).to(torch.int32)
    off = (torch.arange(n + 1, device="cuda", dtype=torch.int64) * w).to(torch.int32)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    val = (torch.rand(n * w, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    x = (torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    return val, off, col, x


def _strict_on_device(val, off, col, x, y, extra):
    lens = (off[1:] - off[:-1]).long()
    prod = val.double() * x.double()[col.long()]
    g = torch.segment_reduce(prod, "sum", lengths=lens, unsafe=True)
    s = torch.segment_reduce(prod.abs(), "sum", lengths=lens, unsafe=True)
    eps = 2.0 ** -24 if val.dtype == torch.float32 else 2.0 ** -53
    tol = 2.0 * (torch.ceil(torch.log2(lens.double() + 1)) + 16 + extra) * eps * s
    return bool(((y.double() - g).abs() <= tol).all())


@gpu
@pytest.mark.parametrize("prec,cols,want_passes", [("f32", 2_400_000, 2), ("f32", 4_000_000, 3), ("f64", 1_600_000, 2)])
def test_detector_accepts_uniform_columns(prec, cols, want_passes):
    """34 M uniformly spread nonzeros, x of 9-15 MB: all 64 windows say "spread", and the automatic call computes
    exactly what the forced passes compute -- plain, prepared, and replayed from a captured graph."""
    tdt = torch.float32 if prec == "f32" else torch.float64
    vb = 4 if prec == "f32" else 8
    rows, per_row = 1_062_500, 32
    val, off, col, x = _uniform(rows, cols, per_row, tdt)
    nnz = rows * per_row
    ws = M.CsrMVWorkspace(rows, nnz, tdt)
    try:
        M.set_band_passes(vb, 0)
        y = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=y, num_cols=cols, workspace=ws)
        torch.cuda.synchronize()
        assert int(M.debug_band_windows(ws, rows, nnz, vb).sum()) == 64
        assert _strict_on_device(val, off, col, x, y, want_passes)
        M.set_band_passes(vb, want_passes)
        yf = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=yf, num_cols=cols, workspace=ws)
        assert torch.equal(y, yf), "automatic call != the passes the policy table names"
        M.set_band_passes(vb, -1)
        yn = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=yn, num_cols=cols, workspace=ws)
        assert _strict_on_device(val, off, col, x, yn, 0)
        assert not torch.equal(y, yn), "the passes re-associate: identical bits mean they did not run"
        M.set_band_passes(vb, 0)
        # prepared call: the coordinate pass (which carries the sampling blocks) is skipped, the stand-alone sampler runs
        ws2 = M.CsrMVWorkspace(rows, nnz, tdt).prepare(off)
        yp = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=yp, num_cols=cols, workspace=ws2)
        assert torch.equal(y, yp)
        # graph capture: nothing in the path looks at device data from the host
        stream = torch.cuda.Stream()
        yg = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        with torch.cuda.stream(stream):
            M.csrmv(val, off, col, x, y=yg, num_cols=cols, workspace=ws)      # warm-up outside the capture
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                M.csrmv(val, off, col, x, y=yg, num_cols=cols, workspace=ws)
        for _ in range(3):
            yg.fill_(float("nan"))
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(y, yg)
    finally:
        M.set_band_passes(vb, 0)


@gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_headline_matrix_at_full_size(prec):
    """BASELINE config 2 exactly as bench.py builds it (3 125 000^2, 32 per row, 10^8 nonzeros): the automatic call runs the
    passes (64/64 windows), every row is within the strict bound of the fp64 segment sums, the result is bitwise the one
    of the forced passes and differs from the one-sweep result only by re-association, and it is linear in x."""
    from merge_spmv_amd import generators as G
    tdt = torch.float32 if prec == "f32" else torch.float64
    vb = 4 if prec == "f32" else 8
    A = G.uniform_csr(3_125_000, 3_125_000, 32, dtype=tdt)
    x = G.uniform_pm1(G.SEED_C2 + 2, A.cols, tdt, "cuda")
    offered = M.band_passes(A.rows, A.cols, A.nnz, vb)
    assert offered == (3 if prec == "f32" else 4)
    ws = M.CsrMVWorkspace(A.rows, A.nnz, tdt)
    run = lambda xx: M.csrmv(A.values, A.row_offsets, A.column_indices, xx, num_cols=A.cols, workspace=ws)
    try:
        y = run(x)
        torch.cuda.synchronize()
        assert int(M.debug_band_windows(ws, A.rows, A.nnz, vb).sum()) == 64
        assert _strict_on_device(A.values, A.row_offsets, A.column_indices, x, y, offered)
        M.set_band_passes(vb, offered)
        assert torch.equal(run(x), y)
        M.set_band_passes(vb, -1)
        y1 = run(x)
        assert _strict_on_device(A.values, A.row_offsets, A.column_indices, x, y1, 0)
        eps = 2.0 ** -24 if prec == "f32" else 2.0 ** -53
        # 32 products of magnitude < 1 per row: |sum| and sum|.| below 32
        assert float((y.double() - y1.double()).abs().max()) <= 2 * (16 + offered + 6) * eps * 32
        M.set_band_passes(vb, 0)
        # linearity: A (x + 2 x2) = A x + 2 A x2 up to rounding
        x2 = G.uniform_pm1(G.SEED_C2 + 7, A.cols, tdt, "cuda")
        lhs = run(x + 2 * x2)
        rhs = y + 2 * run(x2)
        assert float((lhs.double() - rhs.double()).abs().max()) <= 8 * (16 + offered + 6) * eps * 32 * 3
    finally:
        M.set_band_passes(vb, 0)


@gpu
def test_band_passes_on_two_streams_at_once():
    """Two SpMVs in band mode in flight together (own workspaces, own streams, one matrix): the blocks that run the passes
    wait for nothing but memory, so sharing the CUs with another launch can only slow them down; results stay bitwise."""
    tdt = torch.float32
    rows, cols, per_row = 1_062_500, 2_400_000, 32
    val, off, col, x = _uniform(rows, cols, per_row, tdt)
    nnz = rows * per_row
    ref = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
    M.csrmv(val, off, col, x, y=ref, num_cols=cols, workspace=M.CsrMVWorkspace(rows, nnz, tdt))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    wss = [M.CsrMVWorkspace(rows, nnz, tdt) for _ in range(2)]
    ys = [[torch.full((rows,), float("nan"), dtype=tdt, device="cuda") for _ in range(6)] for _ in range(2)]
    for it in range(6):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                M.csrmv(val, off, col, x, y=ys[k][it], num_cols=cols, workspace=wss[k])
    torch.cuda.synchronize()
    for k in range(2):
        assert int(M.debug_band_windows(wss[k], rows, nnz, 4).sum()) == 64
        for it in range(6):
            assert torch.equal(ys[k][it], ref), (k, it)


@gpu
@pytest.mark.parametrize("kind", ["rmat", "banded"])
def test_detector_refuses_matrices_with_reuse(kind):
    """R-MAT (hot columns: 60-88 % distinct lines per window) and a banded matrix (a window spans a sliver of x) sit in
    the size class the policy considers, but no window votes for the passes: the result is the ordinary call's, bit for bit."""
    from merge_spmv_amd import generators as G
    tdt = torch.float32
    if kind == "rmat":
        A = G.rmat_csr(21, 40_000_000, dtype=tdt, seed=G.SEED_C3)
        val, off, col, rows, cols = A.values, A.row_offsets, A.column_indices, A.rows, A.cols
        x = G.uniform_pm1(1, cols, tdt, "cuda")
    else:
        rows = cols = 2_500_000
        val, off, col, x = _banded(rows, 15, tdt)
    nnz = int(val.numel())
    assert M.launch_info(rows, nnz, 4)["items_per_thread"] == 11 and nnz * 8 + 4 * rows > 256 << 20 and nnz >= 8 * rows
    ws = M.CsrMVWorkspace(rows, nnz, tdt)
    try:
        M.set_band_passes(4, 0)
        y = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=y, num_cols=cols, workspace=ws)
        torch.cuda.synchronize()
        assert int(M.debug_band_windows(ws, rows, nnz, 4).sum()) == 0
        # with the passes switched off the call is no longer a candidate and runs ONE launch of row-snapped tiles; the
        # refused candidate ran the classic tiles + fix-up: the same sums as the classic pipeline forced by hand, bit for bit
        M.set_band_passes(4, -1)
        M.set_tuning(4, 0, 0, 0x40000000)
        yn = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=yn, num_cols=cols, workspace=ws)
        assert torch.equal(y, yn)
        assert _strict_on_device(val, off, col, x, y, 0)
        M.set_tuning(4)
        ys = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=ys, num_cols=cols, workspace=ws)
        assert _strict_on_device(val, off, col, x, ys, 0)
    finally:
        M.set_band_passes(4, 0); M.set_tuning(4)


@gpu
@pytest.mark.parametrize("uniform_share,expect_passes", [(0.45, False), (0.70, False), (0.95, True)])
def test_detector_with_split_votes(uniform_share, expect_passes):
    """A matrix whose first rows have uniformly spread columns and whose remaining rows are banded: the 64 windows are
    spaced evenly over the nonzeros, so about `uniform_share` of them vote "spread".  Below the 56-of-64 majority the call
    must run the ordinary body (bit for bit the classic pipeline's result); above it the passes (bit for bit the forced
    passes' result); either way every row stays within the strict bound."""
    tdt, vb = torch.float32, 4
    rows, cols, per_row = 1_062_500, 2_400_000, 32
    n_uni = int(rows * uniform_share)
    val_u, _, col_u, x = _uniform(n_uni, cols, per_row, tdt)
    r = torch.arange(rows - n_uni, device="cuda", dtype=torch.int64)
    start = ((r * 2) % (cols - per_row)).clamp(0, cols - per_row)
    col_b = (start[:, None] + torch.arange(per_row, device="cuda")[None, :]).reshape(-1).to(torch.int32)
    g = torch.Generator(device="cuda"); g.manual_seed(13)
    val_b = (torch.rand((rows - n_uni) * per_row, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    val = torch.cat([val_u, val_b]); col = torch.cat([col_u, col_b])
    off = (torch.arange(rows + 1, device="cuda", dtype=torch.int64) * per_row).to(torch.int32)
    nnz = rows * per_row
    ws = M.CsrMVWorkspace(rows, nnz, tdt)
    try:
        M.set_band_passes(vb, 0)
        offered = M.band_passes(rows, cols, nnz, vb)
        assert offered == 2
        y = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=y, num_cols=cols, workspace=ws)
        torch.cuda.synchronize()
        votes = int(M.debug_band_windows(ws, rows, nnz, vb).sum())
        assert abs(votes - 64 * uniform_share) <= 3, votes
        assert (votes >= 56) == expect_passes
        assert _strict_on_device(val, off, col, x, y, offered)
        yr = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        if expect_passes:
            M.set_band_passes(vb, offered)                       # forced passes
        else:
            M.set_band_passes(vb, -1); M.set_tuning(vb, 0, 0, 0x40000000)      # classic tiles + fix-up, no passes on offer
        M.csrmv(val, off, col, x, y=yr, num_cols=cols, workspace=M.CsrMVWorkspace(rows, nnz, tdt))
        assert torch.equal(y, yr)
    finally:
        M.set_band_passes(vb, 0); M.set_tuning(vb)


@gpu
def test_launch_log_of_a_prepared_banded_call(capfd):
    """debug_synchronous names the launches: coordinates with the sampling blocks folded in (no launch of their own),
    the tile kernel, the fix-up; a prepared call shows the stand-alone sampler instead of the coordinate pass."""
    tdt = torch.float32
    rows, cols, per_row = 1_062_500, 2_400_000, 32
    val, off, col, x = _uniform(rows, cols, per_row, tdt)
    nnz = rows * per_row
    vp = lambda t: __import__("ctypes").c_void_p(t.data_ptr())
    import ctypes
    lib = M.load_library()
    size = ctypes.c_size_t(0)
    assert lib.mspmv_csrmv_f32(None, ctypes.byref(size), None, None, None, None, None, rows, cols, nnz, None, 0) == 0
    tmp = torch.empty(size.value, dtype=torch.uint8, device="cuda")
    y = torch.empty(rows, dtype=tdt, device="cuda")
    capfd.readouterr()
    assert lib.mspmv_csrmv_f32(vp(tmp), ctypes.byref(size), vp(val), vp(off), vp(col), vp(x), vp(y), rows, cols, nnz, None, 1) == 0
    names = [l.split("<<<")[0].replace("mspmv: ", "") for l in capfd.readouterr().out.splitlines() if l.startswith("mspmv: ")]
    assert names == ["coords_scatter_kernel", "tile_kernel_vec", "fixup_onepass_kernel"], names
    assert lib.mspmv_csrmv_prepare(vp(tmp), ctypes.byref(size), vp(off), rows, nnz, 4, None, 0) == 0
    capfd.readouterr()
    assert lib.mspmv_csrmv_prepared_f32(vp(tmp), ctypes.byref(size), vp(val), vp(off), vp(col), vp(x), vp(y), rows, cols, nnz,
                                        ctypes.c_float(1.0), ctypes.c_float(0.0), None, 1) == 0
    names = [l.split("<<<")[0].replace("mspmv: ", "") for l in capfd.readouterr().out.splitlines() if l.startswith("mspmv: ")]
    assert names == ["band_detect_kernel", "tile_kernel_vec", "fixup_onepass_kernel"], names
