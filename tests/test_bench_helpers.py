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


def _fake_detail(n_configs=9, long=False):
    pad = "x" * (700 if long else 20)
    roof = {"kernel": "tile_kernel_vec<BAND>", "achieved": 1004.123, "bound": "hbm", "peak": 8000.0, "unit": "GB/s", "frac": 0.1255, "traffic": 4048229006,
            "traffic_over_algorithmic": 4.834, "traffic_src": "replayed:profiles/r05_c2_f32/pmc_latest.json", "algorithmic_bytes_per_launch": 837500004,
            "kernel_ms": 0.83412, "search_ms": 0.0102, "fixup_ms": 0.0067, "launches_timed": 200, "band_passes": 3, "note": pad}
    cfgs = [{"config": f"config number {i} with a long name {pad}", "label": "c", "workload": pad * 3, "data": "synthetic", "dtype": "f64", "rows": 1 << 24, "cols": 1 << 24,
             "nnz": 234366905, "steps": 50, "ms_per_step": 1.40935, "value": 332.6, "unit": "GFLOP/s", "tile": "256x11", "generation_s": 0.3,
             "roofline": dict(roof, **({"ic_resident": True, "frac_ic": 0.9} if i % 2 else {})), "effective_GBs": 2865.2, "effective_pct_of_peak": 35.82,
             "sampled_check": {"rows_checked": 65539, "worst_ratio": 0.0601, "violations": 0, "longest_row": 123, "depth_term": 12}, "y_finite": True,
             "cpu": {"value": 93.1, "cores": 16, "runs": [pad] * 3}, "vendor": {"ms_per_step": 2.2, "note": pad}} for i in range(n_configs)]
    cfgs.append({"config": "one that failed", "error": "OutOfMemoryError: " + pad * 5})
    return {"metric": "CsrMV GFLOP/s", "value": 239.712, "unit": "GFLOP/s", "n_gpus": 1, "steps": 200, "warmup": 20, "ms_per_step": 0.83434, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 uniform CSR: 3125000 x 3125000, 32 nnz/row, 100000000 nnz " + pad, "tile": "256x11", "partition": "single GPU"},
            "effective_GBs_reference_formula": 1468.2, "effective_pct_of_peak": 18.35, "roofline": roof, "sampled_worst_ratio": 0.0293,
            "sampled_check": {"rows_checked": 65539, "worst_ratio": 0.0293, "what": pad * 4},
            "cpu_baseline": {"value": 24.3, "unit": "GFLOP/s", "cores": 16, "kind": "port", "ms": 8.22, "sample": "same C2 matrix: 12 SpMVs after 4 warm-ups",
                             "kernel": "host/merge_csrmv.hpp (OpenMP merge-path)", "runs": [{"variant": pad}] * 3, "cpu_quota": "16 CPUs"},
            "prepared_plan": {"ms_per_step": 0.505, "value": 395.8, "setup_ms": 5.7, "roofline": {"frac": 0.21, "note": pad}, "api": pad},
            "vendor": {"ms_per_step": 1.241, "analysis_ms": 12.2, "library": pad}, "configs": cfgs,
            "per_rank": {"tile_ms_max": 4.1, "tile_ms_min": 3.9, "note": pad * 2}, "exchange": {"exchange": 1, "backend": "rccl all-gather", "carry_bytes_per_step": 64, "fallbacks": []},
            "single_gpu_same_workload": {"n_gpus": 1, "steps": 5, "ms_per_step": 33.0, "value": 121.3}}


@pytest.mark.parametrize("n_configs,long", [(0, False), (8, False), (9, True), (40, True)])
def test_the_final_line_is_short_strict_json_with_the_contract_keys(n_configs, long):
    """bench.compact_line: one line, strictly JSON (no NaN / Infinity tokens), under 4 KB whatever the detail record holds -- the driver's
    parser gave up on the 27 KB line of round 5 (BENCH_r05.json: parsed null) --, carrying the contract's keys, `roofline` and `cpu_baseline`."""
    import json
    detail = bench._finite(_fake_detail(n_configs, long))
    line = bench.compact_line(detail, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line.encode()) < bench.LINE_LIMIT == 4096
    d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    if n_configs == 8:
        assert len(d["configs"]) == 9 and all(set(c) >= {"config", "dtype", "ms_per_step", "value", "frac", "traffic_over_algorithmic", "worst"} or "error" in c for c in d["configs"])
        assert d["detail"] == "gpurun_out/bench_detail.json"


def test_the_config_5_leg_of_an_n_gpu_run_rides_in_the_line():
    """N > 1, default workload: config 2 weak-scaled is the headline (comparable with the N = 1 line), config 5 (strong scaling) a compact
    sub-record `c5_strong`; a failed leg leaves a short error there and the headline intact."""
    import json
    d = _fake_detail(0)
    d["n_gpus"] = 8; d["scaling"] = "weak"
    d["c5_strong"] = dict(_fake_detail(0), n_gpus=8, scaling="strong", dtype="f64", ms_per_step=3.1, value=1290.0)
    line = bench.compact_line(bench._finite(d), "gpurun_out/bench_detail.json")
    assert len(line.encode()) < bench.LINE_LIMIT
    c = json.loads(line)["c5_strong"]
    assert c["n_gpus"] == 8 and c["scaling"] == "strong" and c["value"] == 1290.0 and "frac" in c and "single_gpu_value" in c
    d["c5_strong"] = {"error": "RuntimeError: " + "x" * 500}
    c = json.loads(bench.compact_line(bench._finite(d)))
    assert len(c["c5_strong"]["error"]) <= 120 and c["value"] == d["value"]


def test_nan_and_inf_never_reach_the_line():
    import json
    d = _fake_detail(2)
    d["per_rank"]["exchange_ms_max"] = float("nan"); d["roofline"]["frac_ic"] = float("inf"); d["configs"][0]["sampled_check"]["worst_ratio"] = float("nan")
    line = bench.compact_line(bench._finite(d))
    assert "NaN" not in line and "Infinity" not in line
    json.loads(line)


def test_replayed_traffic_finds_the_committed_counter_passes():
    """roofline.traffic of the default run is replayed from profiles/*/pmc_latest.json (live with --full): the headline's and every config's label
    must resolve to a committed pass of the same workload and precision."""
    for label, dt in (("c2_f32", "f32"), ("c2", "f64"), ("dense5", "f64"), ("circuit", "f64"), ("c3_web", "f64"), ("c3_orkut", "f64"), ("c4", "f32"), ("dense32", "f32"), ("c5", "f64")):
        tr, src = bench.replayed_traffic(label, dt)
        assert tr and tr > 0 and src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src)), (label, dt)
