"""The PRODUCT's C++ host code (merge_spmv_amd/host/*.hpp through
libmspmv_host.so, and the cpu_spmv driver binary) against golden vectors made
by the reference's own sparse_matrix.h / utils.h (tests/golden/*.json).  CPU only."""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

CASES = load_golden("matrices.json")["cases"]
IDS = [c["label"] for c in CASES]
PKG = os.path.join(ROOT, "merge_spmv_amd")


@pytest.fixture(scope="module")
def H():
    path = os.path.join(PKG, "libmspmv_host.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", PKG, "libmspmv_host.so", "cpu_spmv"])
    lib = ctypes.CDLL(path)
    lib.mspmv_host_matrix_create.restype = ctypes.c_void_p
    lib.mspmv_host_matrix_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_int)]
    lib.mspmv_host_matrix_destroy.argtypes = [ctypes.c_void_p]
    lib.mspmv_host_matrix_error.restype = ctypes.c_char_p
    lib.mspmv_host_matrix_error.argtypes = [ctypes.c_void_p]
    lib.mspmv_host_matrix_shape.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.mspmv_host_matrix_copy.argtypes = [ctypes.c_void_p] * 4
    lib.mspmv_host_matrix_text.restype = ctypes.c_char_p
    lib.mspmv_host_matrix_text.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mspmv_host_adaptive_iterations.argtypes = [ctypes.c_longlong, ctypes.c_ulonglong]
    return lib


def create(H, case, fp32):
    kind = case["kind"].encode()
    args = case["args"]
    a = int(args[0]) if case["kind"] != "mtx" else 0
    b = int(args[1]) if len(args) > 1 and case["kind"] != "mtx" else 0
    path = os.path.join(ROOT, args[0]).encode() if case["kind"] == "mtx" else b""
    st = ctypes.c_int()
    h = H.mspmv_host_matrix_create(kind, a, b, path, int(fp32), ctypes.byref(st))
    assert st.value == 0, H.mspmv_host_matrix_error(h)
    return h


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_csr_stats_histogram_match_reference(H, case, prec):
    h = create(H, case, prec == "f32")
    try:
        r, c, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        H.mspmv_host_matrix_shape(h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(n))
        assert (r.value, c.value, n.value) == (case["rows"], case["cols"], case["nnz"])
        off = np.zeros(r.value + 1, np.int32); col = np.zeros(max(n.value, 1), np.int32)
        val = np.zeros(max(n.value, 1), np.float32 if prec == "f32" else np.float64)
        H.mspmv_host_matrix_copy(h, off.ctypes.data, col.ctypes.data, val.ctypes.data)
        assert off.tolist() == case["row_offsets"]
        assert col[: n.value].tolist() == case["column_indices"]
        want = np.asarray(case[prec]["values"], np.float64).astype(val.dtype)
        assert np.array_equal(val[: n.value], want)
        assert H.mspmv_host_matrix_text(h, 0).decode() == case[prec]["stats_csv"]
        assert H.mspmv_host_matrix_text(h, 2).decode() == case["histogram"]
    finally:
        H.mspmv_host_matrix_destroy(h)


def test_weak_comparator_and_cli_match_reference(H, golden_host):
    for c in golden_host["compare_results"]:
        dt = np.float32 if c["prec"] == "f32" else np.float64

        def expand(v):
            if isinstance(v, dict):
                a = np.full(v["len"], v["fill"], dtype=dt); a[-1] = v["last"]; return a
            return np.asarray(v, dtype=dt)
        a, b = expand(c["computed"]), expand(c["reference"])
        fn = H.mspmv_host_compare_reference_rule_f32 if dt == np.float32 else H.mspmv_host_compare_reference_rule_f64
        assert fn(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), a.size) == c["verdict"]
    for c in golden_host["command_line"]:
        argv = [b"prog"] + [a.encode() for a in c["argv"]]
        arr = (ctypes.c_char_p * len(argv))(*argv)
        out = ctypes.create_string_buffer(512)
        H.mspmv_host_parse_args(len(argv), arr, out, 512)
        assert json.loads(out.value.decode()) == c["parsed"]


def test_adaptive_iterations(H):
    """clamp(2^34 / nnz, 100, cap): gpu_spmv.cu:492-493 (cap 50000), cpu_spmv.cpp:611-616 (cap 200000)."""
    assert H.mspmv_host_adaptive_iterations(59524291, 50000) == 288           # circuit5M, README.md:116
    assert H.mspmv_host_adaptive_iterations(100, 50000) == 50000
    assert H.mspmv_host_adaptive_iterations(100, 200000) == 200000
    assert H.mspmv_host_adaptive_iterations(2_000_000_000, 50000) == 100


def run(*args):
    return subprocess.run([os.path.join(PKG, "cpu_spmv"), *args], capture_output=True, text=True, timeout=120)


def test_cpu_driver_quiet_csv_contract(H):
    """eval_csrmv.sh contract: label, 7 stats fields, then per method
    `name, setup_ms, avg_ms, gflops, GB/s` (eval_csrmv.sh:8, cpu_spmv.cpp:515-520)."""
    case = next(c for c in CASES if c["label"] == "grid3d_4")
    r = run("--quiet", "--grid3d=4", "--i=3", "--threads=3")
    assert r.returncode == 0
    line = r.stdout.strip()
    assert line.startswith("grid3d_4, " + case["f64"]["stats_csv"])
    fields = [f.strip() for f in line.split(",")]
    assert fields[8] == "OMP-row CsrMV" and fields[13] == "Merge CsrMV"
    for i in (9, 10, 11, 12, 14, 15, 16, 17):
        float(fields[i])
    assert len(fields) == 19 and fields[18] == ""


@pytest.mark.parametrize("flags", [["--grid2d=40"], ["--grid3d=9", "--fp32"], ["--wheel=3000"], ["--dense=7"],
                                   ["--mtx=" + os.path.join(ROOT, "tests/golden/mtx/giant_row.mtx")]])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_cpu_driver_verifies_itself(H, flags, threads):
    r = run(*flags, "--i=2", f"--threads={threads}")
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("\tPASS\n") == 2 and "FAIL" not in r.stdout
    assert r.stdout.count("strict check: PASS") == 2
    assert f"Using {threads} threads on" in r.stdout
    assert "timing iterations" not in r.stdout           # only printed when adaptive (cpu_spmv.cpp:611-616)


def test_cpu_driver_edge_behaviour(H):
    r = run("--i=2")
    assert r.returncode == 1 and "No graph type specified." in r.stderr
    r = run("--mtx=" + os.path.join(ROOT, "tests/golden/mtx/array.mtx"), "--i=2")
    assert r.returncode == 0 and "OMP-row CsrMV" in r.stdout
    r = run("--help")
    assert r.returncode == 0 and "--mtx=<matrix market file>" in r.stdout


# ---------------------------------------------------------------------------
# the product's Matrix Market reader parses entry lines with
# all OpenMP threads.  It must be indistinguishable from the line-by-line
# reader the reference has (sparse_matrix.h:217-380) -- same matrix, same first
# error -- so: parallel == one-thread == the oracle's restatement (which is
# pinned to the reference header by tests/golden/host_semantics.json).
# ---------------------------------------------------------------------------
def _random_mtx(rng, path, flavour):
    rows, cols = int(rng.integers(1, 60)), int(rng.integers(1, 60))
    sym = flavour in ("symmetric", "skew")
    if sym:
        cols = rows
    n = int(rng.integers(0, 400))
    lines = []
    banner = {"general": "%%MatrixMarket matrix coordinate real general",
              "symmetric": "%%MatrixMarket matrix coordinate real symmetric",
              "skew": "%%MatrixMarket matrix coordinate real skew-symmetric",
              "pattern": "%%MatrixMarket matrix coordinate pattern general",
              "array": "%%MatrixMarket matrix array real general",
              "late_banner": "%%MatrixMarket matrix coordinate real general",
              "crlf": "%%MatrixMarket matrix coordinate real general",
              "bad_row": "%%MatrixMarket matrix coordinate real general",
              "bad_col": "%%MatrixMarket matrix coordinate real general",
              "too_many": "%%MatrixMarket matrix coordinate real general",
              "long_line": "%%MatrixMarket matrix coordinate real general",
              "no_final_newline": "%%MatrixMarket matrix coordinate real general"}[flavour]
    lines.append(banner)
    lines.append("% a comment")
    if flavour == "array":
        lines.append(f"{rows} {cols}")
        for k in range(rows * cols):
            lines.append(f"{rng.uniform(-2, 2):.6g}")
            if rng.random() < 0.05:
                lines.append("% interleaved comment")
    else:
        lines.append(f"{rows}  {cols} {n + (5 if flavour == 'late_banner' else 0)}")
        for k in range(n):
            r, c = int(rng.integers(1, rows + 1)), int(rng.integers(1, cols + 1))
            form = rng.integers(0, 5)
            rs = hex(r) if form == 1 else ("0%o" % r if form == 2 else str(r))       # strtol base 0 (:330)
            if flavour == "pattern" or form == 3:
                lines.append(f"{rs} {c}")                                          # value defaults to 1.0 (:351-355)
            elif form == 4:
                lines.append(f"  {rs}\t{c}   {rng.uniform(-5, 5):.17g}  trailing")
            else:
                lines.append(f"{rs} {c} {rng.uniform(-5, 5):.9g}")
            if rng.random() < 0.03:
                lines.append("% interleaved comment")
        at = int(rng.integers(3, len(lines) + 1))
        if flavour == "late_banner":        # flags are re-evaluated for the lines that follow (:262-268)
            lines.insert(at, "%%MatrixMarket matrix coordinate real skew-symmetric")
        elif flavour == "bad_row":
            lines.insert(at, "x 3 1.0")
        elif flavour == "bad_col":
            lines.insert(at, "3")
        elif flavour == "too_many":
            lines += ["1 1 1.0", "1 1 2.0"]
        elif flavour == "long_line":
            lines.insert(at, "1 1 1.0" + " " * 1100)                              # getline(line, 1024) fails: parsing stops
    eol = "\r\n" if flavour == "crlf" else "\n"
    text = eol.join(lines) + ("" if flavour == "no_final_newline" else eol)
    with open(path, "w", newline="") as f:
        f.write(text)


def _host_market(H, kind, path, fp32):
    st = ctypes.c_int()
    h = H.mspmv_host_matrix_create(kind.encode(), 0, 0, path.encode(), int(fp32), ctypes.byref(st))
    try:
        if st.value != 0:
            return ("error", H.mspmv_host_matrix_error(h).decode())
        r, c, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        H.mspmv_host_matrix_shape(h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(n))
        off = np.zeros(r.value + 1, np.int32); col = np.zeros(max(n.value, 1), np.int32)
        val = np.zeros(max(n.value, 1), np.float32 if fp32 else np.float64)
        H.mspmv_host_matrix_copy(h, off.ctypes.data, col.ctypes.data, val.ctypes.data)
        return ("ok", r.value, c.value, n.value, off.tolist(), col[: n.value].tolist(), val[: n.value].tolist())
    finally:
        H.mspmv_host_matrix_destroy(h)


MTX_FLAVOURS = ["general", "symmetric", "skew", "pattern", "array", "late_banner", "crlf", "bad_row", "bad_col",
                "too_many", "long_line", "no_final_newline"]


@pytest.mark.parametrize("flavour", MTX_FLAVOURS)
def test_parallel_market_reader_is_the_serial_reader(H, tmp_path, flavour):
    from oracle import oracle as O
    rng = np.random.default_rng(MTX_FLAVOURS.index(flavour) + 77)
    for trial in range(6):
        path = str(tmp_path / f"{flavour}_{trial}.mtx")
        _random_mtx(rng, path, flavour)
        for fp32 in (True, False):
            par = _host_market(H, "mtx", path, fp32)
            ser = _host_market(H, "mtx_serial", path, fp32)
            assert par == ser, (flavour, trial)
        # ... and both are the reference algorithm (oracle restatement)
        try:
            want = O.csr_from_coo(*O.coo_market(path))
        except O.MarketError as e:
            assert par[0] == "error" and str(e).split(" at edge")[0].split(":")[0] in par[1], (par, str(e))
            continue
        except (IndexError, ValueError):
            # entries outside the declared shape: the product rejects them, the reference would
            # index out of bounds (sparse_matrix.h:676-728 has no check)
            assert par[0] == "error"
            continue
        if par[0] == "error":
            assert "out of range" in par[1], par[1]
            continue
        assert (par[1], par[2], par[3]) == (want.rows, want.cols, want.nnz)
        assert par[4] == want.row_offsets.tolist() and par[5] == want.column_indices.tolist()
        assert np.array_equal(np.asarray(par[6], np.float64), want.values.astype(np.float64))


def test_market_index_fields_follow_strtol_base_0(H, tmp_path):
    """The reader parses plain decimal indices itself and hands everything else to strtol(., ., 0) as the reference does
    (sparse_matrix.h:330-345): signs, hex, octal, ten digits, form feeds and digits glued to text must come out the same,
    and a value field made of blanks is a missing value (default 1.0, :351-355)."""
    from oracle import oracle as O
    lines = ["%%MatrixMarket matrix coordinate real general", "40 40 13",
             "3 4 1.5", "+7 3 2.5", "0x10 0X1f -1", "010 7 3", "\t 12\t 13\t4e0", "  9   9   ", "9 8 \t", "1 2\r",
             "\x0c5 6 7", "0000000005 3 1", "3 21.5", "1 1", "40 40 -0.25"]
    path = str(tmp_path / "strtol.mtx")
    with open(path, "w", newline="") as f:
        f.write("\n".join(lines) + "\n")
    want = O.csr_from_coo(*O.coo_market(path))
    for fp32 in (True, False):
        par = _host_market(H, "mtx", path, fp32)
        ser = _host_market(H, "mtx_serial", path, fp32)
        assert par == ser and par[0] == "ok", par
        assert (par[1], par[2], par[3]) == (want.rows, want.cols, want.nnz)
        assert par[4] == want.row_offsets.tolist() and par[5] == want.column_indices.tolist()
        assert np.array_equal(np.asarray(par[6], np.float64), want.values.astype(np.float32 if fp32 else np.float64).astype(np.float64))
    # "17abc 3 9": strtol stops at 'a', the column parse then fails on "abc" -> the reference's "badly formed col" exit
    bad = str(tmp_path / "bad.mtx")
    with open(bad, "w", newline="") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n5 5 1\n1abc 2 3\n")
    res = _host_market(H, "mtx", bad, False)
    assert res[0] == "error" and "badly formed col" in res[1], res


def test_parallel_market_reader_throughput(H, tmp_path):
    """not a pass/fail speed gate (CI hosts vary) -- checks a 1M-entry file parses identically with all
    threads and prints both times for the record"""
    import time
    rng = np.random.default_rng(5)
    n, rows = 1_000_000, 200_000
    r = rng.integers(1, rows + 1, n); c = rng.integers(1, rows + 1, n); v = rng.uniform(-1, 1, n)
    path = str(tmp_path / "big.mtx")
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (rows, rows, n))
        f.write("\n".join("%d %d %.9g" % t for t in zip(r.tolist(), c.tolist(), v.tolist())) + "\n")
    t0 = time.perf_counter(); par = _host_market(H, "mtx", path, False); t1 = time.perf_counter()
    ser = _host_market(H, "mtx_serial", path, False); t2 = time.perf_counter()
    assert par[0] == "ok" and par == ser
    print(f"\nMatrix Market 1M entries (read + COO->CSR): all threads {t1 - t0:.3f} s, one thread {t2 - t1:.3f} s")


def test_binary_csr_cache_of_the_drivers(tmp_path):
    """--cache: the second run is served from <mtx>.<prec>.csrbin and reports the same
    matrix; a newer .mtx, another precision or a damaged image are not trusted."""
    import shutil, subprocess, time
    exe = os.path.join(ROOT, "merge_spmv_amd", "cpu_spmv")
    src = os.path.join(ROOT, "tests/golden/mtx/giant_row.mtx")
    mtx = str(tmp_path / "m.mtx"); shutil.copy(src, mtx)
    def run(*extra):
        r = subprocess.run([exe, "--mtx=" + mtx, "--i=1", *extra], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        return r.stdout
    def stats(out):                      # name + the 7 statistics fields of the quiet CSV line
        return out.split(", OMP-row CsrMV")[0]
    plain = run("--quiet")
    first = run("--quiet", "--cache")
    img = mtx + ".fp64.csrbin"
    assert os.path.exists(img) and stats(first) == stats(plain)
    assert "Reading binary CSR image" in run("--cache") and stats(run("--quiet", "--cache")) == stats(plain)
    assert not os.path.exists(mtx + ".fp32.csrbin")
    assert "Reading binary CSR image" not in run("--cache", "--fp32")            # other precision: parsed, own image written
    assert os.path.exists(mtx + ".fp32.csrbin")
    # damaged image -> ignored and rewritten
    with open(img, "r+b") as f:
        f.seek(40); f.write(b"\xff\xff\xff\x7f")
    out = run("--cache")
    assert "Reading binary CSR image" not in out and "PASS" in out
    assert "Reading binary CSR image" in run("--cache")
    # the .mtx changes -> the image is stale
    time.sleep(1.1)
    with open(mtx, "a") as f:
        f.write("% touched\n")
    os.utime(mtx, None)
    assert "Reading binary CSR image" not in run("--cache")


def test_eval_csrmv_sweep_and_ingest_timing(tmp_path):
    """tools/eval_csrmv.sh = the reference's eval_csrmv.sh:8-17 for these drivers: its header, then one `--quiet` CSV
    line per Matrix Market file; --timing prints the ingest phases (non-quiet only, so the CSV is untouched)."""
    import subprocess
    out = subprocess.run(["bash", os.path.join(ROOT, "tools/eval_csrmv.sh"), os.path.join(ROOT, "tests/golden/mtx"), "cpu_spmv", "--i=1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == ("file, num_rows, num_cols, num_nonzeros, row_length_mean, row_length_std_dev, row_length_variation, "
                        "row_length_skewness, method_name, setup_ms, avg_spmv_ms, gflops, effective_GBs")
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "tests/golden/mtx")) if f.endswith(".mtx"))
    body = [l for l in lines[1:] if l.strip()]
    assert len(body) >= 4 and all(".mtx, " in l and "Merge CsrMV, " in l for l in body) and len(body) <= len(files)
    exe = os.path.join(ROOT, "merge_spmv_amd", "cpu_spmv")
    src = os.path.join(ROOT, "tests/golden/mtx/giant_row.mtx")
    r = subprocess.run([exe, "--mtx=" + src, "--i=1", "--timing"], capture_output=True, text=True, timeout=120)
    assert "ingest seconds: read " in r.stdout and "COO->CSR" in r.stdout
    q = subprocess.run([exe, "--mtx=" + src, "--i=1", "--timing", "--quiet"], capture_output=True, text=True, timeout=120)
    assert "ingest seconds" not in q.stdout


def test_pattern_mtx_writer_round_trips(H, tmp_path):
    """mspmv_host_write_pattern_mtx (the corpus-scale ingest tool's writer) -> InitMarket gives back the entries,
    mirrored for a symmetric banner"""
    rng = np.random.default_rng(8)
    n = 5000
    r = rng.integers(0, 300, n).astype(np.int32); c = rng.integers(0, 300, n).astype(np.int32)
    H.mspmv_host_write_pattern_mtx.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_int]
    for sym in (0, 1):
        path = str(tmp_path / f"w{sym}.mtx")
        assert H.mspmv_host_write_pattern_mtx(path.encode(), 300, 300, n, r.ctypes.data, c.ctypes.data, sym) == 0
        got = _host_market(H, "mtx", path, False)
        from oracle import oracle as O
        want = O.csr_from_coo(*O.coo_market(path))
        assert got[0] == "ok" and got[4] == want.row_offsets.tolist() and got[5] == want.column_indices.tolist()
        assert got[3] == n + (int((r != c).sum()) if sym else 0)
