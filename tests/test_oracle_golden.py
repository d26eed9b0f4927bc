"""Pins the oracle (oracle/merge_oracle.c + oracle/oracle.py) against the
golden vectors that oracle/make_golden.py produced from the reference's own
code, and against the reference's doc-comment known answer
(cub/device/device_spmv.cuh:90-123).  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from conftest import ROOT, load_golden

CASES = load_golden("matrices.json")["cases"]
IDS = [c["label"] for c in CASES]


def build(case, dtype):
    args = list(case["args"])
    if case["kind"] == "mtx":
        args = [os.path.join(ROOT, args[0])]
    return O.make(case["kind"], *args, dtype=dtype)


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_csr_layout_matches_reference(case, prec):
    """sparse_matrix.h generators / InitMarket / CsrMatrix::Init."""
    dtype = np.float32 if prec == "f32" else np.float64
    csr = build(case, dtype)
    assert (csr.rows, csr.cols, csr.nnz) == (case["rows"], case["cols"], case["nnz"])
    assert csr.row_offsets.tolist() == case["row_offsets"]
    assert csr.column_indices.tolist() == case["column_indices"]
    ref_vals = np.asarray(case[prec]["values"], dtype=np.float64).astype(dtype)
    assert np.array_equal(csr.values, ref_vals)
    assert O.stats_csv(csr) == case[prec]["stats_csv"]
    assert O.histogram_text(csr) == case["histogram"]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_merge_path_search_matches_reference(case):
    """Every diagonal 0..rows+nnz+3 vs cub::MergePathSearch."""
    csr = build(case, np.float64)
    for d, (x, y) in enumerate(case["merge_path"]):
        assert O.merge_path_search(d, csr.row_end_offsets, csr.rows, csr.nnz) == (x, y), d
        off64 = csr.row_end_offsets.astype(np.int64)
        assert O.merge_path_search_i64(d, off64, csr.rows, csr.nnz) == (x, y), d


def test_known_answer_device_spmv(golden_kat):
    k = golden_kat
    for dtype in (np.float32, np.float64):
        csr = O.Csr(k["rows"], k["cols"], np.asarray(k["row_offsets"], np.int32),
                    np.asarray(k["column_indices"], np.int32), np.asarray(k["values"], dtype))
        x = np.asarray(k["x"], dtype)
        want = np.asarray(k["y"], dtype)
        assert np.array_equal(O.spmv_gold(csr, x), want)
        for t in (1, 2, 3, 4, 5, 8, 13, 64, 300):
            assert np.array_equal(O.omp_merge_csrmv(csr, x, t), want), t
        for tile in (1, 2, 3, 7, 33, 64):
            y, ck, cv = O.tiled_csrmv(csr, x, tile)
            assert np.array_equal(y, want), tile
            assert ck[-1] == k["rows"] and cv[-1] == 0


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_merge_csrmv_equals_sequential_definition(case):
    """OmpMergeCsrmv restatement == SpmvGold for every thread count when the
    arithmetic is exact (small integers / dyadic values), every row written."""
    rng = np.random.default_rng(7)
    csr = build(case, np.float64)
    x = rng.integers(-4, 5, size=csr.cols).astype(np.float64)
    gold = O.spmv_gold(csr, x)
    for t in (1, 2, 3, 7, 16, 64, 257):
        y = O.omp_merge_csrmv(csr, x, t)
        assert not np.isnan(y).any()
        if np.array_equal(csr.values, np.round(csr.values * 1024) / 1024):
            assert np.array_equal(y, gold), t
        else:
            assert np.allclose(y, gold, rtol=1e-13, atol=1e-13), t
    for tile in (1, 5, 64):
        y, _, _ = O.tiled_csrmv(csr, x, tile)
        assert np.allclose(y, gold, rtol=1e-13, atol=1e-13)


def _expand(v, dtype):
    if isinstance(v, dict):
        a = np.full(v["len"], v["fill"], dtype=dtype)
        a[-1] = v["last"]
        return a
    return np.asarray(v, dtype=dtype)


def test_weak_compare_results_matches_reference(golden_host):
    """utils.h:692-742."""
    for c in golden_host["compare_results"]:
        dtype = np.float32 if c["prec"] == "f32" else np.float64
        got = O.compare_results(_expand(c["computed"], dtype), _expand(c["reference"], dtype))
        assert got == c["verdict"], c


def test_reduce_by_key_fold(golden_host):
    """ReduceByKeyOp<Sum> (thread_operators.cuh:291-301): op(a,b) =
    (b.key, a.key==b.key ? a.v+b.v : b.v); an in-order fold of carry pairs."""
    for c in golden_host["reduce_by_key"]:
        acc = None
        out = []
        for k, v in c["pairs"]:
            acc = (k, v) if acc is None else (k, acc[1] + v if acc[0] == k else v)
            out.append([acc[0], acc[1]])
        assert out == c["inclusive"]


def test_strict_check_behaviour():
    csr = O.make("wheel", 40, dtype=np.float32)
    x = np.ones(csr.cols, np.float32)
    g, s = O.spmv_gold_acc64(csr, x)
    ok, worst = O.strict_check(csr, g.astype(np.float32), g, s)
    assert ok and worst == 0.0
    bad = g.astype(np.float32).copy(); bad[0] += 1.0
    assert not O.strict_check(csr, bad, g, s)[0]
    nan = g.astype(np.float32).copy(); nan[3] = np.nan
    assert not O.strict_check(csr, nan, g, s)[0]


REF_SEARCH = os.path.join(ROOT, "oracle", "_ref", "ref_search")


@pytest.mark.skipif(not os.access(REF_SEARCH, os.X_OK), reason="oracle/_ref/ref_search (the reference's cub::MergePathSearch compiled here by oracle/Makefile) is not present")
@pytest.mark.parametrize("seed", range(6))
def test_oracle_search_against_the_compiled_reference_on_fresh_inputs(tmp_path, seed):
    """Beyond the committed golden vectors: the oracle's MergePathSearch against the REFERENCE's own cub::MergePathSearch
    (cub/thread/thread_search.cuh:53-84, compiled from /root/reference by oracle/Makefile into oracle/_ref/ref_search -- a binary that
    travels with the snapshot, not a source) on freshly drawn row-length patterns: every diagonal of every matrix, clamped ones included.
    Runs wherever the binary is (the authoring container, the GPU box); skipped where it is not."""
    import subprocess
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.integers(1, 400))
    kind = seed % 3
    lens = (rng.integers(0, 6, rows) if kind == 0 else
            np.where(rng.random(rows) < 0.03, rng.integers(50, 900, rows), 0) if kind == 1 else
            np.minimum((rng.pareto(1.0, rows) * 2).astype(np.int64), 500))
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    path = tmp_path / "offsets.txt"
    path.write_text(f"{rows} {nnz}\n" + " ".join(map(str, off)) + "\n")
    out = subprocess.run([REF_SEARCH, str(path)], check=True, capture_output=True, text=True).stdout
    row_end = off[1:].astype(np.int32)
    n = 0
    for line in out.splitlines():
        d, x, y = map(int, line.split())
        assert O.merge_path_search(d, row_end, rows, nnz) == (x, y), (seed, d)
        n += 1
    assert n == rows + nnz + 4
