"""The corpus sweep's tooling (CPU): tools/make_corpus.py lists >= 150 seeded stand-ins over the paper's axes and writes files the product's
Matrix Market ingest reads; tools/eval_csrmv.sh (the reference's eval_csrmv.sh:8-17 contract) turns a directory of them into one CSV line
per file; tools/corpus_summary.py reduces a sweep's CSV to per-decade harmonic means, roofline fractions and the list of files behind the
vendor column.  The GPU sweep itself is tools/corpus_sweep.sh (profiles/r06_corpus_*)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
torch = pytest.importorskip("torch")


def test_the_corpus_spans_the_axes():
    import make_corpus as MC
    items = MC.corpus()
    names = [t[0] for t in items]
    assert len(items) >= 150 and len(set(names)) == len(names)
    assert min(t[3] for t in items) == 10_000 and max(t[3] for t in items) == 200_000_000
    fams = {t[1] for t in items}
    assert {"grid2d", "grid3d", "band", "femblock", "uniform", "rmat", "rmatsym", "circuit", "pareto", "wheel", "dense", "degenerate"} <= fams
    assert items == sorted(items, key=lambda t: (t[3], t[0]))                 # small files first: a chunk of the sweep holds many of them


def test_a_chunk_is_written_read_and_swept(tmp_path):
    d = tmp_path / "corpus"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_corpus.py"), "--dir", str(d), "--from", "0", "--budget-nnz", "2.3e5", "--device", "cpu"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    nxt = int(r.stdout.strip().splitlines()[-1].split()[1])
    files = sorted(os.listdir(d))
    assert nxt == len(files) == 23                                            # every family at 10 k nonzeros
    for f in files:
        head = open(d / f).read(400).splitlines()
        assert head[0].startswith("%%MatrixMarket matrix coordinate") and "STAND-IN written by tools/make_standin_mtx.py" in head[1], f
    # the reference's sweep contract over the directory, with the CPU driver (eval_csrmv.sh:8-17): header + one line per file
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "eval_csrmv.sh"), str(d), "cpu_spmv", "--i=3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert lines[0].startswith("file, num_rows, num_cols, num_nonzeros, row_length_mean") and len(lines) == 1 + len(files)
    cvs = {}
    for l in lines[1:]:
        f = [t.strip() for t in l.split(",")]
        assert int(f[3]) > 4000 and "Merge CsrMV" in l
        cvs[os.path.basename(f[0])] = float(f[6])
    # the row-length variation axis: stencils ~ 0, the wheel / arrow / degenerate shapes far beyond the paper's 60
    assert cvs["009_grid2d_10k.mtx"] < 0.1 and cvs["020_uniform8_10k.mtx"] == 0 and cvs["022_wheel_10k.mtx"] > 15 and cvs["004_degenerate_10k.mtx"] > 15


def test_summary_of_a_sweep(tmp_path):
    import corpus_summary as CS
    hdr = "file, num_rows, num_cols, num_nonzeros, row_length_mean, row_length_std_dev, row_length_variation, row_length_skewness, method_name, setup_ms, avg_spmv_ms, gflops, effective_GBs\n"
    rows = [("a_10M.mtx", 1000000, 1000000, 10000000, 10.0, 1.0, 0.1, 0.0, 0.02, 0.03),
            ("b_10M.mtx", 500000, 500000, 10000000, 20.0, 400.0, 20.0, 5.0, 0.05, 0.04),
            ("c_10k.mtx", 1000, 1000, 10000, 10.0, 1.0, 0.1, 0.0, 0.004, 0.003)]
    with open(tmp_path / "corpus_fp64.csv", "w") as f:
        f.write(hdr)
        for name, r, c, n, mean, sd, cv, sk, ours, ven in rows:
            f.write(f"/x/{name}, {r}, {c}, {n}, {mean}, {sd}, {cv}, {sk}, gfx950, fp64, Merge-based CsrMV, 0.0, {ours}, {2e-6 * n / ours}, 1.0, rocSPARSE CsrMV, 1.5, {ven}, {2e-6 * n / ven}, 1.0, \n")
    (tmp_path / "corpus_checks.txt").write_text("strict-check, /x/a_10M.mtx, fp64, PASS, 0, 0.04\nstrict-check, /x/b_10M.mtx, fp64, FAIL, 3, 1.7\n")
    recs = CS.parse(str(tmp_path / "corpus_fp64.csv"), 8)
    assert len(recs) == 3 and recs[0]["methods"]["rocSPARSE CsrMV"]["ms"] == 0.03
    b_alg = 10000000 * 12 + 1000001 * 4 + 2 * 1000000 * 8
    assert abs(recs[0]["frac"] - b_alg / 0.02e-3 / 8e12) < 1e-9
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "corpus_summary.py"), str(tmp_path)], capture_output=True, text=True, check=True).stdout
    assert "3 files" in out and "below 0.9 x the vendor's rate on 2" in out and "b_10M.mtx" in out and "1 of 2 PASS" in out and "FAIL strict-check" in out
