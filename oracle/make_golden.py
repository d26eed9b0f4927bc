#!/usr/bin/env python3
"""make_golden.py -- TEST INFRASTRUCTURE: generate tests/golden/*.json from the
REFERENCE's own code, in the authoring container (needs /root/reference).

Sources of truth executed here (built by `make -C oracle ref`):
  oracle/_ref/ref_host    = reference sparse_matrix.h + utils.h (g++)
  oracle/_ref/ref_search  = reference cub::MergePathSearch + ReduceByKeyOp (hipcc host)
Fixtures are DATA ONLY: inputs and the outputs the reference produced.  The
small .mtx inputs under tests/golden/mtx/ are our own files.

Run:  make -C oracle ref && python oracle/make_golden.py
"""
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_HOST = os.path.join(HERE, "_ref", "ref_host")
REF_SEARCH = os.path.join(HERE, "_ref", "ref_search")
GOLD = os.path.join(ROOT, "tests", "golden")

GENERATED = [
    ("dense", [4, 5]), ("dense", [1, 7]), ("dense", [33, 3]), ("dense", [6, 1]),
    ("grid2d", [3]), ("grid2d", [2]), ("grid2d", [7]),
    ("grid3d", [2]), ("grid3d", [3]), ("grid3d", [4]),
    ("wheel", [1]), ("wheel", [5]), ("wheel", [40]),
]
MTX = ["general_dups", "symmetric", "skew", "pattern", "array", "unterminated",
       "hexidx_crlf", "giant_row"]


def run(*cmd):
    return subprocess.run(list(cmd), check=True, capture_output=True, text=True).stdout


def coords_for(row_offsets, rows, nnz):
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(f"{rows} {nnz}\n" + " ".join(map(str, row_offsets)) + "\n")
        name = f.name
    try:
        out = run(REF_SEARCH, name)
    finally:
        os.unlink(name)
    return [[int(t) for t in line.split()[1:]] for line in out.strip().splitlines()]


def matrix_case(kind, args, label):
    case = {"label": label, "kind": kind, "args": args}
    for prec in ("f32", "f64"):
        j = json.loads(run(REF_HOST, "csr", prec, kind, *map(str, args)))
        case[prec] = {"values": j["values"], "stats_csv": j["stats_csv"]}
        case.update(rows=j["rows"], cols=j["cols"], nnz=j["nnz"],
                    row_offsets=j["row_offsets"], column_indices=j["column_indices"],
                    stats=j["stats"])
        case["histogram"] = run(REF_HOST, "hist", prec, kind, *map(str, args))
    # every diagonal 0 .. rows+nnz+3 -> (x, y) from the reference's MergePathSearch
    case["merge_path"] = coords_for(case["row_offsets"], case["rows"], case["nnz"])
    return case


def main():
    os.makedirs(GOLD, exist_ok=True)
    cases = []
    for kind, args in GENERATED:
        cases.append(matrix_case(kind, args, f"{kind}_" + "x".join(map(str, args))))
    for name in MTX:
        rel = os.path.join("tests", "golden", "mtx", name + ".mtx")
        c = matrix_case("mtx", [os.path.join(ROOT, rel)], "mtx_" + name)
        c["args"] = [rel]
        cases.append(c)
    with open(os.path.join(GOLD, "matrices.json"), "w") as f:
        json.dump({"generator": "oracle/make_golden.py via oracle/_ref/ref_host + ref_search",
                   "cases": cases}, f, separators=(",", ":"))

    # CompareResults (utils.h:692-742) verdicts on crafted vectors
    cmp_cases = []
    probes = [
        ("f32", [1.0, 2.0, 3.0], [1.0, 2.0, 3.0]),
        ("f32", [1.0, 2.0, 3.0], [1.0, 2.0000002, 3.0]),
        ("f32", [1.0, 2.0, 3.0], [1.0, 2.5, 3.0]),
        ("f32", [1.0, 2.0, 3.0], [1.0, -2.0, 3.0]),
        ("f32", [0.0, 0.0], [0.0, 1e-30]),
        ("f64", [1.0, 2.0, 3.0], [1.0, 2.0 + 1e-12, 3.0]),
        ("f64", [1.0, 2.0, 3.0], [1.0, 2.1, 3.0]),
        ("f64", [5.0] * 40, [5.0] * 39 + [5.0001]),
        ("f64", [5.0] * 40, [5.0] * 39 + [6.0]),
        ("f32", [1.0] * 3000, [1.0] * 2999 + [1.5]),
        ("f32", [1.0] * 3000, [1.0] * 2999 + [100.0]),
    ]
    for prec, a, b in probes:
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(f"{len(a)}\n" + " ".join(repr(v) for v in a) + "\n" + " ".join(repr(v) for v in b) + "\n")
            name = f.name
        verdict = int(run(REF_HOST, "cmp", prec, name).strip())
        os.unlink(name)
        if len(a) > 64:   # long constant vectors stored as a spec: [fill]*(len-1) + [last]
            a = {"len": len(a), "fill": a[0], "last": a[-1]}
            b = {"len": len(b), "fill": b[0], "last": b[-1]}
        cmp_cases.append({"prec": prec, "computed": a, "reference": b, "verdict": verdict})

    # ReduceByKeyOp<Sum> (thread_operators.cuh:278-302) left folds
    rbk_cases = []
    for pairs in ([(1, 1.0), (1, 2.0), (2, 3.0), (2, 0.5), (3, 1.0)],
                  [(0, 0.25)] * 6,
                  [(4, 1.5), (5, -1.5), (5, 2.0), (9, 0.0), (9, 0.0), (9, 7.0)]):
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(f"{len(pairs)}\n" + "\n".join(f"{k} {v!r}" for k, v in pairs) + "\n")
            name = f.name
        out = run(REF_SEARCH, "rbk", name).strip().splitlines()
        os.unlink(name)
        rbk_cases.append({"pairs": [list(p) for p in pairs],
                          "inclusive": [[int(l.split()[0]), float(l.split()[1])] for l in out]})

    # CommandLineArgs (utils.h:280-387)
    argv_cases = []
    for argv in (["--quiet", "--i=7", "--grid2d=12", "foo", "--alpha=0.5", "--mtx=a/b.mtx"],
                 ["--fp32", "--grid2d", "--i=abc"],
                 ["-quiet", "--i=3", "--i=9", "bar", "baz"]):
        argv_cases.append({"argv": argv, "parsed": json.loads(run(REF_HOST, "args", *argv))})

    with open(os.path.join(GOLD, "host_semantics.json"), "w") as f:
        json.dump({"generator": "oracle/make_golden.py", "compare_results": cmp_cases,
                   "reduce_by_key": rbk_cases, "command_line": argv_cases}, f, separators=(",", ":"))

    # The reference's own doc-comment known answer, cub/device/device_spmv.cuh:90-123
    kat = {
        "source": "cub/device/device_spmv.cuh:90-123 (doc comment)",
        "rows": 9, "cols": 9, "nnz": 24,
        "values": [1] * 24,
        "column_indices": [1, 3, 0, 2, 4, 1, 5, 0, 4, 6, 1, 3, 5, 7, 2, 4, 8, 3, 7, 4, 6, 8, 5, 7],
        "row_offsets": [0, 2, 5, 7, 10, 14, 17, 19, 22, 24],
        "x": [1] * 9,
        "y": [2, 3, 2, 3, 4, 3, 2, 3, 2],
    }
    with open(os.path.join(GOLD, "kat_device_spmv.json"), "w") as f:
        json.dump(kat, f)
    print("wrote", os.listdir(GOLD))


if __name__ == "__main__":
    main()
