#!/usr/bin/env python3
"""tools/big_workloads.py -- SURVEY.md 8(d) configurations beyond C2 on one GPU: C3 stand-ins (R-MAT with
webbase-1M's and com-Orkut's row/nonzero counts, fp64), C4 (fp32 degenerate, both value sets) and C5 at
G = 1 (R-MAT scale 26, 2e9 edges, fp64).  Reference timing protocol (warm-up, N back-to-back calls, hipEvents);
results verified on a row sample against a sequential fp64-accumulated sum computed on the host."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G


def sample_check(A, x, y, nsample=2000, ipt=11):
    rows = A.rows
    rng = np.random.default_rng(1)
    lens = (A.row_offsets[1:].long() - A.row_offsets[:-1].long())
    pick = torch.from_numpy(rng.integers(0, rows, nsample)).cuda()
    pick = torch.cat([pick, torch.topk(lens, 4).indices])          # and the longest rows
    worst = 0.0
    eps = 2.0 ** -24 if A.values.dtype == torch.float32 else 2.0 ** -53
    for r in pick.tolist():
        a, b = int(A.row_offsets[r]), int(A.row_offsets[r + 1])
        v = A.values[a:b].double(); xv = x[A.column_indices[a:b].long()].double()
        g = float((v * xv).sum()); s = float((v * xv).abs().sum())
        c = 2 * (int(np.ceil(np.log2(b - a + 1))) + ipt + 8)
        got = float(y[r])
        if b == a:
            assert got == 0.0, (r, got)
        else:
            worst = max(worst, abs(got - g) / (c * eps * s + 1e-300))
    return worst


def run(label, A, x, out):
    vb = A.values.element_size()
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    torch.cuda.synchronize()
    worst = sample_check(A, x, y)
    iters = int(min(max((1 << 34) // max(A.nnz, 1), 100), 2000))      # reference: clamp(2^34/nnz, 100, 50000); capped for time
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    balg = A.nnz * (vb + 4) + (A.rows + 1) * 4 + A.rows * vb + A.cols * vb
    eff = A.nnz * (2 * vb + 4) + A.rows * (4 + vb)
    rec = {"workload": label, "rows": A.rows, "cols": A.cols, "nnz": A.nnz, "dtype": "f32" if vb == 4 else "f64", "iters": iters,
           "ms": round(ms, 5), "gflops": round(2 * A.nnz / ms / 1e6, 2), "B_alg_GBs": round(balg / ms / 1e6, 1),
           "effective_GBs_reference_formula": round(eff / ms / 1e6, 1), "strict_tolerance_ratio_on_sample": round(worst, 4)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    assert worst <= 1.0, worst


def main():
    which = sys.argv[1:] or ["c3web", "c3orkut", "c4", "c5"]
    out = []
    for w in which:
        t0 = time.time()
        if w == "c3web":
            A = G.rmat_csr(20, 3_105_536, dtype=torch.float64, seed=G.SEED_C3)
            run("C3 stand-in webbase-1M-sized: R-MAT scale 20, 3 105 536 edges, fp64", A, G.uniform_pm1(G.SEED_C3 + 2, A.cols, torch.float64, "cuda"), out)
        elif w == "c3orkut":
            A = G.rmat_csr(22, 234_370_166, dtype=torch.float64, seed=G.SEED_C3)
            run("C3 stand-in com-Orkut-sized: R-MAT scale 22, 234 370 166 edges, fp64", A, G.uniform_pm1(G.SEED_C3 + 2, A.cols, torch.float64, "cuda"), out)
        elif w == "c4":
            for ones in (True, False):
                A = G.degenerate_csr(dtype=torch.float32, ones=ones)
                x = torch.ones(A.cols, dtype=torch.float32, device="cuda") if ones else G.uniform_pm1(G.SEED_C4 + 2, A.cols, torch.float32, "cuda")
                run("C4 degenerate fp32, " + ("all ones" if ones else "uniform values"), A, x, out)
                if ones:
                    y = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
                    assert float(y[A.rows // 2]) == float(1 << 26), float(y[A.rows // 2])      # closed form, exactly representable
        elif w == "c5":
            A = G.rmat_csr(26, 2_000_000_000, dtype=torch.float64, seed=G.SEED_C5)
            run("C5 at G=1: R-MAT scale 26, 2e9 edges, fp64", A, G.uniform_pm1(G.SEED_C5 + 2, A.cols, torch.float64, "cuda"), out)
        print(f"# {w}: built + run in {time.time() - t0:.1f} s", flush=True)
        del A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
