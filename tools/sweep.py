#!/usr/bin/env python3
"""tools/sweep.py -- GPU tuning sweep: every compiled tile shape x flag set on
a few workloads; prints a table of per-kernel milliseconds (hipEvents) and
end-to-end ms.  Development aid, not part of the product or the tests.
usage: python tools/sweep.py [workload ...]   (c2 dense32 c4 rmat grid)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G

SHAPES = {4: [(256, 7), (256, 11)], 8: [(256, 7), (256, 11)]}       # product shapes
if "exp" in os.environ.get("MSPMV_LIB", ""):                         # the dev build (MSPMV_LIB=.../libmspmv_exp.so) has the sweep shapes
    SHAPES = {4: [(256, 7), (256, 11), (256, 9), (256, 15), (256, 5), (128, 7), (512, 7)], 8: [(256, 7), (256, 11), (256, 5), (256, 9), (256, 3), (128, 5), (512, 5)]}


def workloads(names):
    for n in names:
        if n == "c2":
            A = G.uniform_csr(3_125_000, 3_125_000, 32, dtype=torch.float32)
            yield "c2_f32", A, G.uniform_pm1(1, A.cols, torch.float32, "cuda")
        elif n == "c2d":
            A = G.uniform_csr(3_125_000, 3_125_000, 32, dtype=torch.float64)
            yield "c2_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "dense32":
            A = G.dense_csr(3_125_000, 32, dtype=torch.float32, ones=False)
            yield "dense32_f32", A, G.uniform_pm1(1, A.cols, torch.float32, "cuda")
        elif n == "dense32d":
            A = G.dense_csr(3_125_000, 32, dtype=torch.float64, ones=False)
            yield "dense32_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "c4":
            A = G.degenerate_csr(dtype=torch.float32, ones=False)
            yield "c4_f32", A, G.uniform_pm1(1, A.cols, torch.float32, "cuda")
        elif n == "rmat":
            A = G.rmat_csr(22, 60_000_000, dtype=torch.float64, seed=G.SEED_C3)
            yield "rmat22_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n.startswith("g2d"):                      # g2d<width>: a small 5-point grid, e.g. g2d100
            A = G.grid2d_csr(int(n[3:]), torch.float64)
            yield f"grid2d_{n[3:]}_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "circuit":
            A = G.circuit_csr(dtype=torch.float64)
            yield "circuit5M_shaped_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "rmat24":
            A = G.rmat_csr(24, 250_000_000, dtype=torch.float64, seed=G.SEED_C5)
            yield "rmat24_250M_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "orkut":
            A = G.rmat_symmetric_csr(G.C3_ORKUT_SCALE, G.C3_ORKUT_EDGES, dtype=torch.float64, seed=G.SEED_C3)
            yield "orkut_sized_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "web":
            A = G.rmat_csr(20, 3_105_536, dtype=torch.float64, seed=G.SEED_C3)
            yield "rmat20_webbase_like_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "dense5d":
            A = G.dense_csr((1 << 24) // 5, 5, dtype=torch.float64, ones=False)
            yield "dense5_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "grid2d4096":
            A = G.grid2d_csr(4096, torch.float64)
            yield "grid2d_4096_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "grid2d":
            A = G.grid2d_csr(2000, torch.float64)
            yield "grid2d_2000_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "grid3d":
            A = G.grid3d_csr(200, torch.float64)
            yield "grid3d_200_f64", A, G.uniform_pm1(1, A.cols, torch.float64, "cuda")
        elif n == "band":
            # banded: 5 nnz/row near the diagonal (grid-like locality)
            rows = 16_000_000
            off = (torch.arange(rows + 1, dtype=torch.int64, device="cuda") * 5).to(torch.int32)
            r = torch.arange(rows, dtype=torch.int64, device="cuda").repeat_interleave(5)
            d = torch.tensor([-4000, -1, 0, 1, 4000], device="cuda").repeat(rows)
            c = (r + d).clamp_(0, rows - 1).to(torch.int32)
            A = G.DeviceCsr(rows, rows, off, c, G.uniform_pm1(2, rows * 5, torch.float32, "cuda"))
            yield "band5_f32", A, G.uniform_pm1(1, rows, torch.float32, "cuda")


def time_it(A, x, iters=30):
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    for _ in range(3):
        M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / iters * 1e3
    M.profile_begin(iters)
    for _ in range(iters):
        M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    torch.cuda.synchronize()
    p = M.profile_end()
    return total, p


def main():
    names = sys.argv[1:] or ["c2", "dense32"]
    flag_sets = [int(f, 0) for f in os.environ.get("SWEEP_FLAGS", "0").split(",")]
    only_default_shape = os.environ.get("SWEEP_DEFAULT_SHAPE") == "1"
    for label, A, x in workloads(names):
        vb = A.values.element_size()
        balg = A.nnz * (vb + 4) + (A.rows + 1) * 4 + A.rows * vb + A.cols * vb
        print(f"== {label}: rows {A.rows} nnz {A.nnz}  B_alg {balg/1e6:.1f} MB", flush=True)
        info0 = M.launch_info(A.rows, A.nnz, vb)
        for (b, i) in ([(info0["block_threads"], info0["items_per_thread"])] if only_default_shape else SHAPES[vb]):
            for fl in flag_sets:
                M.set_tuning(vb, b, i, fl)
                total, p = time_it(A, x)
                print(f"  {b:4d}x{i:<3d} flags {fl}: total {total:8.4f} ms  search {p['search_ms']:.4f} tile {p['tile_ms']:.4f} "
                      f"fix {p['fixup_ms']:.4f}  | {2*A.nnz/total/1e6:9.1f} GFLOP/s  alg {balg/p['tile_ms']/1e6:8.1f} GB/s", flush=True)
        M.set_tuning(vb)
        total, pr = time_it(A, x)
        info = M.launch_info(A.rows, A.nnz, vb, num_cols=A.cols)      # (the shape the default call runs: by the column count too)
        print(f"  DEFAULT {info['block_threads']}x{info['items_per_thread']}: total {total:8.4f} ms  search {pr['search_ms']:.4f} tile {pr['tile_ms']:.4f} fix {pr['fixup_ms']:.4f}  | {2*A.nnz/total/1e6:9.1f} GFLOP/s", flush=True)
        ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype).prepare(A.row_offsets)
        yp = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
        call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=yp, num_cols=A.cols, workspace=ws)
        for _ in range(3): call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): call()
        torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 30 * 1e3
        print(f"  prepared (tile coordinates found once): total {tp:8.4f} ms | {2*A.nnz/tp/1e6:9.1f} GFLOP/s", flush=True)
        try:
            if os.environ.get("SWEEP_NO_ROCSPARSE"): raise RuntimeError("skipped (SWEEP_NO_ROCSPARSE)")
            import rocsparse_ref
            ana, avg, yr = rocsparse_ref.time_csrmv(A, x)
            ym = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
            err = float((ym.double() - yr.double()).abs().max())
            print(f"  rocSPARSE csrmv: analysis {ana:.3f} ms, avg {avg:.4f} ms | {2*A.nnz/avg/1e6:9.1f} GFLOP/s  (max |mspmv - rocsparse| = {err:.3g})", flush=True)
        except Exception as e:
            print("  rocSPARSE unavailable:", e)
        del A, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
