#!/usr/bin/env python3
"""tools/band_passes_bench.py [case ...] -- the column-band passes of the stateless call (mspmv_set_band_passes):
never / automatic / forced 2, 3, 4 passes (BAND_FORCE=2,3,4,5,6,8 in the environment: other counts) on uniformly spread matrices of growing x (what the policy table in
csrc/mspmv_api.hip: band_passes_for was read from), on the headline C2 in both precisions, and on matrices the
device-side windows must refuse (R-MAT, banded, streaming) -- where `overhead` re-times never vs automatic,
interleaved, 5 x 300 calls.  Every result is checked (strict bound, reproducibility) before it is timed.
    python tools/band_passes_bench.py            # everything
    python tools/band_passes_bench.py overhead   # only the never-vs-automatic cost on the refused matrices"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G

f32, f64 = torch.float32, torch.float64


def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def banded(n, w, dt):
    """row i: w consecutive columns around i (clipped): a local matrix"""
    rows = torch.arange(n, device="cuda", dtype=torch.int64)
    start = (rows - w // 2).clamp(0, n - w)
    cols = (start[:, None] + torch.arange(w, device="cuda")[None, :]).reshape(-1).to(torch.int32)
    off = (torch.arange(n + 1, device="cuda", dtype=torch.int64) * w).to(torch.int32)
    vals = (torch.rand(n * w, device="cuda", dtype=torch.float64) * 2 - 1).to(dt)
    return G.DeviceCsr(n, n, off, cols, vals)


def cases():
    yield "c2_f32", lambda: G.uniform_csr(3_125_000, 3_125_000, 32, dtype=f32)
    yield "c2_f64", lambda: G.uniform_csr(3_125_000, 3_125_000, 32, dtype=f64)
    for mb in (4, 6, 8, 12, 16, 24, 32, 48, 64):
        yield f"u{mb}MB_f32", (lambda n=mb * 2**20 // 4: G.uniform_csr(3_000_000, n, 32, dtype=f32))
    for mb in (4, 8, 12, 16, 24, 32, 48, 64):
        yield f"u{mb}MB_f64", (lambda n=mb * 2**20 // 8: G.uniform_csr(3_000_000, n, 32, dtype=f64))
    # 160-256 MB of CSR stream: ordinary loads, the matrix stays in the Infinity Cache from pass to pass
    yield "u24Mnnz_7.6MB_f32", lambda: G.uniform_csr(1_000_000, 2_000_000, 24, dtype=f32)
    yield "u24Mnnz_11MB_f32", lambda: G.uniform_csr(750_000, 3_000_000, 32, dtype=f32)
    yield from refused()
    yield "dense32_f32", lambda: G.dense_csr(3_125_000, 32, dtype=f32, ones=False)


def refused():
    yield "band_2M_x12_f32", lambda: banded(2_000_000, 12, f32)
    yield "band_3M_x33_f32", lambda: banded(3_000_000, 33, f32)
    yield "band_2M_x40_f64", lambda: banded(2_000_000, 40, f64)
    yield "rmat21_64M_f32", lambda: G.rmat_csr(21, 64_000_000, dtype=f32, seed=G.SEED_C3)
    yield "rmat22_100M_f32", lambda: G.rmat_csr(22, 100_000_000, dtype=f32, seed=G.SEED_C3)


def fmt(pr):
    return f"[search {pr['search_ms']*1e3:.1f} tile {pr['tile_ms']*1e3:.1f} fix {pr['fixup_ms']*1e3:.1f} us]"


def profile(call, n):
    M.profile_begin(n)
    for _ in range(n): call()
    torch.cuda.synchronize()
    return M.profile_end()


def sweep(only):
    for name, make in cases():
        if only and name not in only: continue
        A = make()
        dt = A.values.dtype; vb = A.values.element_size()
        x = G.uniform_pm1(1, A.cols, dt, "cuda")
        ws = M.CsrMVWorkspace(A.rows, A.nnz, dt); y = torch.empty(A.rows, dtype=dt, device="cuda")
        call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
        lens = (A.row_offsets[1:] - A.row_offsets[:-1]).long()
        prod = A.values.double() * x.double()[A.column_indices.long()]
        g = torch.segment_reduce(prod, "sum", lengths=lens, unsafe=True); s = torch.segment_reduce(prod.abs(), "sum", lengths=lens, unsafe=True)
        del prod
        eps = 2.0 ** -24 if vb == 4 else 2.0 ** -53
        tol = 2.0 * (torch.ceil(torch.log2(lens.double() + 1)) + 32) * eps * s
        out = []
        forced = [int(v) for v in os.environ.get("BAND_FORCE", "2,3,4").split(",") if v]
        for label, passes in [("never", -1), ("auto", 0), ("passes", 0)] + [(f"force{b}", b) for b in forced] + [("force_clocked", 4)]:
            # auto / force_clocked: the clock-scheduled one-pass form (csrc/mspmv_tdm.hpp) where the passes are offered / on any call; passes, forceN: the passes
            M.set_band_passes(vb, passes); M.set_tdm(vb, 0 if label in ("auto", "never") else 1 if label == "force_clocked" else -1)
            y.fill_(float("nan")); call(); torch.cuda.synchronize()
            bad = int(((y.double() - g).abs() > tol).sum()) + int(torch.isnan(y).sum())
            y2 = y.clone(); call(); torch.cuda.synchronize()
            rep = bool(torch.equal(y, y2))
            ms = timeit(call)
            extra = ""
            if passes == 0:
                offered = M.band_passes(A.rows, A.cols, A.nnz, vb)
                extra = f" (offered {offered}"
                if offered: extra += f", windows {int(M.debug_band_windows(ws, A.rows, A.nnz, vb).sum())}/64"
                extra += ") " + fmt(profile(call, 20))
            out.append(f"{label} {ms:7.4f}{'' if bad == 0 else ' BAD=' + str(bad)}{'' if rep else ' NONREPRO'}{extra}")
        M.set_band_passes(vb, 0); M.set_tdm(vb, 0)
        print(f"{name:18s} x {A.cols * vb / 2**20:5.1f} MiB nnz {A.nnz/1e6:5.1f}M: " + "  ".join(out), flush=True)
        del A, x, y, ws, g, s


def overhead():
    print("never vs automatic on matrices the windows refuse (min and median of 5 x 300 calls, interleaved):")
    for name, make in refused():
        A = make(); dt = A.values.dtype; vb = A.values.element_size()
        x = G.uniform_pm1(1, A.cols, dt, "cuda")
        ws = M.CsrMVWorkspace(A.rows, A.nnz, dt); y = torch.empty(A.rows, dtype=dt, device="cuda")
        call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
        res = {-1: [], 0: []}; prof = {}
        for _ in range(5):
            for mode in (-1, 0):
                M.set_band_passes(vb, mode)
                res[mode].append(timeit(call, 300))
        for mode in (-1, 0):
            M.set_band_passes(vb, mode)
            prof[mode] = profile(call, 100)
        M.set_band_passes(vb, 0)
        print(f"{name:18s} never {min(res[-1])*1e3:7.1f} us (median {sorted(res[-1])[2]*1e3:7.1f}) {fmt(prof[-1])}   "
              f"auto {min(res[0])*1e3:7.1f} us (median {sorted(res[0])[2]*1e3:7.1f}) {fmt(prof[0])}", flush=True)
        del A, x, y, ws


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    if args != ["overhead"]:
        sweep(set(a for a in args if a != "overhead"))
    if not args or "overhead" in args:
        overhead()
