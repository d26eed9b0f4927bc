#!/usr/bin/env python3
"""tools/tdm_check.py [quick|c2|sweep] -- the clock-scheduled column bands (csrc/mspmv_tdm.hpp) against the one-sweep kernel (bitwise)
and against the column-band passes (time), on the GPU box.  Development aid."""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import merge_spmv_amd as M


NO_FUSED = 16
TWO_LAUNCH = 0x40000000


def uniform_csr(rows, cols, npr, dtype, seed=1, dev="cuda"):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    col = torch.randint(0, cols, (rows, npr), generator=g, device=dev, dtype=torch.int32)
    col, _ = torch.sort(col, dim=1)
    off = (torch.arange(rows + 1, device=dev, dtype=torch.int64) * npr).to(torch.int32)
    val = (torch.rand(rows * npr, generator=g, device=dev, dtype=torch.float64) * 2 - 1).to(dtype)
    x = (torch.rand(cols, generator=g, device=dev, dtype=torch.float64) * 2 - 1).to(dtype)
    return val, off, col.reshape(-1).contiguous(), x


def ragged_csr(rows, cols, dtype, seed=2, dev="cuda"):
    """row lengths 0 ... 40 with a few long rows, nnz not a multiple of 4"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 41, rows).astype(np.int64)
    lens[rows // 3] = 20000; lens[rows // 2] = 7; lens[-1] = 3
    if int(lens.sum()) % 4 == 0:
        lens[0] += 1
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    col = rng.integers(0, cols, nnz).astype(np.int32)
    val = rng.uniform(-1, 1, nnz)
    x = rng.uniform(-1, 1, cols)
    t = lambda a, d: torch.from_numpy(a).to(d).cuda()
    return t(val, dtype), t(off.astype(np.int32), torch.int32), t(col, torch.int32), t(x, dtype)


def run(val, off, col, x, cols, reps=0, alpha=None, beta=None, y0=None):
    rows = off.numel() - 1
    ws = M.CsrMVWorkspace(rows, val.numel(), val.dtype)
    if alpha is None:
        y = M.csrmv(val, off, col, x, num_cols=cols, workspace=ws).clone()
    else:
        y = y0.clone()
        M.csrmv(val, off, col, x, y=y, num_cols=cols, workspace=ws, alpha=alpha, beta=beta)
    torch.cuda.synchronize()
    ms = None
    if reps:
        out = torch.empty(rows, dtype=val.dtype, device="cuda")
        for _ in range(3):
            M.csrmv(val, off, col, x, y=out, num_cols=cols, workspace=ws)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            M.csrmv(val, off, col, x, y=out, num_cols=cols, workspace=ws)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
    return y, ms


def quick():
    ok = True
    for dtype, vb in ((torch.float32, 4), (torch.float64, 8)):
        for name, (val, off, col, x), cols in (
                ("uniform32", uniform_csr(220_000, 200_000, 32, dtype), 200_000),
                ("uniform9_wide", uniform_csr(700_001, 3_000_000, 9, dtype), 3_000_000),
                ("ragged", ragged_csr(300_000, 250_000, dtype), 250_000)):
            # the reference: the classic three launches' one-sweep tile kernel (the one-launch kernel snaps its tiles to rows: another association)
            M.set_tuning(vb, flags=TWO_LAUNCH); M.set_band_passes(vb, -1); M.set_tdm(vb, 0)
            ref, _ = run(val, off, col, x, cols)
            for shift in (0, 12, 13, 15, 17, 20):
                M.set_band_passes(vb, 3); M.set_tdm(vb, 1, 0, 0, shift)
                y, _ = run(val, off, col, x, cols)
                same = torch.equal(y, ref)
                ok &= same
                print(f"fp{vb * 8} {name:14s} band_shift {shift:2d}: bitwise equal to the one-sweep y: {same}"
                      + ("" if same else f"  max |diff| {float((y - ref).abs().max()):.3e}, differing rows {int((y != ref).sum())}"))
            # alpha / beta
            y0 = torch.randn(off.numel() - 1, dtype=dtype, device="cuda")
            M.set_band_passes(vb, -1); M.set_tdm(vb, 0)
            ra, _ = run(val, off, col, x, cols, alpha=0.5, beta=-1.25, y0=y0)
            M.set_band_passes(vb, 3); M.set_tdm(vb, 1, 0, 0, 14)
            ya, _ = run(val, off, col, x, cols, alpha=0.5, beta=-1.25, y0=y0)
            same = torch.equal(ya, ra); ok &= same
            print(f"fp{vb * 8} {name:14s} alpha/beta: {same}")
            M.set_band_passes(vb, 0); M.set_tdm(vb, 0); M.set_tuning(vb)
    print("ALL EQUAL" if ok else "MISMATCH")
    return ok


def c2(sweep=False):
    for dtype, vb in ((torch.float32, 4), (torch.float64, 8)):
        val, off, col, x = uniform_csr(3_125_000, 3_125_000, 32, dtype)
        cols = 3_125_000
        M.set_tuning(vb, flags=TWO_LAUNCH); M.set_band_passes(vb, -1); M.set_tdm(vb, 0)
        ref, t_one = run(val, off, col, x, cols, reps=30)
        M.set_tuning(vb)
        M.set_band_passes(vb, 0); M.set_tdm(vb, -1)
        yp, t_pass = run(val, off, col, x, cols, reps=30)
        print(f"fp{vb * 8} C2: one sweep {t_one:.4f} ms | column-band passes {t_pass:.4f} ms", flush=True)
        grid = [(0, 0, 0, 11)]
        if sweep:
            grid = [(sp, la, sh, ipt) for ipt in (11,) for sh in ((18,) if vb == 4 else (17,)) for sp in (700, 800, 875, 950, 1000, 1050, 1125, 1200) for la in (2, 3, 4)]
        for sp, la, sh, ipt in grid:
            M.set_band_passes(vb, 0 if ipt == 11 else 4); M.set_tdm(vb, 1, sp, la, sh)
            if ipt != 11:
                M.set_tuning(vb, 256, ipt, TWO_LAUNCH); M.set_band_passes(vb, -1)
                ref7, _ = run(val, off, col, x, cols)
                M.set_tuning(vb, 256, ipt); M.set_band_passes(vb, 4)
            y, t = run(val, off, col, x, cols, reps=30)
            print(f"fp{vb * 8} C2: clocked bands 256x{ipt} slot {sp or 1000:5d} per mille, lookahead {la - 1 if la else 2}, band shift {sh or 18}: {t:.4f} ms  bitwise the one-sweep y: {torch.equal(y, ref if ipt == 11 else ref7)}", flush=True)
            M.set_tuning(vb)
        M.set_band_passes(vb, 0); M.set_tdm(vb, 0)


def rows8():
    """rows of 8 (a tile holds 313 rows and 2503 nonzeros): where does the slot's optimum sit against the model's?"""
    for dtype, vb in ((torch.float32, 4), (torch.float64, 8)):
        val, off, col, x = uniform_csr(12_500_000, 3_125_000, 8, dtype)
        cols = 3_125_000
        M.set_band_passes(vb, -1); M.set_tdm(vb, 0)
        _, t_one = run(val, off, col, x, cols, reps=20)
        M.set_band_passes(vb, 0); M.set_tdm(vb, -1)
        _, t_pass = run(val, off, col, x, cols, reps=20)
        print(f"fp{vb * 8} rows of 8: one launch {t_one:.4f} ms | column-band passes {t_pass:.4f} ms (offered {M.band_passes(12_500_000, cols, 100_000_000, vb)})", flush=True)
        for sp in (700, 800, 850, 900, 950, 1000, 1100):
            M.set_band_passes(vb, 0); M.set_tdm(vb, 1, sp, 0, 0)
            _, t = run(val, off, col, x, cols, reps=20)
            print(f"fp{vb * 8} rows of 8: clocked bands slot {sp} per mille: {t:.4f} ms", flush=True)
        M.set_band_passes(vb, 0); M.set_tdm(vb, 0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "quick"
    if what == "quick":
        sys.exit(0 if quick() else 1)
    if what == "rows8":
        rows8()
    else:
        c2(sweep=what == "sweep")
