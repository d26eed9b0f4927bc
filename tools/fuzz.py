#!/usr/bin/env python3
"""tools/fuzz.py [seconds] -- randomized differential test of the C ABI on the GPU: random CSR shapes
(row-length families, empty rows, giant rows), precisions, array alignments (views at element offsets
0..3: the unaligned ones take the scalar fallback), tuning flags, alpha/beta, SpMM widths and leading
dimensions, forced column bands (the clock-scheduled one-pass form under random clocks, or the passes), the prepared band-major plan (random band counts, sorted and unsorted rows, alpha/beta) and the C multi-GPU
operator (1..8 parts on this device, peer exchange); results compared with an fp64 segment-sum on the GPU under the strict
per-row bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import multi_gpu as MG

# 16 = beyond the self-searching small-problem kernel: ONE launch of tile_kernel_snap; + 0x40000000 = the classic three launches
FLAGS = [0, 0, 0, 2, 4, 8, 16, 16, 16, 20, 24, 48, 80, 128, 144, 0xF000010, 0x3000010, 0xE000010, 0x20000010, 0x20000000, 0x40000000, 0x40000000,
         0x40000010, 0x40000010, 0x60000010, 0x4E000010, 0x40000030, 0x40000050,
         -0x80000000, -0x80000000 | 16, -0x80000000 | 32]          # (MSPMV_TUNE_NO_LEAN: closed short-row tiles through the general reduction as well)
SHAPES = {4: [(256, 7), (256, 11)], 8: [(256, 7), (256, 11)]}           # the product library's shapes (the sweep shapes live in the dev build)


def random_lens(rng, rows):
    fam = rng.integers(0, 7)
    if fam == 0: lens = rng.integers(0, 8, rows)
    elif fam == 1: lens = np.full(rows, rng.integers(1, 70))
    elif fam == 2: lens = np.minimum((rng.pareto(1.1, rows) * 3).astype(np.int64), 50000)
    elif fam == 3: lens = np.zeros(rows, np.int64); lens[rng.integers(0, rows, max(1, rows // 1000))] = rng.integers(1, 200000)
    elif fam == 4: lens = np.where(rng.random(rows) < 0.9, 0, rng.integers(1, 40, rows))
    elif fam == 5: lens = rng.integers(0, 3, rows); lens[rng.integers(0, rows)] = rng.integers(10000, 400000)
    else: lens = np.zeros(rows, np.int64)
    return np.asarray(lens, np.int64)


def offset_view(t, off):
    """a view whose data pointer is `off` elements past an aligned allocation"""
    # surroundings: NaN for floating-point arrays (anything read from there and used shows in the result), 0 for indices
    buf = torch.full((t.numel() + 8,), float("nan") if t.is_floating_point() else 0, dtype=t.dtype, device=t.device)
    v = buf[off: off + t.numel()]
    v.copy_(t)
    return v


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    t_end = time.time() + budget
    cases = 0
    compact_pairs = 0
    worst = 0.0
    # every case is watched: one that does not come back within 90 s is reported with what it was and the run ends (exit 3) -- a hang
    # must not look like a budget that ran out
    import threading
    state = {"what": "start", "since": time.time(), "case": 0}
    def watch():
        while True:
            time.sleep(5)
            if time.time() - state["since"] > 90:
                sys.stderr.write(f"fuzz: WATCHDOG -- case {state['case']} has not come back for 90 s: {state['what']}\n"); sys.stderr.flush()
                sys.stdout.write(f"fuzz: WATCHDOG seed={seed} case={state['case']}: {state['what']}\n"); sys.stdout.flush()
                os._exit(3)
    threading.Thread(target=watch, daemon=True).start()
    t_note = time.time()
    while time.time() < t_end:
        if time.time() - t_note > 120:
            print(f"fuzz: ... {cases} cases so far", flush=True); t_note = time.time()
        f32 = bool(rng.integers(0, 2))
        tdt = torch.float32 if f32 else torch.float64
        vb = 4 if f32 else 8
        big = rng.random() < float(os.environ.get("FUZZ_BIG", "0.04"))   # now and then a problem beyond the fused small-problem path (> 2048 tiles)
        rows = int(rng.choice([1_000_000, 3_000_000])) if big else int(rng.choice([1, 2, 3, 5, 17, 100, 1000, 5000, 40000, 300000]))
        cols = int(rng.choice([1, 2, 7, 64, 1000, 100000]))
        lens = random_lens(rng, rows)
        if rng.random() < float(os.environ.get("FUZZ_LONG", "0.01")):      # now and then a matrix whose rows are ALL long, around the sizes at which such
            big = True                                                       # calls switch to the classic three launches (mspmv_api.hip: long_rows_rule)
            rows = int(rng.integers(15_000, 60_000)); lens = rng.integers(int(rng.choice([150, 230, 380])), 700, rows).astype(np.int64)
        cap = 30_000_000 if big else 1_500_000
        if lens.sum() > cap: lens = lens // (lens.sum() // cap + 1)
        off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
        nnz = int(off[-1])
        col = torch.randint(0, cols, (nnz,), device="cuda", dtype=torch.int32)
        val = (torch.rand(nnz, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
        offs = torch.from_numpy(off.astype(np.int32)).cuda()
        a_off, c_off, r_off = (int(rng.integers(0, 4)) if rng.random() < 0.25 else 0 for _ in range(3))
        val_v, col_v, off_v = offset_view(val, a_off), offset_view(col, c_off), offset_view(offs, r_off)
        lens_i = torch.from_numpy(lens).cuda()
        segsum = lambda data: torch.segment_reduce(data, "sum", lengths=lens_i, axis=0, unsafe=True)
        flags = int(rng.choice(FLAGS))
        shape = SHAPES[vb][int(rng.integers(0, 2))] if rng.random() < 0.5 else (0, 0)
        eps = 2.0 ** -24 if f32 else 2.0 ** -53
        lens_t = torch.from_numpy(lens).cuda().double()
        # column-band passes forced now and then (taken by the 256x11 three-pass path only): a re-association of the same sums
        passes = int(rng.choice([0, 0, 0, 2, 3, 7]))
        if os.environ.get("FUZZ_BAND"):            # every case through the passes: the 256x11 three-pass path, 2..9 bands
            shape = (256, 7) if (not f32 and rng.random() < 0.4) else (256, 11)
            flags |= 16; flags &= ~4; passes = int(rng.integers(2, 10))
        cfac = 2.0 * (torch.ceil(torch.log2(lens_t + 1)) + 16 + 8 + passes)
        state.update(what=f"f32={f32} rows={rows} cols={cols} nnz={nnz} flags={flags:#x} shape={shape} passes={passes} offsets={a_off, c_off, r_off}", since=time.time(), case=cases)
        try:
            M.set_tuning(vb, shape[0], shape[1], flags)
            M.set_band_passes(vb, passes)
            # a banded call runs the clock-scheduled one-pass form (csrc/mspmv_tdm.hpp; any band width, slot length, lookahead: the clock
            # must not matter) or, with that switched off, the passes
            if rng.random() < 0.35: M.set_tdm(vb, -1)
            else: M.set_tdm(vb, int(rng.choice([0, 1])), int(rng.choice([0, 0, 1, 300, 3000, 100000])), int(rng.choice([0, 0, 1, 2, 5, 33])), int(rng.choice([0, 0, 3, 8, 11, 14, 17, 20])))
            polls = int(rng.choice([0, 0, 0, 1, -1]))
            M.set_record_polls(polls)       # default / one look / never look at the published records: the recomputing path
            mode = rng.integers(0, 5)
            state["what"] += f" mode={int(mode)} polls={polls}"
            if mode == 3:                              # prepared band-major plan (mspmv_csrmv_plan_*)
                x = (torch.rand(cols, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
                colp = col
                if rng.random() < 0.5 and nnz:         # the reference's CSR has sorted rows: the order-preserving scatter path
                    rowid = torch.repeat_interleave(torch.arange(rows, device="cuda"), lens_i)
                    colp = col[torch.sort(rowid * cols + col.long(), stable=True).indices].contiguous()
                bands = int(rng.choice([0, 1, 2, 3, 8, 16, 24, 64]))
                if bands * rows + nnz > 2**31 - 65537: bands = 1
                alpha, beta = (1.0, 0.0) if rng.random() < 0.5 else (float(rng.uniform(-2, 2)), float(rng.choice([0.0, 0.5, -1.0])))
                y0 = (torch.rand(rows, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
                y = y0.clone()
                plan = M.CsrMVPlan(val, offs, colp, cols, bands=bands)
                plan(x, y, alpha=alpha, beta=beta)
                prod = val.double() * x.double()[colp.long()]
                g = segsum(prod); s = segsum(prod.abs())
                want = alpha * g + beta * y0.double()
                cf = cfac + 2.0 * plan.bands
                tol = cf * eps * (abs(alpha) * s + abs(beta) * y0.double().abs()) + 4 * eps * want.abs()
                err = (y.double() - want).abs()
                bad = err > tol
                if alpha == 1.0 and beta == 0.0: bad |= (lens_t == 0) & (y != 0)
                ratio = float((err / (tol + 1e-300)).max()) if rows else 0.0
            elif mode == 4:                            # the C multi-GPU operator, all parts on this device (peer exchange)
                parts = int(rng.choice([1, 2, 3, 5, 8]))
                x = (torch.rand(cols, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
                row_split, nz_split = MG.partition(off, parts)
                plan = MG.MgPlan(row_split, nz_split, cols, tdt, list(range(parts)), [0] * parts)
                try:
                    for gpart in range(parts):
                        lo = MG.local_offsets(off, row_split[gpart], row_split[gpart + 1], nz_split[gpart], nz_split[gpart + 1])
                        a, b = int(nz_split[gpart]), int(nz_split[gpart + 1])
                        plan.set_part(gpart, val[a:b].clone(), torch.from_numpy(lo).cuda(), col[a:b].clone())
                    plan.x(0).copy_(x); torch.cuda.synchronize()
                    plan.csrmv(); plan.synchronize()
                    y = torch.cat([plan.y(gpart) for gpart in range(parts)]) if rows else torch.empty(0, dtype=tdt, device="cuda")
                finally:
                    plan.close()
                prod = val.double() * x.double()[col.long()]
                g = segsum(prod); s = segsum(prod.abs())
                tol = (cfac + 2.0 * parts) * eps * s
                err = (y.double() - g).abs()
                bad = (err > tol) | ((lens_t == 0) & (y != 0))
                ratio = float((err / (tol + 1e-300)).max()) if rows else 0.0
            elif mode < 2:                               # CsrMV / axpby (+ prepared)
                x = (torch.rand(cols, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
                alpha, beta = (1.0, 0.0) if mode == 0 else (float(rng.uniform(-2, 2)), float(rng.choice([0.0, 0.5, -1.0])))
                y0 = (torch.rand(rows, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
                y = y0.clone()
                ws = M.CsrMVWorkspace(rows, nnz, tdt)
                if rng.random() < 0.3: ws.prepare(off_v)
                M.csrmv(val_v, off_v, col_v, x, y=y, num_cols=cols, workspace=ws,
                        alpha=None if mode == 0 else alpha, beta=None if mode == 0 else beta)
                # the compact front end (small problems; its fast lane needs right hints: the SECOND call on a workspace)
                # against the general kernel of the same library: bit for bit, whatever the flags make of the call
                if rng.random() < 0.5 and polls != 1 and not (flags & 2):          # (not with the atomic fix-up, whose order varies from run to run; one look: whether a record is there at that look depends on timing -- the last bits of a long row may differ from call to call)
                    ys = []
                    # (a large fp64 matrix of short rows over a tiny x takes the small shape by the column count -- mspmv_api.hip: skinny_rule --
                    #  and its result is bit for bit the general kernel's OF THAT SHAPE: the three variants then run under the forced small shape)
                    skinny = M.launch_info(rows, nnz, vb, num_cols=cols)["items_per_thread"] != M.launch_info(rows, nnz, vb)["items_per_thread"]
                    if skinny: M.set_tuning(vb, 256, 7, flags)
                    for ct in ((1 << 30) if skinny else 0, -1, int(rng.choice([(1 << 30) if skinny else 0, 3, 100000]))):
                        M.set_compact_tiles(ct)
                        yc = y0.clone()
                        M.csrmv(val_v, off_v, col_v, x, y=yc, num_cols=cols, workspace=ws,
                                alpha=None if mode == 0 else alpha, beta=None if mode == 0 else beta)
                        ys.append(yc)
                    M.set_compact_tiles(0)
                    if skinny: M.set_tuning(vb, shape[0], shape[1], flags)
                    # (the first call's y joins the comparison unless it may have run another shape: a PREPARED call of a skinny-rule matrix runs the default shape)
                    if not (torch.equal(ys[0], ys[1]) and torch.equal(ys[2], ys[1]) and (skinny or torch.equal(y, ys[1]))):
                        print(f"COMPACT != GENERAL seed={seed} case={cases}: f32={f32} rows={rows} cols={cols} nnz={nnz} flags={flags:#x} shape={shape} mode={mode}", flush=True)
                        sys.exit(1)
                    compact_pairs += 1
                prod = val.double() * x.double()[col.long()]
                g = segsum(prod)
                s = segsum(prod.abs())
                want = alpha * g + beta * y0.double()
                tol = cfac * eps * (abs(alpha) * s + abs(beta) * y0.double().abs()) + (0 if mode == 0 else 4 * eps * want.abs())
                err = (y.double() - want).abs()
                bad = err > tol
                if mode == 0: bad |= (lens_t == 0) & (y != 0)
                ratio = float((err / (tol + 1e-300)).max()) if rows else 0.0
            else:                                      # SpMM
                k = int(rng.integers(1, 40)); padx, pady = int(rng.integers(0, 3)), int(rng.integers(0, 3))
                Xw = (torch.rand(cols, k + padx, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt); X = Xw[:, padx:]
                Yw = torch.zeros(rows, k + pady, dtype=tdt, device="cuda"); Y = Yw[:, pady:]
                M.csrmm(val_v, off_v, col_v, X, Y=Y)
                prod = val.double()[:, None] * X.double()[col.long()]
                g = segsum(prod)
                s = segsum(prod.abs())
                err = (Y.double() - g).abs(); tol = cfac[:, None] * eps * s
                bad = (err > tol) | ((lens_t == 0)[:, None] & (Y != 0))
                if pady: bad = bad | (Yw[:, :pady] != 0).any(dim=1, keepdim=True)
                ratio = float((err / (tol + 1e-300)).max()) if rows else 0.0
            if bool(bad.any()):
                print(f"MISMATCH seed={seed} case={cases}: f32={f32} rows={rows} cols={cols} nnz={nnz} flags={flags:#x} shape={shape} passes={passes} "
                      f"offsets={a_off, c_off, r_off} mode={mode} bad={int(bad.sum())}", flush=True)
                sys.exit(1)
            worst = max(worst, ratio)
        finally:
            M.set_tuning(vb)
            M.set_band_passes(vb, 0)
            M.set_tdm(vb, 0)
            M.set_record_polls(0)
            M.set_compact_tiles(0)
        cases += 1
    torch.cuda.synchronize()
    print(f"fuzz: {cases} cases in {budget:.0f} s, all within tolerance (worst |err|/bound = {worst:.3f}); {compact_pairs} of them also run with the "
          f"compact front end on / off / at a random limit: bit for bit the same y")


if __name__ == "__main__":
    main()
