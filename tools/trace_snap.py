#!/usr/bin/env python3
"""tools/trace_snap.py [workload ...] -- development: where a block of the one-launch kernel (tile_kernel_snap) spends its life.
The -DMSPMV_DEV build stamps the 100 MHz wall clock at the block's phase boundaries (thread 0) and its hardware ids; this prints
the phase averages in microseconds, the blocks alive per CU over the launch, and the launch's span.
    make -C merge_spmv_amd exp && MSPMV_LIB=merge_spmv_amd/libmspmv_exp.so python tools/trace_snap.py dense5d grid3d"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, numpy as np
import merge_spmv_amd as M
import sweep
lib = M.load_library()
lib.mspmv_dev_set_trace.argtypes = [ctypes.c_void_p]
flags = int(os.environ.get("TRACE_FLAGS", "0"), 0)
for label, A, x in sweep.workloads(sys.argv[1:] or ["dense5d"]):
    vb = A.values.element_size()
    M.set_tuning(vb, 0, 0, flags)
    info = M.launch_info(A.rows, A.nnz, vb)
    nblk = info["num_tiles"]
    buf = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    for _ in range(3): call()
    torch.cuda.synchronize()
    assert lib.mspmv_dev_set_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    call(); torch.cuda.synchronize()
    assert lib.mspmv_dev_set_trace(None) == 0
    t = buf.cpu().numpy().reshape(nblk, 16)
    lean_us = t[:, 8:13].astype(np.float64) / 100.0
    us = t[:, :6].astype(np.float64) / 100.0
    if not (t[:, 0] != 0).any():
        print(f"== {label}: this call does not run tile_kernel_snap (column-band candidates take the classic launches): nothing traced")
        M.set_tuning(vb); continue
    t0 = us[:, 0].min()
    span = us[:, 5].max() - t0
    hw, xcc = t[:, 6], t[:, 7] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)      # cu_id, sh_id, se_id, xcc
    ncu = len(np.unique(cu))
    life = us[:, 5] - us[:, 0]
    d = lambda a, b: float(np.mean(us[:, b] - us[:, a]))
    print(f"== {label}: {nblk} blocks of {info['block_threads']}x{info['items_per_thread']}, launch span {span:.1f} us, {ncu} CUs seen, flags {flags:#x}")
    print(f"  hints requested -> arrived            : {d(0,1):6.2f} us")
    print(f"  boundaries, stream loads issued       : {d(1,2):6.2f}")
    print(f"  staging (loads awaited, x, LDS, barrier): {d(2,3):6.2f}")
    print(f"  in-tile reduction + y stores issued   : {d(3,4):6.2f}")
    ln = lean_us[:, 0] > 0
    if ln.any():
        e = lambda a, b: float(np.mean(lean_us[ln, b] - lean_us[ln, a]))
        print(f"    lean reduction ({int(ln.sum())} blocks): staging barrier -> its start {float(np.mean(lean_us[ln, 0] - us[ln, 3])):5.2f}, row ends read {e(0,1):5.2f}, "
              f"products read + first {8} added {e(1,2):5.2f}, longer rows {e(2,3):5.2f}, y stores issued {e(3,4):5.2f}")
    print(f"  stores acknowledged                   : {d(4,5):6.2f}")
    print(f"  block life                            : {life.mean():6.2f}  (p10 {np.percentile(life,10):.2f}, p50 {np.percentile(life,50):.2f}, p90 {np.percentile(life,90):.2f})")
    print(f"  blocks alive per CU (sum of lives / span / CUs): {life.sum() / span / ncu:5.2f}")
    # gap between a block's end and the start of the next block that began on the same CU afterwards
    gaps = []
    for c in np.unique(cu)[:64]:
        sel = np.where(cu == c)[0]
        starts = np.sort(us[sel, 0]); ends = np.sort(us[sel, 5])
        k = min(len(starts), len(ends))
        # after the first `resident` blocks, the i-th start follows the (i - resident)-th end
        res = int(round((life[sel].sum() / span)))
        if res >= 1 and k > res:
            gaps.append(np.mean(starts[res:k] - ends[:k - res]))
    if gaps: print(f"  a slot's end -> the next block's first instruction (64 CUs): {np.mean(gaps):6.2f} us")
    # the longest-lived blocks: when they start and end in the launch, and the phase that made them long
    for b in np.argsort(-life)[:3]:
        ph = [us[b, i + 1] - us[b, i] for i in range(5)]
        print(f"  longest: block {b} lives {life[b]:.1f} us, from {us[b, 0] - t0:.1f} to {us[b, 5] - t0:.1f} of {span:.1f}; hints {ph[0]:.1f} issue {ph[1]:.1f} staging {ph[2]:.1f} reduction {ph[3]:.1f} ack {ph[4]:.1f}")
    # fraction of the span during which the first / last 10 % of the blocks run
    order = np.argsort(us[:, 0]); n10 = max(nblk // 10, 1)
    print(f"  last block starts at {us[:, 0].max() - t0:.1f} us; the last 10 % of the blocks start after {us[order[-n10], 0] - t0:.1f} us")
    M.set_tuning(vb)
