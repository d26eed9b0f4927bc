#!/usr/bin/env python3
"""tools/trace_tiles.py [workload] -- development: per-phase cycle stamps inside tile_kernel_vec
(the ABLATE=6 build of the kernel writes clock64() at phase boundaries for the first 16 tiles of each block).
Needs the development library: make -C merge_spmv_amd exp && MSPMV_LIB=merge_spmv_amd/libmspmv_exp.so python tools/trace_tiles.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, numpy as np
import merge_spmv_amd as M
import sweep
lib = M.load_library()
lib.mspmv_dev_set_trace.argtypes = [ctypes.c_void_p]
names = sys.argv[1:] or ["dense32d"]
for label, A, x in sweep.workloads(names):
    vb = A.values.element_size()
    info = M.launch_info(A.rows, A.nnz, vb)
    M.set_tuning(vb, info["block_threads"], info["items_per_thread"], 0x60000 | 0x400)    # stamps build, resident grid of 4 blocks per CU
    nblk = 4 * 256 + 64
    buf = torch.zeros(nblk * 16 * 8, dtype=torch.int64, device="cuda")
    assert lib.mspmv_dev_set_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    for _ in range(3):
        buf.zero_()
        M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nblk, 16, 8).astype(np.float64)
    print("  stamps recorded per slot:", [(int((t[:, :, i] > 0).sum())) for i in range(8)])
    ok = (t[:, :, 5] > 0) & (t[:, :, 0] > 0)
    ok[:, :2] = False            # steady state only
    ok[:, 12:] = False
    sel = t[ok]
    print(f"== {label}: tile {info['block_threads']}x{info['items_per_thread']}, {ok.sum()} traced tiles (cycles of the 100 MHz*? shader clock; averages)")
    def d(a, b): return float(np.mean(sel[:, b] - sel[:, a]))
    print(f"  wait for the tile's stream loads (issued one iteration ago) : {d(0,1):8.0f}")
    print(f"  staging: row offsets + x gathers + products/flags to LDS + barrier: {d(1,2):8.0f}")
    print(f"  issue next tile's loads                                     : {d(2,3):8.0f}")
    print(f"  nonzero phase up to the block scan                          : {d(3,6):8.0f}")
    print(f"  block scan + carry + S write + barrier                      : {d(6,7):8.0f}")
    print(f"  row phase (y stores) + carry-out                            : {d(7,4):8.0f}")
    print(f"  end-of-tile barrier                                         : {d(4,5):8.0f}")
    # iteration period: stamp 0 of consecutive tiles
    per = t[:, 3:12, 0] - t[:, 2:11, 0]
    okp = (t[:, 3:12, 0] > 0) & (t[:, 2:11, 0] > 0) & (t[:, 3:12, 5] > 0)
    print(f"  whole iteration                                             : {float(np.mean(per[okp])):8.0f}")
    M.set_tuning(vb)
