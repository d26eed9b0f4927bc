#!/usr/bin/env python3
"""tools/hw_ceilings.py <probe ...> -- the hardware ceilings DESIGN.md prices the kernels against, measured on the GPU
box (kernels: tools/hw_ceilings.hip).  Output is committed under profiles/ (r02_hw_ceilings_<probe>.txt).

  stream    float4 vs dword read stream by grid size; 100 M (index, value, x) gathers by size of x
  shapes    a read stream issued as one-tile-per-block / persistent loops
  policy    100 M random 4-byte gathers over 1 MiB / 16 MiB / 1 GiB tables with every sc0/sc1/nt combination
            (the L2-resident, Infinity-Cache-resident and DRAM random-line rates)
  pattern   cycles per wave gather instruction by address pattern (texture-address unit)
  scalar    the same gather issued as scalar loads (v_readlane + s_load_dword): 64-byte scalar-cache lines vs 128-byte TCP lines
  banded    column-banded traversal of the C2 matrix WITHOUT re-laying it out (what a stateless call could do): the probe
            that motivated the prepared plan
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhw_ceilings.so"))
vp = ctypes.c_void_p


def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def stream():
    lib.ub_stream_f4.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, vp]
    lib.ub_stream_f1.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, vp]
    lib.ub_gather.argtypes = [vp, vp, vp, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp]
    out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
    buf = torch.empty(800_000_000 // 4, dtype=torch.float32, device="cuda").uniform_()
    for blocks in (2048, 4096, 8192, 16384):
        t4 = timeit(lambda: lib.ub_stream_f4(vp(buf.data_ptr()), buf.numel() * 4, vp(out.data_ptr()), blocks, s), 20)
        t1 = timeit(lambda: lib.ub_stream_f1(vp(buf.data_ptr()), buf.numel() * 4, vp(out.data_ptr()), blocks, s), 20)
        print(f"stream 800MB blocks {blocks}: float4 {t4:.4f} ms = {0.8/t4:.2f} TB/s | dword {t1:.4f} ms = {0.8/t1:.2f} TB/s", flush=True)
    n = 100_000_000
    val = buf[:n]
    for xbytes in (1 << 20, 3 << 20, 12_500_000, 100_000_000, 1_000_000_000):
        xn = xbytes // 4
        x = torch.empty(xn, dtype=torch.float32, device="cuda").uniform_()
        idx = torch.randint(0, xn, (n,), dtype=torch.int32, device="cuda")
        for unroll in (4, 8):
            t = timeit(lambda: lib.ub_gather(vp(idx.data_ptr()), vp(val.data_ptr()), vp(x.data_ptr()), n, vp(out.data_ptr()), 16384, unroll, s))
            print(f"gather 100M idx+val, x {xbytes/1e6:8.1f} MB, unroll {unroll}: {t:.4f} ms = {n/t/1e6:.1f} Ggather/s", flush=True)
        del x, idx


def shapes():
    lib.ub3_stream.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp]
    out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
    a = torch.empty(100_000_000, device="cuda").uniform_()
    for mode in (0, 1):
        for blocks in (2048, 8192):
            t = timeit(lambda: lib.ub3_stream(vp(a.data_ptr()), 400_000_000, vp(out.data_ptr()), blocks, mode, s), 20)
            print(f"stream 400MB mode {mode} blocks {blocks}: {t:.4f} ms = {0.4/t:.2f} TB/s", flush=True)


def policy():
    lib.ub5.argtypes = [vp, ctypes.c_uint, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp]
    out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
    x = torch.empty(1 << 28, device="cuda").uniform_()     # 1 GiB
    names = ["plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"]
    blocks, iters = 8192, 6     # ~100M gathers
    total = blocks * 256 * iters * 8
    for xbytes in (1 << 20, 2 << 20, 4 << 20, 16 << 20, 128 << 20, 1 << 30):
        mask = xbytes // 4 - 1
        for pol in range(6):
            t = timeit(lambda: lib.ub5(vp(x.data_ptr()), mask, iters, vp(out.data_ptr()), blocks, pol, s), 5)
            print(f"table {xbytes >> 20:5d} MiB  {names[pol]:12s}: {t:.4f} ms  {total / t / 1e6:8.1f} G gathers/s  "
                  f"({total / t / 1e6 * 128 / 1e3:6.2f} TB/s of 128-byte lines)", flush=True)


def pattern():
    lib.ub6.argtypes = [vp, ctypes.c_uint, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    out = torch.zeros(4, device="cuda", dtype=torch.float64); s = vp(torch.cuda.current_stream().cuda_stream)
    x = torch.empty(1 << 22, device="cuda", dtype=torch.float64).uniform_()
    names = ["same addr", "consecutive", "stride-4 (chunk layout)", "quad-transposed (runs of 4)", "16-lane transposed (runs of 16)",
             "quad stride 2", "quad run of 4 permuted", "quad 0,1,2,2", "quad run at 4n+1", "quad run at 4n+2", "quad 0,1,2,9", "quad 0,1,8,9"]
    blocks, iters = 8192, 64
    total = blocks * 256 * iters * 8
    for dbl in (0, 1):
        for mode in range(len(names)):
            t = timeit(lambda: lib.ub6(vp(x.data_ptr()), 2047, iters, vp(out.data_ptr()), blocks, mode, dbl, s))
            print(f"{'f64' if dbl else 'f32'} table 2048 elems  {names[mode]:34s}: {total/t/1e6/256/2.4:6.2f} lanes/clk/CU@2.4GHz = "
                  f"{64/(total/t/1e6/256/2.4):5.1f} clk per wave gather", flush=True)


def banded():
    import merge_spmv_amd as M
    from merge_spmv_amd import generators as G
    A = G.uniform_csr(3_125_000, 3_125_000, 32, dtype=torch.float32)
    x = G.uniform_pm1(1, A.cols, torch.float32, "cuda")
    s = vp(torch.cuda.current_stream().cuda_stream)
    yref = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
    t_ref = timeit(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=yref, num_cols=A.cols), 20, 3)
    print(f"merge CsrMV (stateless): {t_ref:.4f} ms")
    for nb in (1, 2, 4, 8):
        bw = (A.cols + nb - 1) // nb
        splits = torch.empty((nb + 1) * A.rows, dtype=torch.int32, device="cuda")
        ypart = torch.empty(nb * A.rows, dtype=torch.float32, device="cuda")
        y = torch.empty(A.rows, dtype=torch.float32, device="cuda")
        t_s = timeit(lambda: lib.ub7_splits(vp(A.row_offsets.data_ptr()), vp(A.column_indices.data_ptr()), vp(splits.data_ptr()), A.rows, nb, bw, s), 20, 3)
        for rpb in (256, 1024):
            t_b = timeit(lambda: lib.ub7_banded(vp(A.values.data_ptr()), vp(A.column_indices.data_ptr()), vp(splits.data_ptr()), vp(x.data_ptr()),
                                                vp(ypart.data_ptr()), vp(y.data_ptr()), A.rows, nb, rpb, s), 20, 3)
            err = float((y.double() - yref.double()).abs().max())
            print(f"bands {nb}: finding the per-row band boundaries {t_s:.4f} ms, banded pass over the UNCHANGED CSR + combine ({rpb} rows/block) "
                  f"{t_b:.4f} ms, total {t_s + t_b:.4f} ms  (max diff vs merge {err:.2e})", flush=True)
    plan = M.CsrMVPlan(A.values, A.row_offsets, A.column_indices, A.cols)
    yp = torch.empty_like(yref)
    t_p = timeit(lambda: plan(x, yp), 20, 3)
    print(f"prepared band-major plan ({plan.bands} bands, matrix re-laid out once): {t_p:.4f} ms  (max diff vs merge {float((yp.double() - yref.double()).abs().max()):.2e})")


def scalar():
    """gather issued as scalar loads (64 per wave instruction slot): bytes per miss and issue rate vs the vector gather"""
    lib.ub5.argtypes = [vp, ctypes.c_uint, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp]
    lib.ub_gather_scalar.argtypes = [vp, ctypes.c_uint, ctypes.c_int, vp, ctypes.c_int, vp]
    out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
    x = torch.empty(1 << 28, device="cuda").uniform_()
    blocks, iters = 8192, 6
    total = blocks * 256 * iters * 8
    for xbytes in (64 << 10, 1 << 20, 4 << 20, 16 << 20, 128 << 20, 1 << 30):
        mask = xbytes // 4 - 1
        tv = timeit(lambda: lib.ub5(vp(x.data_ptr()), mask, iters, vp(out.data_ptr()), blocks, 0, s), 5)
        ts = timeit(lambda: lib.ub_gather_scalar(vp(x.data_ptr()), mask, iters, vp(out.data_ptr()), blocks, s), 5)
        print(f"table {xbytes / 2**20:8.2f} MiB: vector gather {tv:.4f} ms = {total / tv / 1e6:7.1f} G/s | scalar-load gather {ts:.4f} ms = {total / ts / 1e6:7.1f} G/s", flush=True)


PROBES = {"scalar": scalar, "stream": stream, "shapes": shapes, "policy": policy, "pattern": pattern, "banded": banded}
if __name__ == "__main__":
    for name in sys.argv[1:] or list(PROBES):
        print(f"## {name}", flush=True)
        PROBES[name]()
