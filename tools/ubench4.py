import ctypes, os
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libubench4.so"))
vp = ctypes.c_void_p
lib.ub4.argtypes = [vp, ctypes.c_uint, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp]
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
x = torch.empty(1 << 26, device="cuda").uniform_()
names = {0: "same addr", 1: "consecutive", 2: "stride-4 chunk layout", 3: "random/lane", 4: "runs of 4", 5: "runs of 8", 6: "runs of 16"}
blocks, iters = 8192, 64
total = blocks * 256 * iters * 8
for xbytes in (128, 8 << 10, 1 << 20, 16 << 20):
    mask = xbytes // 4 - 1
    for mode in range(7):
        t = timeit(lambda: lib.ub4(vp(x.data_ptr()), mask, iters, vp(out.data_ptr()), blocks, mode, s))
        print(f"x {xbytes:>9} B  {names[mode]:24s}: {t:.4f} ms  {total/t/1e6:8.1f} G lane-loads/s  = {total/t/1e6/256/2.1:.2f} lanes/clk/CU@2.1GHz", flush=True)
