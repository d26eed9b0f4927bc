import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libubench5.so"))
vp = ctypes.c_void_p
lib.ub5.argtypes = [vp, ctypes.c_uint, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp]
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
x = torch.empty(1 << 28, device="cuda").uniform_()     # 1 GiB
names = ["plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"]
blocks, iters = 8192, 6     # ~100M gathers
total = blocks * 256 * iters * 8
only = [int(a) for a in sys.argv[1:]] or range(6)
for xbytes in (1 << 20, 16 << 20, 1 << 30):
    mask = xbytes // 4 - 1
    for pol in only:
        t = timeit(lambda: lib.ub5(vp(x.data_ptr()), mask, iters, vp(out.data_ptr()), blocks, pol, s))
        print(f"table {xbytes >> 20:5d} MiB  {names[pol]:12s}: {t:.4f} ms  {total / t / 1e6:8.1f} G gathers/s", flush=True)
