import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libubench2.so"))
vp = ctypes.c_void_p
lib.ub_gather_aux.argtypes = [vp, vp, vp, ctypes.c_uint, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
n = 100_000_000
val = torch.empty(n, device="cuda").uniform_()
for xbytes in (3 << 20, 12_500_000, 100_000_000):
    xn = xbytes // 4
    x = torch.empty(xn, device="cuda").uniform_()
    idx = torch.randint(0, xn, (n,), dtype=torch.int32, device="cuda")
    for aux in (0, 1, 2, 3, 16, 17, 18, 19):
        t = timeit(lambda: lib.ub_gather_aux(vp(idx.data_ptr()), vp(val.data_ptr()), vp(x.data_ptr()), xn * 4, n, vp(out.data_ptr()), 16384, aux, s))
        print(f"x {xbytes/1e6:7.1f} MB aux {aux:2d} (nt stream loads): {t:.4f} ms", flush=True)
