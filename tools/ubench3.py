import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libubench3.so"))
vp = ctypes.c_void_p
lib.ub3_stream.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp]
lib.ub3_tiles.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = torch.zeros(4, device="cuda"); s = vp(torch.cuda.current_stream().cuda_stream)
a = torch.empty(100_000_000, device="cuda").uniform_(); b = torch.empty(100_000_000, device="cuda").uniform_()
for mode in (0, 1):
    for blocks in (2048, 8192):
        t = timeit(lambda: lib.ub3_stream(vp(a.data_ptr()), 400_000_000, vp(out.data_ptr()), blocks, mode, s))
        print(f"stream 400MB mode {mode} blocks {blocks}: {t:.4f} ms = {0.4/t:.2f} TB/s", flush=True)
for mode in (0, 1):
    for per4 in (448, 960, 4096):
        for blocks in (2048, 1280, 55804):
            t = timeit(lambda: lib.ub3_tiles(vp(a.data_ptr()), vp(b.data_ptr()), 400_000_000, per4, vp(out.data_ptr()), blocks, mode, s))
            print(f"tiles 2x400MB mode {mode} per_block {per4*16}B blocks {blocks}: {t:.4f} ms = {0.8/t:.2f} TB/s", flush=True)
