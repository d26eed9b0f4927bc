import ctypes, os
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libubench6.so"))
vp = ctypes.c_void_p
lib.ub6.argtypes = [vp, ctypes.c_uint, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = torch.zeros(4, device="cuda", dtype=torch.float64); s = vp(torch.cuda.current_stream().cuda_stream)
x = torch.empty(1 << 22, device="cuda", dtype=torch.float64).uniform_()
names = ["same addr", "consecutive", "stride-4 (chunk layout)", "quad-transposed (runs of 4)", "16-lane transposed (runs of 16)", "quad stride 2", "quad run of 4 permuted", "quad 0,1,2,2", "quad run at 4n+1", "quad run at 4n+2", "quad 0,1,2,9", "quad 0,1,8,9"]
blocks, iters = 8192, 64
total = blocks * 256 * iters * 8
for dbl in (0,):
    for elems in (2048,):
        for mode in range(len(names)):
            t = timeit(lambda: lib.ub6(vp(x.data_ptr()), elems - 1, iters, vp(out.data_ptr()), blocks, mode, dbl, s))
            print(f"{'f64' if dbl else 'f32'} table {elems:7d} elems  {names[mode]:34s}: {total/t/1e6/256/2.4:6.2f} lanes/clk/CU@2.4GHz = {64/(total/t/1e6/256/2.4):5.1f} clk per wave gather", flush=True)
