#!/usr/bin/env python3
"""tools/record_soak.py [seconds] [seed] -- a soak of the tagged-record protocol of the one-launch kernel (mspmv_kernels.hpp: "EVERY
RECORD SLOT IS CLEAN": announce / record / take, cancel by atomic exchange) under the conditions it exists for: random matrices of
long rows (rows over a few to hundreds of tiles, back to back, between runs of short rows), random poll budgets (the default, one
look, a handful, never), streams restricted to a random fraction of the CUs (the residency the dispatcher assumes is then several
times what the stream has), a GEMM hogging the device now and then, and CAPTURED calls replayed with new x after such episodes.
After every call: every row against an fp64 reference under the strict bound, and THE WHOLE RECORD REGION OF THE TEMP STORAGE ZERO.
Every wait in here is bounded by a watchdog: a call that does not return within 60 s aborts the soak."""
import ctypes, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import merge_spmv_amd as M

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
hip = ctypes.CDLL("libamdhip64.so")
cus = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(fraction):
    words = (cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in range(max(int(cus * fraction), 8)):
        mask[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    return st if hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask) == 0 else None


def watchdog(seconds, what):
    done = threading.Event()

    def run():
        if not done.wait(seconds):
            sys.stderr.write(f"record_soak: WATCHDOG -- {what} did not return within {seconds} s\n"); sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=run, daemon=True).start()
    return done


def random_matrix(f32):
    vb = 4 if f32 else 8
    tile = M.launch_info(100000, 1000000, vb)["tile_items"]
    pieces = []
    for _ in range(int(rng.integers(1, 6))):
        kind = rng.integers(0, 4)
        if kind == 0: pieces.append(rng.integers(0, 6, int(rng.integers(10, 4000))))
        elif kind == 1: pieces.append([int(rng.integers(1, 300)) * tile // int(rng.integers(1, 4)) + int(rng.integers(0, 500))])
        elif kind == 2: pieces.append([int(rng.integers(200, 3000))] * int(rng.integers(1, 40)))
        else: pieces.append(np.zeros(int(rng.integers(1, 3000)), np.int64))
    lens = np.concatenate([np.asarray(p, np.int64) for p in pieces])
    if lens.sum() > 60_000_000: lens = np.minimum(lens, 20_000_000)
    return lens


cases = episodes = replays = 0
worst = 0.0
t_end = time.time() + budget
while time.time() < t_end:
    f32 = bool(rng.integers(0, 2)); tdt = torch.float32 if f32 else torch.float64; vb = 4 if f32 else 8
    lens = random_matrix(f32)
    rows = lens.size; cols = int(rng.integers(1, 200000))
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:]); nnz = int(off[-1])
    if rows + nnz > 2**31 - 70000 or nnz == 0: continue
    col = torch.randint(0, cols, (nnz,), device="cuda", dtype=torch.int32)
    val = (torch.rand(nnz, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    offs = torch.from_numpy(off.astype(np.int32)).cuda()
    lens_i = torch.from_numpy(lens).cuda()
    info = M.launch_info(rows, nnz, vb)
    ws = M.CsrMVWorkspace(rows, nnz, tdt); ws.buffer.zero_()
    rec_lo, rec_n = info["records_offset"], (info["num_tiles"] + info["num_tiles"] // 64 + 1) * 16
    diag = info["diag_offset"]
    eps = 2.0 ** -24 if f32 else 2.0 ** -53
    cfac = 2.0 * (torch.ceil(torch.log2(lens_i.double() + 1)) + 16 + 8 + 2)
    st = masked_stream(float(rng.choice([0.05, 0.125, 0.25, 0.5]))) if rng.random() < 0.6 else None
    stream_arg = st.value if st is not None else None
    x = (torch.rand(cols, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    y = torch.empty(rows, dtype=tdt, device="cuda")

    def check(tag):
        global worst
        prod = val.double() * x.double()[col.long()]
        g = torch.segment_reduce(prod, "sum", lengths=lens_i, axis=0, unsafe=True)
        s = torch.segment_reduce(prod.abs(), "sum", lengths=lens_i, axis=0, unsafe=True)
        err = (y.double() - g).abs(); tol = cfac * eps * s
        bad = (err > tol) | ((lens_i == 0) & (y != 0)) | ~torch.isfinite(y)
        if bool(bad.any()):
            print(f"MISMATCH seed={seed} case={cases} {tag}: f32={f32} rows={rows} nnz={nnz} bad={int(bad.sum())}", flush=True); os._exit(1)
        left = int(ws.buffer[rec_lo: rec_lo + rec_n].view(torch.int64).count_nonzero().item())
        if left:
            print(f"RECORDS NOT CLEAN seed={seed} case={cases} {tag}: {left} nonzero words of {rec_n // 8}; f32={f32} rows={rows} nnz={nnz}", flush=True); os._exit(2)
        worst = max(worst, float((err / (tol + 1e-300)).max()))

    try:
        for step in range(int(rng.integers(2, 6))):
            polls = int(rng.choice([0, 0, 1, 1, 3, 40, -1]))
            M.set_record_polls(polls)
            hogging = rng.random() < 0.15
            if hogging:
                a = torch.randn(4096, 4096, device="cuda")
                with torch.cuda.stream(torch.cuda.Stream()):
                    for _ in range(4): a = torch.tanh(a @ a) * 0.01
            x.copy_((torch.rand(cols, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt))
            torch.cuda.synchronize()
            e0 = int(ws.buffer[diag + 4: diag + 8].view(torch.int32).item())
            done = watchdog(60, f"csrmv (case {cases}, step {step}, polls {polls}, masked {st is not None})")
            y.fill_(float("nan"))
            M.csrmv(val, offs, col, x, y=y, num_cols=cols, workspace=ws, stream=stream_arg)
            if st is not None: assert hip.hipStreamSynchronize(st) == 0
            torch.cuda.synchronize(); done.set()
            episodes += int(ws.buffer[diag + 4: diag + 8].view(torch.int32).item()) - e0
            check(f"step {step} polls {polls}")
        # a captured call replayed with new x after whatever the steps above left behind
        if rng.random() < 0.4:
            M.set_record_polls(0)
            g = torch.cuda.CUDAGraph(); s2 = torch.cuda.Stream()
            call = lambda: M.csrmv(val, offs, col, x, y=y, num_cols=cols, workspace=ws)
            with torch.cuda.stream(s2):
                call()
                with torch.cuda.graph(g, stream=s2): call()
            for k in range(3):
                if k == 1:                                   # a recomputing episode between two replays
                    M.set_record_polls(-1); call(); torch.cuda.synchronize(); M.set_record_polls(0)
                x.copy_((torch.rand(cols, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt))
                done = watchdog(60, f"graph replay (case {cases})")
                g.replay(); torch.cuda.synchronize(); done.set()
                check(f"replay {k}"); replays += 1
    finally:
        M.set_record_polls(0)
        if st is not None: hip.hipStreamDestroy(st)
    cases += 1
print(f"record_soak: {cases} matrices in {budget:.0f} s, {episodes} recomputing episodes, {replays} graph replays: every row within the strict bound "
      f"(worst |err|/bound = {worst:.3f}), the record region zero after every call")
