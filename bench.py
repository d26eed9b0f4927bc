#!/usr/bin/env python3
"""bench.py -- CsrMV throughput of the MI355X-native merge-based SpMV.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one y = A*x through the C ABI (include/mspmv.h) on the synthetic
CSR workload BASELINE.json's metric is quoted on for one GPU -- config C2:
fp32, 3 125 000 x 3 125 000, exactly 32 nnz/row = 100 000 000 nnz, uniform
random columns.  Inputs are resident in HBM before timing.
With N GPUs the matrix has N x 3 125 000 rows over the SAME 3 125 000 columns
(per-GPU rows, nonzeros and x working set all fixed: weak scaling; growing
the column space with N would instead measure how the x gather falls out of
cache), is merge-partitioned by diagonal across the ranks
(merge_spmv_amd/multi_gpu.py) and each step adds the one RCCL all-gather of the
boundary-row carries.  Rank 0 prints ONE JSON line.

value           = 2 * nnz_total / t  (GFLOP/s, whole job; reference formula gpu_spmv.cu:451-465)
roofline        = algorithmic (compulsory) bytes of one tile_kernel launch / its
                  average duration from hipEvents recorded on the launch stream
                  (mspmv_profile_begin/_end), against the 8 TB/s HBM3E peak
cpu_baseline    = the oracle's OpenMP merge-path port (oracle/merge_oracle.c,
                  restating cpu_spmv.cpp:292-353) on the same matrix on this
                  box's host cores, bounded sample; a checker timed beside the
                  product, never used by it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (rows per GPU, nnz per row, default dtype).  "c2" is the headline (BASELINE config 2,
    # uniform random columns); "dense32" is its reference-compatible pure-streaming variant
    # (--dense=32 --size=100000000, gpu_spmv.cu:645-650): same sizes, x has 32 entries.
    "c2": (3_125_000, 32, "f32"),
    "dense32": (3_125_000, 32, "f32"),
}


def algorithmic_bytes(rows, cols, nnz, vb):
    """every CSR array, x and y touched exactly once."""
    return nnz * (vb + 4) + (rows + 1) * 4 + rows * vb + cols * vb


def effective_bytes(rows, nnz, vb):
    """the reference's byte model, gpu_spmv.cu:452-456"""
    return nnz * (2 * vb + 4) + rows * (4 + vb)


def cpu_baseline(A, x, budget_s=12.0, max_iters=40):
    """Time the oracle port on the host cores (rank 0, N = 1 only)."""
    import numpy as np
    from oracle import oracle as O
    csr = O.Csr(A.rows, A.cols, A.row_offsets.cpu().numpy(), A.column_indices.cpu().numpy(),
                A.values.cpu().numpy())
    xh = x.cpu().numpy()
    threads = O.max_threads()
    O.omp_merge_csrmv(csr, xh, threads)                      # warm-up (cf. cpu_spmv.cpp:390-392)
    t0 = time.perf_counter(); iters = 0
    while iters < max_iters and (time.perf_counter() - t0) < budget_s:
        O.omp_merge_csrmv(csr, xh, threads); iters += 1
    dt = (time.perf_counter() - t0) / max(iters, 1)
    return {"value": round(2.0 * csr.nnz / dt / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"same C2 matrix ({csr.nnz} nnz), {iters} OpenMP merge-path SpMVs, {dt * 1e3:.2f} ms each",
            "effective_GBs": round(effective_bytes(csr.rows, csr.nnz, csr.values.dtype.itemsize) / dt / 1e9, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", default=None, help="development: BLOCKxIPT[:flags] passed to mspmv_set_tuning")
    args = ap.parse_args()

    import torch
    import merge_spmv_amd as M
    from merge_spmv_amd import generators as G, multi_gpu as MG
    import numpy as np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP merge-path kernels have no CPU fallback)")
    # MSPMV_BENCH_ONE_DEVICE=1 + MSPMV_BENCH_BACKEND=gloo: exercise the multi-rank path on a
    # single-GPU box (all ranks on cuda:0, carries exchanged through gloo) -- a functional
    # check of the sharding / exchange code, not a measurement
    if os.environ.get("MSPMV_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    M.load_library()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MSPMV_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    rows_per_gpu, npr, default_dtype = WORKLOADS[args.workload]
    dtype_name = args.dtype or default_dtype
    if args.tune:
        shape, _, fl = args.tune.partition(":")
        b, _, i = shape.partition("x")
        M.set_tuning(4 if dtype_name == "f32" else 8, int(b or 0), int(i or 0), int(fl or "0", 0))
    tdt = torch.float32 if dtype_name == "f32" else torch.float64
    vb = 4 if dtype_name == "f32" else 8
    rows = rows_per_gpu * world
    cols = rows_per_gpu if args.workload == "c2" else npr
    nnz_total = rows * npr
    if args.workload != "c2" and world > 1:
        raise SystemExit("only the c2 workload is sharded across GPUs")

    # ---- build this rank's swath directly in HBM -------------------------
    if world == 1:
        # the plain drop-in call: no shard wrapper, no collective
        A = (G.uniform_csr(rows, cols, npr, dtype=tdt, device=dev) if args.workload == "c2"
             else G.dense_csr(rows, npr, dtype=tdt, device=dev, ones=False))
        local_rows, local_nnz = rows, nnz_total
        ws = M.CsrMVWorkspace(rows, nnz_total, tdt, device=dev)
        y = torch.empty(rows, dtype=tdt, device=dev)

        def op(xv):
            return M.csrmv(A.values, A.row_offsets, A.column_indices, xv, y=y, num_cols=cols, workspace=ws)
    else:
        shard = MG.uniform_shard(rows, cols, npr, rank, world, tdt, device=dev)
        local_rows, local_nnz = shard.local_rows, shard.local_nnz
        op = MG.ShardedCsrMV(shard, group=None)
    x = G.uniform_pm1(G.SEED_C2 + 2, cols, tdt, dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        op(x)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        op(x)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- per-kernel durations: the same K steps again with hipEvents on the launch stream
    M.profile_begin(args.steps)
    for _ in range(args.steps):
        op(x)
    torch.cuda.synchronize()
    prof = M.profile_end()

    if rank == 0:
        gflops = 2.0 * nnz_total / (ms_per_step * 1e-3) / 1e9
        info = M.launch_info(local_rows, local_nnz, vb)
        b_alg = algorithmic_bytes(local_rows, cols, local_nnz, vb)
        tile_s = prof["tile_ms"] * 1e-3
        achieved = b_alg / tile_s / 1e9 if tile_s > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("workload") == args.workload and pmc.get("dtype") == dtype_name and world == 1:
                    traffic = pmc.get("tile_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "CsrMV GFLOP/s", "value": round(gflops, 3), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": (f"C2 uniform CSR: {rows} x {cols}, {npr} nnz/row, {nnz_total} nnz "
                                    f"({rows_per_gpu * npr} nnz per GPU), uniform random sorted columns, values/x in [-1,1)")
                       if args.workload == "c2" else
                       f"dense {rows} x {npr} as CSR ({nnz_total} nnz): the streaming variant of C2 (--dense=32 --size=100000000)",
                       "tile": f"{info['block_threads']}x{info['items_per_thread']}",
                       "partition": "single GPU" if world == 1 else f"merge-path diagonal split over {world} GPUs + 1 RCCL all-gather of carries"},
            "effective_GBs_reference_formula": round(effective_bytes(rows, nnz_total, vb) / (ms_per_step * 1e-3) / 1e9, 2),
            "compulsory_GBs": round(algorithmic_bytes(rows, cols, nnz_total, vb) / (ms_per_step * 1e-3) / 1e9, 2),
            "roofline": {"bound": "hbm", "kernel": "tile_kernel_vec", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "algorithmic_bytes_per_launch": b_alg,
                         "kernel_ms": {"search": round(prof["search_ms"], 5), "tile": round(prof["tile_ms"], 5),
                                       "fixup": round(prof["fixup_ms"], 5)},
                         "events": f"hipEvents on the launch stream, {prof['calls']} launches"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, x)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
