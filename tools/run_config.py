#!/usr/bin/env python3
"""tools/run_config.py <pmc label> [--steps K] [--mode stateless|plan|hotcols] -- run ONE of bench.py's configurations K times
(the command tools/gpu_profile.sh puts under rocprofv3 to collect that configuration's counters; bench.py then replays
profiles/<dir>/pmc_latest.json for the matching label).  Labels: c2_f32 (the headline), plus those of bench.config_specs:
dense5 c2 circuit c3_web c3_orkut c4 dense32 c5; rmat24 = an R-MAT scale-24 / 250 M-edge matrix for the hot-column plan."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
import bench

ap = argparse.ArgumentParser()
ap.add_argument("label")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--mode", default="stateless", choices=["stateless", "plan", "hotcols"])
ap.add_argument("--save", default=None, help="generate the configuration's matrix, write its CSR image to this file (raw arrays) and exit")
ap.add_argument("--load", default=None, help="read the CSR image --save wrote instead of generating (host reads + copies, no kernels: what lets "
                                             "config 5 through rocprofv3 --pmc, which dies in the generation of its 2e9 edges)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
if args.load:
    A, x_seed = G.load_csr_image(args.load, dev)
elif args.label == "c2_f32":
    A = G.uniform_csr(bench.C2_ROWS_PER_GPU, bench.C2_ROWS_PER_GPU, bench.C2_NPR, dtype=torch.float32, device=dev); x_seed = G.SEED_C2 + 2
elif args.label == "rmat24":
    A = G.rmat_csr(24, 250_000_000, dtype=torch.float64, device=dev, seed=G.SEED_C5); x_seed = G.SEED_C5 + 2
else:
    spec = [s for s in bench.config_specs(torch, G, dev, args.steps) if s[1] == args.label]
    if not spec:
        raise SystemExit(f"unknown label {args.label}")
    A, x_seed = spec[0][5]()
if args.save:
    G.save_csr_image(A, x_seed, args.save)
    print(f"{args.label}: CSR image written to {args.save} ({os.path.getsize(args.save) / 1e9:.1f} GB)")
    raise SystemExit(0)
x = G.uniform_pm1(x_seed, A.cols, A.values.dtype, dev)
y = torch.empty(A.rows, dtype=A.values.dtype, device=dev)
if args.mode == "plan":
    plan = M.CsrMVPlan(A.values, A.row_offsets, A.column_indices, A.cols)
    call = lambda: plan(x, y)
elif args.mode == "hotcols":
    plan = M.CsrMVHotColumns(A.values, A.row_offsets, A.column_indices, A.cols)
    call = lambda: plan(x, y)
else:
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype, device=dev)
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
for _ in range(3): call()
torch.cuda.synchronize()
for _ in range(args.steps): call()
torch.cuda.synchronize()
print(f"{args.label} {args.mode}: rows {A.rows} nnz {A.nnz} {args.steps} steps done, |y|_1 = {float(y.abs().sum()):.6g}")
