#!/usr/bin/env python3
"""bench.py -- CsrMV throughput of the MI355X-native merge-based SpMV.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... -- or
     plainly, in which case bench.py starts that launcher itself: self_launch)

A "step" is one y = A*x through the C ABI (include/mspmv.h); inputs are resident in HBM before timing.

N = 1 (default workload "c2"): BASELINE.json config 2, the configuration the metric is quoted on for one
GPU -- fp32, 3 125 000 x 3 125 000, exactly 32 nnz/row = 100 000 000 nnz, uniform random sorted columns
 -- through the stateless drop-in call mspmv_csrmv_f32 (which, on a matrix like this one, lets the
tile kernel run its column-band passes: `column_band_passes` in the roofline record says whether and how many;
DESIGN.md 4).  The same line carries a
`prepared_plan` sub-record: the opt-in band-major plan (mspmv_csrmv_plan_*; set-up reported separately,
like the reference reports the HYB conversion, gpu_spmv.cu:106-257) on the same matrix.

N > 1 (default workload "c5"): BASELINE.json config 5 -- fp64 R-MAT scale 26 (67 108 864^2), 2 000 000 000
edges, ONE matrix independent of N, merge-partitioned by diagonal into N swaths (mspmv_mg_partition);
every rank builds its swath in its own HBM from the counter-based generator and drives it through the C
multi-GPU operator (mspmv_mg_plan_*: the part's CsrMV launches + ONE RCCL all-gather of the N
boundary-row carries + the owner's add, all below the C ABI).  Strong scaling: `value` = 2 * nnz_total /
max-over-ranks time; `per_rank` carries the spread of the ranks' tile-kernel and step times, `hot_column_plan` the same job
with every rank's columns renumbered by reference count (mspmv_mg_plan_hot_columns: opt-in, set-up apart).  Rank 0 then also
runs the WHOLE matrix alone on its GPU in the same job (`single_gpu_same_workload`, 5 steps; --no-single-gpu-leg skips it), so
the line is self-contained for an efficiency figure.
"c2" can also be run sharded (--workload c2 --gpus N: weak scaling, N x 3 125 000 rows over the same
3 125 000 columns); "dense32" is C2's pure-streaming variant (--dense=32 --size=100000000).

value           = 2 * nnz_total / t  (GFLOP/s, whole job; reference formula gpu_spmv.cu:451-465)
roofline        = algorithmic (compulsory) bytes of one tile_kernel launch / its average duration from
                  hipEvents recorded on the launch stream (mspmv_profile_begin/_end).  `bound` / `resident` say what that rate is
                  read against: "hbm" -- the 8 TB/s HBM3E spec peak -- when the call's arrays are beyond the 256 MB Infinity Cache;
                  "infinity_cache" when they fit it (they then stay there between SpMVs): `peak` is then the rate of a bare
                  16-byte-per-lane read stream over a buffer of the same size MEASURED IN THIS RUN (mspmv_probe_read_stream),
                  `frac_of_hbm_spec_peak` kept beside it.  `traffic` = L2 <-> fabric bytes of the tile kernel from two live
                  rocprofv3 --pmc child runs (Infinity-Cache hits included: fabric traffic, not HBM traffic, for a resident record)
sampled_check   = an untimed correctness witness of the very record: 2^16 seeded rows + the first, last and longest recomputed in fp64
                  with torch gathers against the stated bound (no oracle import; parity proper is tests/ -m gpu); < 1 passes
configs         = (N = 1, default workload) one sub-record per remaining single-GPU configuration of BASELINE.json, each
                  timed the same way on its own synthetic matrix -- config 1's --dense=5 matrix in fp64 with the product's
                  cpu_spmv kernel on the host cores beside it (`cpu`), C2 in fp64 (the reference's default precision,
                  gpu_spmv.cu:727-735), a circuit5M-SHAPED stand-in (the matrix of the reference's one published number),
                  config 3's two matrices as size-matched R-MAT stand-ins (the SuiteSparse files cannot be fetched
                  offline), config 4, config 5 on ONE GPU, and the reference's own --dense=32 streaming input -- with
                  ms_per_step, GFLOP/s, the reference's effective-bandwidth share of peak (gpu_spmv.cu:452-465), the tile
                  kernel's roofline fraction, live counter traffic (config 5 too: its CSR image is parked in /dev/shm for the two
                  child runs), the sampled check, and rocSPARSE csrmv on the same arrays (`vendor`).  --mtx-dir DIR (default
                  $MSPMV_C3_DIR): webbase-1M.mtx / com-Orkut.mtx / circuit5M.mtx found there replace the generated stand-ins,
                  through the product's Matrix Market ingest; `data` says whether a record ran on the SuiteSparse file, on a
                  stand-in file (tools/make_standin_mtx.py) or on a generated stand-in.
cpu_baseline    = the PRODUCT's OpenMP merge-path kernel (merge_spmv_amd/host/merge_csrmv.hpp, what cpu_spmv
                  runs; pinned bit for bit against the oracle by tests/test_cpu_product_parity.py) on the
                  same matrix on this box's host cores: private first-touched arrays, threads = the cgroup
                  CPU quota, bound to distinct cores of socket 0 when the cpuset allows; bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20      # the die-level L3 (same guide): a call whose arrays fit it is not bound by HBM once they are in there
TRAFFIC_IS = ("L2 <-> fabric bytes of the tile kernel (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction): requests the L2s send to the "
              "memory side -- Infinity-Cache hits INCLUDED, so for a record whose arrays live in that cache this is fabric traffic, not HBM traffic")

WORKLOADS = {
    # name: default dtype.  "c2" is the N = 1 headline (BASELINE config 2); "dense32" its reference-compatible
    # pure-streaming variant (--dense=32 --size=100000000, gpu_spmv.cu:645-650); "c5" = BASELINE config 5.
    "c2": "f32",
    "dense32": "f32",
    "c5": "f64",
}
C2_ROWS_PER_GPU, C2_NPR = 3_125_000, 32


def algorithmic_bytes(rows, cols, nnz, vb):
    """every CSR array, x and y touched exactly once."""
    return nnz * (vb + 4) + (rows + 1) * 4 + rows * vb + cols * vb


def effective_bytes(rows, nnz, vb):
    """the reference's byte model, gpu_spmv.cu:452-456"""
    return nnz * (2 * vb + 4) + rows * (4 + vb)


_CACHE_RATE = {}


def roofline_bound(M, b_alg, achieved_gbs):
    """The fields of a `roofline` record that say WHAT the achieved rate is read against.  Arrays beyond the 256 MB Infinity Cache:
    the HBM3E spec peak.  Arrays that fit it (they stay there from SpMV to SpMV): the HBM peak is not the bound -- `peak` is then the
    rate of a bare 16-byte-per-lane read stream over a buffer of the same size, measured in this run on this box
    (mspmv_probe_read_stream), `frac` is against THAT, and `frac_of_hbm_spec_peak` keeps the old figure beside it, labelled."""
    if b_alg > INFINITY_CACHE_BYTES:
        return {"bound": "hbm", "resident": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "peak_source": "HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (its measured float4-copy rate is 6.29 TB/s = 0.79 of it)"}
    key = int(b_alg) >> 20
    if key not in _CACHE_RATE:
        try:
            _CACHE_RATE[key] = M.cache_stream_rate(b_alg)
        except Exception as e:  # noqa: BLE001
            _CACHE_RATE[key] = None
            sys.stderr.write(f"cache stream probe failed: {e}\n")
    peak = _CACHE_RATE[key]
    rec = {"bound": "infinity_cache", "resident": "infinity_cache", "unit": "GB/s", "frac_of_hbm_spec_peak": round(achieved_gbs / HBM_PEAK_GBS, 4),
           "note": f"the call's arrays ({b_alg / 2**20:.0f} MB) fit the 256 MB Infinity Cache and stay there between SpMVs: a cache-bandwidth figure, not an HBM one"}
    if peak:
        rec.update({"peak": round(peak, 1), "frac": round(achieved_gbs / peak, 4),
                    "peak_source": f"measured in this run: bare 16-byte-per-lane read stream (mspmv_probe_read_stream) over a {b_alg / 2**20:.0f} MB buffer resident in the Infinity Cache"})
    else:
        rec.update({"peak": None, "frac": None, "peak_source": "cache stream probe unavailable"})
    return rec


def _cgroup_cpu_stat():
    """(nr_throttled, throttled_usec) of this container's CPU controller: a quota-sized OpenMP team that is descheduled as
    a whole shows up here, which is what made the pinned figure swing between runs."""
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, _, v = line.partition(" ")
            out[k] = int(v)
    except (OSError, ValueError):
        pass
    return out.get("nr_throttled"), out.get("throttled_usec")


def cpu_baseline(A, x, label, budget_s=12.0, max_iters=40):
    """Time the product's OpenMP merge-path kernel on the host cores (rank 0, N = 1 only): three team shapes --
    quota-sized and bound to socket 0, quota-sized and unbound, two threads under the quota and unbound (so that the
    process's other threads do not push the cgroup over its quota) -- each with the cgroup's throttling counters read
    around the timed loop.  `value` is the fastest; all three are reported."""
    import numpy as np
    H = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
    vp, i = ctypes.c_void_p, ctypes.c_int
    H.mspmv_host_usable_cpus.restype = i
    H.mspmv_host_hardware_threads.restype = i
    off = A.row_offsets.cpu().numpy(); col = A.column_indices.cpu().numpy(); val = A.values.cpu().numpy()
    xh = x.cpu().numpy()
    f32 = val.dtype == np.float32
    fn = H.mspmv_host_merge_csrmv_bench_f32 if f32 else H.mspmv_host_merge_csrmv_bench_f64
    fn.restype = i
    fn.argtypes = [i, i, i, i, i, vp, vp, vp, vp, ctypes.c_double, i, vp, vp, vp, vp, vp]
    quota_threads = int(H.mspmv_host_usable_cpus())
    nnz = int(val.size)
    variants = [("pinned", quota_threads, 1), ("unpinned", quota_threads, 0)]
    if quota_threads > 4:
        variants.append(("unpinned_below_quota", quota_threads - 2, 0))
    runs = []
    for name, threads, pin in variants:
        avg = ctypes.c_double(); iters = ctypes.c_int(); pinned = ctypes.c_int(); packages = ctypes.c_int()
        thr0 = _cgroup_cpu_stat()
        st = fn(threads, pin, A.rows, A.cols, nnz, off.ctypes.data, col.ctypes.data, val.ctypes.data, xh.ctypes.data,
                float(budget_s) / len(variants), int(max_iters), ctypes.byref(avg), ctypes.byref(iters), ctypes.byref(pinned),
                ctypes.byref(packages), None)
        thr1 = _cgroup_cpu_stat()
        if st != 0:
            runs.append({"variant": name, "error": f"mspmv_host_merge_csrmv_bench returned {st}"})
            continue
        dt = avg.value * 1e-3
        runs.append({"variant": name, "threads": threads, "bound_to_socket0_cores": bool(pinned.value), "sockets_visible": packages.value,
                     "ms": round(avg.value, 3), "iters": iters.value, "value": round(2.0 * nnz / dt / 1e9, 3),
                     "cgroup_throttled_periods_during_run": None if thr0[0] is None else thr1[0] - thr0[0],
                     "cgroup_throttled_ms_during_run": None if thr0[1] is None else round((thr1[1] - thr0[1]) / 1e3, 1)})
    good = [r for r in runs if "value" in r]
    if not good:
        return {"error": "; ".join(r.get("error", "?") for r in runs)}
    best = max(good, key=lambda r: r["value"])
    quota = "unlimited"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = "unlimited" if q == "max" else f"{int(q) / int(per):g} CPUs (cpu.max {q} {per})"
    except (OSError, ValueError):
        pass
    try:
        cpuset = open("/sys/fs/cgroup/cpuset.cpus.effective").read().strip()
    except OSError:
        cpuset = "?"
    return {"value": best["value"], "unit": "GFLOP/s", "cores": best["threads"], "kind": "port",
            "value_is": best["variant"] + " (the fastest of `runs`)",
            "kernel": "merge_spmv_amd/host/merge_csrmv.hpp (product OpenMP merge-path CsrMV; -O3 -march=x86-64-v3 -ffp-contract=off)",
            "sample": f"{label} ({nnz} nnz), {best['iters']} SpMVs after 4 warm-ups, {best['ms']:.2f} ms each",
            "runs": runs,
            "hardware_threads": int(H.mspmv_host_hardware_threads()), "cpu_quota": quota, "cpuset": cpuset,
            "first_touch": "every thread first-touches the swath of the arrays it streams",
            "effective_GBs": round(effective_bytes(A.rows, nnz, val.dtype.itemsize) / (best["ms"] * 1e-3) / 1e9, 2)}


def time_stateless(M, torch, A, x, steps, warmup):
    """(ms per SpMV by the wall clock around `steps` back-to-back calls, per-kernel averages from hipEvents over `steps` more)"""
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype, device=A.values.device)
    y = torch.empty(A.rows, dtype=A.values.dtype, device=A.values.device)
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    for _ in range(max(warmup, 1)):
        call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    M.profile_begin(steps)
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    return ms, M.profile_end(), ws, y


def replayed_traffic(workload, dtype_name):
    """L2 <-> fabric bytes (Infinity-Cache hits included) per launch of the tile kernel from the committed rocprofv3 --pmc passes of the same workload
    (profiles/*/pmc_latest.json, written by tools/gpu_profile.sh): replayed constants, labelled as such.  The newest
    round's file wins."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_latest.json")) + [os.path.join(ROOT, "profiles", "pmc_latest.json")]):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        if pmc.get("workload") == workload and pmc.get("dtype") == dtype_name and pmc.get("tile_kernel_hbm_bytes_per_launch"):
            best = (pmc, os.path.relpath(path, ROOT))
    if best is None:
        return None, None
    pmc, rel = best
    return pmc["tile_kernel_hbm_bytes_per_launch"], (f"{TRAFFIC_IS}; {rel} (replayed, NOT measured in this run): FETCH_SIZE / WRITE_SIZE of the tile kernel "
                                                      "from separate rocprofv3 --pmc passes over this workload, " + str(pmc.get("collected", "see profiles/README.md")))


_LIVE_PMC = {"ok": True, "why": ""}       # one failure (no rocprofv3, a crash, a time-out) turns the live passes off for the rest of the run


def live_traffic(label, steps=24, timeout_s=60, extra_args=()):
    if not _LIVE_PMC["ok"]:
        return None, "live counters switched off after an earlier failure in this run: " + _LIVE_PMC["why"]
    tr, why = _live_traffic(label, steps, timeout_s, extra_args)
    if tr is None:
        _LIVE_PMC["ok"] = False; _LIVE_PMC["why"] = why
    return tr, why


def _live_traffic(label, steps, timeout_s, extra_args=()):
    """L2 <-> fabric bytes (Infinity-Cache hits included) per launch of the tile kernel, MEASURED IN THIS RUN on this box: two separate `rocprofv3 --kernel-trace --pmc`
    passes (FETCH_SIZE, then WRITE_SIZE: never combined with other trace domains) over `tools/run_config.py <label>`, which runs
    the same call on the same synthetic matrix in a child process; corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes
    (KB -> bytes, and gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes: doubled).  None when rocprofv3 is missing, fails
    or takes longer than `timeout_s` per pass -- the caller then falls back to the committed passes (replayed_traffic)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="mspmv_pmc_", dir="/tmp")
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    got = {}
    try:
        for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, pmc)
            # (counters only for the tile kernel -- --kernel-include-regex: the generation of the matrix in the child runs unprofiled,
            #  which is what lets config 5's 36 GB through --; the child in a session of its own, so that a time-out takes the
            #  grandchild python along instead of leaving it on the GPU beside the configurations timed next)
            cmd = [exe, "--kernel-trace", "--pmc", pmc, "--kernel-include-regex", "tile_kernel", "--output-format", "csv", "-d", out, "-o", "b", "--",
                   sys.executable, os.path.join(ROOT, "tools", "run_config.py"), label, "--steps", str(steps), *extra_args]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.communicate()
                return None, f"rocprofv3 --pmc {pmc} took longer than {timeout_s} s (its process group was killed)"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if proc.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {pmc} failed (rc {proc.returncode})"
            n, total = 0, 0.0
            for row in csv.DictReader(open(files[0])):
                if "tile_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == pmc:
                    n += 1; total += float(row.get("Counter_Value", 0) or 0)
            if n == 0:
                return None, f"no tile-kernel dispatch in the {pmc} pass"
            got[pmc] = (total / n, n)
    except Exception as e:  # noqa: BLE001 - measurement garnish: never at the price of the line
        return None, f"{type(e).__name__}: {e}"[:200]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fetch_kb, nf = got["FETCH_SIZE"]; write_kb, nw = got["WRITE_SIZE"]
    return int((2.0 * fetch_kb + write_kb) * 1024), (f"{TRAFFIC_IS}; measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/run_config.py {label}, "
                                                    f"average of {nf} / {nw} dispatches of the tile kernel; FETCH_SIZE {fetch_kb:.0f} KB x 1024 x 2 (gfx950 correction) + WRITE_SIZE {write_kb:.0f} KB x 1024")


REFERENCE_PUBLISHED_PCT = 62.96      # circuit5M fp64 on a K40: 181.6 effective GB/s of 288.4 (README.md:116,137-138)


def effective_record(rows, nnz, vb, ms):
    """the reference's own headline figure for one method (gpu_spmv.cu:452-465): bytes of its model / time, as a share of the
    device's peak memory bandwidth"""
    gbs = effective_bytes(rows, nnz, vb) / (ms * 1e-3) / 1e9
    return {"effective_GBs": round(gbs, 2), "effective_pct_of_peak": round(100.0 * gbs / HBM_PEAK_GBS, 2)}


def vendor_record(torch, A, x, y_ours, iters):
    """rocSPARSE csrmv on the same device arrays -- the column the reference always prints beside its own (cuSPARSE there,
    gpu_spmv.cu:262-364,565-578): analysis time apart, average SpMV time, and ours / theirs."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import rocsparse_ref
        ana, avg, yr = rocsparse_ref.time_csrmv(A, x, iters=iters)
        rec = {"library": "rocSPARSE csrmv (rocsparse_[sd]csrmv after rocsparse_[sd]csrmv_analysis), via tools/rocsparse_ref.py",
               "analysis_ms": round(ana, 3), "ms_per_step": round(avg, 5), "steps": iters, "value": round(2.0 * A.nnz / (avg * 1e-3) / 1e9, 3), "unit": "GFLOP/s"}
        if y_ours is not None:
            rec["max_abs_diff_vs_ours"] = float((yr.double() - y_ours.double()).abs().max().item())
        del yr
        return rec
    except Exception as e:  # noqa: BLE001 - the comparison column must never cost the headline
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def config_specs(torch, G, dev, steps):
    """(name, pmc label, description, dtype, steps, make) of every single-GPU configuration besides the headline -- shared with
    tools/run_config.py, which profiles one of them under rocprofv3."""
    f32, f64 = torch.float32, torch.float64
    return [
        ("C1 dense5 fp64", "dense5", "BASELINE config 1's matrix: the reference's --dense=5 (cpu_spmv.cpp:581-587: (1 << 24) / 5 rows x 5 columns, every entry 1.0), fp64 -- "
         "the GPU beside the product's OpenMP merge-path kernel on this box's host cores (`cpu`)", f64, max(steps, 100),
         lambda: (G.dense_csr((1 << 24) // 5, 5, dtype=f64, device=dev, ones=True), G.SEED_C2 + 2)),
        ("C2 fp64", "c2", "BASELINE config 2's matrix in the reference's default precision (gpu_spmv.cu:727-735)", f64, steps,
         lambda: (G.uniform_csr(C2_ROWS_PER_GPU, C2_ROWS_PER_GPU, C2_NPR, dtype=f64, device=dev), G.SEED_C2 + 2)),
        ("circuit5M-shaped stand-in", "circuit", f"the matrix of the reference's one published number (README.md:116,137-138: circuit5M, {G.CIRCUIT5M_ROWS}^2, "
         f"{G.CIRCUIT5M_NNZ} nonzeros, fp64, 62.96 % of the K40's peak in its effective-bandwidth metric); STAND-IN of exactly those sizes with a circuit "
         "matrix's row-length spread (generators.circuit_csr): the SuiteSparse file cannot be fetched offline", f64, steps,
         lambda: (G.circuit_csr(dtype=f64, device=dev), G.SEED_CIRCUIT + 9)),
        ("C3 webbase-1M-sized stand-in", "c3_web", f"BASELINE config 3: R-MAT scale {G.C3_WEBBASE_SCALE}, {G.C3_WEBBASE_EDGES} edges (webbase-1M's count, "
         "ufl_matrices.txt:2379), fp64; STAND-IN: the SuiteSparse file cannot be fetched offline", f64, max(steps, 200),
         lambda: (G.rmat_csr(G.C3_WEBBASE_SCALE, G.C3_WEBBASE_EDGES, dtype=f64, device=dev, seed=G.SEED_C3), G.SEED_C3 + 2)),
        ("C3 com-Orkut-sized stand-in", "c3_orkut", f"BASELINE config 3: R-MAT scale {G.C3_ORKUT_SCALE}, {G.C3_ORKUT_EDGES} stored entries mirrored as a symmetric "
         "matrix (com-Orkut's count), fp64; STAND-IN: the SuiteSparse file cannot be fetched offline", f64, steps,
         lambda: (G.rmat_symmetric_csr(G.C3_ORKUT_SCALE, G.C3_ORKUT_EDGES, dtype=f64, device=dev, seed=G.SEED_C3), G.SEED_C3 + 2)),
        ("C4 fp32", "c4", "BASELINE config 4: 16 777 216 rows, one row of 67 108 864 nonzeros, one nonzero in every 4096-th other row, "
         "the rest empty; uniform values", f32, steps,
         lambda: (G.degenerate_csr(dtype=f32, device=dev, ones=False), G.SEED_C4 + 2)),
        ("dense32 fp32", "dense32", "the reference's own streaming input --dense=32 --size=100000000 (gpu_spmv.cu:645-650): 3 125 000 x 32", f32, steps,
         lambda: (G.dense_csr(C2_ROWS_PER_GPU, C2_NPR, dtype=f32, device=dev, ones=False), G.SEED_C2 + 2)),
        ("C5 at G = 1", "c5", "BASELINE config 5 on ONE GPU: fp64 R-MAT scale 26, 2 000 000 000 edges (x = 512 MB, beyond the Infinity Cache)",
         f64, 5, lambda: (G.rmat_csr(26, 2_000_000_000, dtype=f64, device=dev, seed=G.SEED_C5), G.SEED_C5 + 2)),
    ]


REAL_FILES = {"c3_web": "webbase-1M.mtx", "c3_orkut": "com-Orkut.mtx", "circuit": "circuit5M.mtx"}      # SuiteSparse names (ufl_matrices.txt)
STANDIN_MARK = "STAND-IN written by tools/make_standin_mtx.py"


def load_mtx(torch, G, path, tdt, dev):
    """A Matrix Market file through the PRODUCT's ingest (libmspmv_host.so: CooMatrix::InitMarket + CsrMatrix::Init, what
    gpu_spmv --mtx runs; sparse_matrix.h:217-380) -> DeviceCsr.  Returns (A, is_stand_in): a file written by
    tools/make_standin_mtx.py says so in its first comment line."""
    import numpy as np
    H = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
    H.mspmv_host_matrix_create.restype = ctypes.c_void_p
    H.mspmv_host_matrix_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
    H.mspmv_host_matrix_error.restype = ctypes.c_char_p
    H.mspmv_host_matrix_error.argtypes = [ctypes.c_void_p]
    H.mspmv_host_matrix_shape.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 3
    H.mspmv_host_matrix_copy.argtypes = [ctypes.c_void_p] * 4
    H.mspmv_host_matrix_destroy.argtypes = [ctypes.c_void_p]
    stand_in = False
    with open(path, "rb") as f:
        f.readline()
        stand_in = STANDIN_MARK.encode() in f.readline()
    st = ctypes.c_int(0)
    f32 = tdt == torch.float32
    h = H.mspmv_host_matrix_create(b"mtx", 0, 0, path.encode(), 1 if f32 else 0, ctypes.byref(st))
    try:
        if st.value != 0:
            raise RuntimeError(f"{path}: {H.mspmv_host_matrix_error(h).decode(errors='replace')}")
        r, c, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        H.mspmv_host_matrix_shape(h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(n))
        off = np.empty(r.value + 1, np.int32); col = np.empty(max(n.value, 1), np.int32)[:n.value]
        val = np.empty(max(n.value, 1), np.float32 if f32 else np.float64)[:n.value]
        H.mspmv_host_matrix_copy(h, off.ctypes.data, col.ctypes.data, val.ctypes.data)
    finally:
        H.mspmv_host_matrix_destroy(h)
    A = G.DeviceCsr(r.value, c.value, torch.from_numpy(off).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev))
    return A, stand_in


def config_records(M, torch, G, dev, steps, warmup, budget_s=150.0, vendor=True, live_pmc=True, mtx_dir=None, c5_pmc=True):
    """The `configs` array: every single-GPU configuration of BASELINE.json that the headline does not cover (and the
    circuit5M-shaped matrix of the reference's published number), through the same stateless call.  Generation is on the GPU
    and not timed; a configuration that would start after `budget_s` of this function's wall time is reported as skipped
    rather than run.  Every record carries the reference's own metric (`effective_pct_of_peak`, gpu_spmv.cu:452-465), the
    tile kernel's roofline fraction, replayed counter traffic where a committed rocprofv3 pass of that workload exists, and
    the vendor library's time on the same arrays (`vendor`)."""
    out = []
    t_start = time.perf_counter()
    for name, label, desc, tdt, k, make in config_specs(torch, G, dev, steps):
        if time.perf_counter() - t_start > budget_s:
            out.append({"config": name, "skipped": f"the configs leg had used its {budget_s:.0f} s budget"})
            continue
        t0 = time.perf_counter()
        try:
            # --mtx-dir / $MSPMV_C3_DIR: the SuiteSparse file itself when it is there (webbase-1M.mtx, com-Orkut.mtx, circuit5M.mtx) --
            # or the stand-in tools/make_standin_mtx.py wrote under that name --, through the product's Matrix Market ingest
            from_file = False
            data_label = "synthetic (generated on the GPU)"
            path = os.path.join(mtx_dir, REAL_FILES[label]) if (mtx_dir and label in REAL_FILES) else None
            if path and os.path.exists(path):
                A, stand_in = load_mtx(torch, G, path, tdt, dev)
                x_seed = G.SEED_C3 + 2
                from_file = True
                data_label = (f"stand-in read from {path} through the Matrix Market ingest" if stand_in
                              else f"suitesparse: {path} (the real matrix), through the Matrix Market ingest")
            else:
                A, x_seed = make()
            x = G.uniform_pm1(x_seed, A.cols, tdt, dev)
            torch.cuda.synchronize()
            gen_s = time.perf_counter() - t0
            ms, prof, ws, y = time_stateless(M, torch, A, x, k, min(warmup, 3))
            vb = A.values.element_size()
            b_alg = algorithmic_bytes(A.rows, A.cols, A.nnz, vb)
            info = M.launch_info(A.rows, A.nnz, vb, num_cols=A.cols)          # (with the column count: the shape the stateless call runs)
            offered = M.band_passes(A.rows, A.cols, A.nnz, vb)
            # the tile kernel's duration: its hipEvent average on the launch stream (one-launch calls: the kernel IS the step -- the
            # wall clock per step of K back-to-back calls is reported beside it, not mixed into it)
            one_launch = offered <= 1
            tile_s = prof["tile_ms"] * 1e-3
            achieved = b_alg / tile_s / 1e9 if tile_s > 0 else 0.0
            rec = {"config": name, "workload": desc, "data": data_label, "dtype": "f32" if vb == 4 else "f64", "rows": A.rows, "cols": A.cols, "nnz": A.nnz,
                   "steps": k, "ms_per_step": round(ms, 5), "value": round(2.0 * A.nnz / (ms * 1e-3) / 1e9, 3), "unit": "GFLOP/s",
                   "tile": f"{info['block_threads']}x{info['items_per_thread']}", "generation_s": round(gen_s, 2),
                   "roofline": {"kernel": "tile_kernel_snap (one launch)" if one_launch else "tile_kernel_vec<.., BAND>",
                                "achieved": round(achieved, 2), **roofline_bound(M, b_alg, achieved),
                                "algorithmic_bytes_per_launch": b_alg,
                                "kernel_ms": {"search": round(prof["search_ms"], 5), "tile": round(prof["tile_ms"], 5), "fixup": round(prof["fixup_ms"], 5)},
                                "duration_used_ms": round(tile_s * 1e3, 5), "duration_source": "hipEvent average of the tile kernel on the launch stream",
                                "wall_ms_per_step": round(ms, 5),
                                "events": f"hipEvents on the launch stream, {prof['calls']} launches"}}
            rec.update(effective_record(A.rows, A.nnz, vb, ms))
            if label == "circuit":
                rec["reference_published_pct_of_peak"] = REFERENCE_PUBLISHED_PCT
                rec["reference_published_note"] = "circuit5M itself, fp64, merge-based CsrMV on a Tesla K40 (README.md:116,137-138); other hardware, the real matrix"
            if offered > 1:
                spread = int(M.debug_band_windows(ws, A.rows, A.nnz, vb).sum())
                rec["roofline"]["column_band_passes"] = {"offered_by_policy": offered, "windows_spread_of_64": spread, "passes_run": offered if spread >= 56 else 0}
            rep_tr, src = (None, None) if from_file else replayed_traffic(label, "f32" if vb == 4 else "f64")
            if not live_pmc or from_file:
                live_tr, live_src = None, ("skipped (--no-live-pmc)" if not live_pmc else "skipped: the child run regenerates the stand-in, this record is a file")
            elif label == "c5":
                # (rocprofv3 --pmc dies -- SIGSEGV inside the tool -- while a child GENERATES the 2e9 edges, counters restricted or not;
                #  so the matrix this record was timed on is parked as a raw image in RAM-backed /dev/shm and the two child passes load it)
                live_tr, live_src = None, "skipped (--no-c5-pmc)"
                if c5_pmc:
                    img = f"/dev/shm/mspmv_bench_c5_{os.getpid()}.img"
                    try:
                        import shutil
                        need = A.nnz * (vb + 4) + 4 * (A.rows + 1) + (1 << 30)
                        if shutil.disk_usage("/dev/shm").free < need:
                            live_src = "skipped: /dev/shm has no room for the CSR image"
                        else:
                            G.save_csr_image(A, x_seed, img)
                            live_tr, live_src = live_traffic(label, steps=5, timeout_s=180, extra_args=("--load", img))
                    except Exception as e:  # noqa: BLE001
                        live_tr, live_src = None, f"{type(e).__name__}: {e}"[:200]
                    finally:
                        try:
                            os.remove(img)
                        except OSError:
                            pass
            else:
                live_tr, live_src = live_traffic(label)
            tr = live_tr if live_tr is not None else rep_tr
            if tr is None:
                rec["roofline"]["traffic"] = None
                rec["roofline"]["traffic_source"] = f"not available: live counters: {live_src}; no committed passes for this label"
            if tr is not None:
                rec["roofline"]["traffic"] = tr
                rec["roofline"]["traffic_source"] = live_src if live_tr is not None else src + f" [live counters: {live_src}]"
                rec["roofline"]["traffic_is"] = "l2_fabric_bytes_infinity_cache_hits_included"
                rec["roofline"]["traffic_over_algorithmic"] = round(tr / b_alg, 3)
                rec["roofline"]["traffic_replayed_from_committed_passes"] = rep_tr
            # a correctness witness of this very record (parity proper is tests/ -m gpu against the oracle): 2^16 seeded rows + the first,
            # last and longest recomputed in fp64 with torch gathers, against the stated bound of SURVEY 8d (no oracle import)
            rec["y_finite"] = bool(torch.isfinite(y).all().item())
            chk = M.sampled_check(A, x, y)
            rec["sampled_worst_ratio"] = chk["worst_ratio"]
            rec["sampled_check"] = chk
            rec["gathers_per_s_G"] = round(A.nnz / (ms * 1e-3) / 1e9, 2)
            if label == "c5":
                rec["roofline"]["note"] = ("x (512 MB) is beyond every cache: a gather that misses moves a whole 128-byte line whatever the load's cache "
                                           "policy, and the chip delivers ~55 G random lines/s (tools/gather_granularity, profiles/r03_gather_granularity.txt); "
                                           "`gathers_per_s_G` is to be read against that, `frac` counts each x entry once")
            if label in ("c5", "c3_orkut"):
                # scale-free graphs with an x beyond the caches: the opt-in hot-column plan on the same matrix
                rec["hot_column_plan"] = M.hotcols_bench_record(A, x, y, steps=k, warmup=2, peak_gbs=HBM_PEAK_GBS)
            if vendor:
                v_iters = 3 if label == "c5" else 5 if label == "c4" else min(k, 30)
                rec["vendor"] = vendor_record(torch, A, x, y, v_iters)
                if "ms_per_step" in rec["vendor"]:
                    rec["vendor"]["vendor_time_over_ours"] = round(rec["vendor"]["ms_per_step"] / ms, 3)
            if label == "dense5":
                # BASELINE config 1 proper: the product's cpu_spmv kernel on the same matrix, on this box's host cores
                rec["cpu"] = cpu_baseline(A, x, "C1: --dense=5 fp64", budget_s=4.0, max_iters=60)
            out.append(rec)
            del A, x, ws, y
        except Exception as e:                   # e.g. out of memory on a smaller part: report, keep the headline
            out.append({"config": name, "error": f"{type(e).__name__}: {e}"[:300]})
        torch.cuda.empty_cache()
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (one rank per
    GPU, what the launcher form of the contract does) and hand its output through -- everything the ranks wrote to stdout
    first, rank 0's JSON line LAST.  Returns the exit code.  With MSPMV_BENCH_ONE_DEVICE=1 (a one-GPU box: every rank on
    cuda:0) the process group defaults to gloo -- RCCL admits one rank per device -- and the ranks still go through the C
    operator (its exchange falls back rccl -> hipIpc together on every rank)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")       # (torch.distributed.run would set 1 and say so on stderr)
    if env.get("MSPMV_BENCH_ONE_DEVICE") == "1":
        env.setdefault("MSPMV_BENCH_BACKEND", "gloo")
        env.setdefault("MSPMV_BENCH_FORCE_C_OPERATOR", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)       # stderr passes straight through
    lines = r.stdout.splitlines()
    last_json = max((i for i, l in enumerate(lines) if l.startswith("{") and l.rstrip().endswith("}")), default=None)
    for i, l in enumerate(lines):
        if i != last_json:
            print(l)
    if last_json is not None:
        rec = lines[last_json]
        try:
            d = json.loads(rec)
            d["launched_by"] = f"bench.py itself (no WORLD_SIZE in the environment): torch.distributed.run --nproc-per-node {n}"
            rec = json.dumps(d)
        except ValueError:
            pass
        print(rec, flush=True)
    return r.returncode if (r.returncode != 0 or last_json is not None) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2 on one GPU, c5 (one R-MAT matrix cut N ways) on N > 1")
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--c5-scale", type=int, default=26)
    ap.add_argument("--c5-edges", type=int, default=2_000_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plan", action="store_true", help="skip the prepared_plan sub-record (N = 1, c2)")
    ap.add_argument("--no-single-gpu-leg", action="store_true",
                    help="N > 1, c5: skip rank 0's run of the WHOLE matrix alone afterwards (about 15 s during which the other ranks idle)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "ipc"],
                    help="N > 1: how the C operator exchanges the boundary-row carries -- one RCCL all-gather per step (default), or the hipIpc "
                         "peer backend (carries written straight into the owner's mailbox, step tags instead of a collective; never timed over links)")
    ap.add_argument("--no-live-pmc", action="store_true", help="N = 1: do not measure the headline's counter traffic with two rocprofv3 --pmc child runs (replay the committed passes instead)")
    ap.add_argument("--no-vendor", action="store_true", help="N = 1: skip the rocSPARSE comparison column (`vendor` sub-records)")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the `configs` sub-records (the other single-GPU configurations)")
    ap.add_argument("--mtx-dir", default=os.environ.get("MSPMV_C3_DIR"),
                    help="N = 1: a directory that may hold webbase-1M.mtx, com-Orkut.mtx, circuit5M.mtx (the SuiteSparse files, or the stand-ins "
                         "tools/make_standin_mtx.py writes under those names): a file that is there replaces the generated stand-in of its record, read "
                         "through the product's Matrix Market ingest, and the record's `data` says which it was (default: $MSPMV_C3_DIR)")
    ap.add_argument("--no-c5-pmc", action="store_true", help="N = 1: do not collect config 5's counter traffic (its CSR image parked in /dev/shm, two child runs that load it: ~40 s)")
    ap.add_argument("--configs-budget", type=float, default=240.0, help="seconds the `configs` leg may take before it stops starting new ones")
    ap.add_argument("--dist-timeout", type=int, default=900, help="N > 1: seconds a collective may block before the job aborts")
    ap.add_argument("--tune", default=None, help="development: BLOCKxIPT[:flags] passed to mspmv_set_tuning")
    ap.add_argument("--band-passes", type=int, default=0,
                    help="A/B: mspmv_set_band_passes (0 automatic = the product default, -1 never, >= 2 always that many)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (below) -- the same job the launcher form runs
        raise SystemExit(self_launch(args.gpus))

    import torch
    import merge_spmv_amd as M
    from merge_spmv_amd import generators as G, multi_gpu as MG
    import numpy as np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher that started {world} ranks (WORLD_SIZE={world}): the two must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP merge-path kernels have no CPU fallback)")
    # MSPMV_BENCH_ONE_DEVICE=1 + MSPMV_BENCH_BACKEND=gloo: exercise the multi-rank path on a
    # single-GPU box (all ranks on cuda:0, carries exchanged through gloo by the Python twin of the C
    # operator, because RCCL admits one rank per device) -- a functional check of the sharding /
    # exchange / reporting code, not a measurement
    one_device = os.environ.get("MSPMV_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    M.load_library()
    dist = None
    backend = os.environ.get("MSPMV_BENCH_BACKEND", "nccl")
    # MSPMV_BENCH_FORCE_MG=1 with ONE rank (torch.distributed.run --nproc-per-node 1): take the N > 1 code path anyway --
    # process group, shipped RCCL id, the C operator's multi-process form with its all-gather in the timed loop, the
    # single-GPU leg -- so that everything but "more than one rank" is exercised on a one-GPU box
    mg = world > 1 or os.environ.get("MSPMV_BENCH_FORCE_MG") == "1"
    if mg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)          # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, timeout=tmo)

    workload = args.workload or ("c5" if mg else "c2")
    dtype_name = args.dtype or WORKLOADS[workload]
    if args.tune:
        shape, _, fl = args.tune.partition(":")
        b, _, i = shape.partition("x")
        M.set_tuning(4 if dtype_name == "f32" else 8, int(b or 0), int(i or 0), int(fl or "0", 0))
    tdt = torch.float32 if dtype_name == "f32" else torch.float64
    vb = 4 if dtype_name == "f32" else 8
    if args.band_passes:
        M.set_band_passes(vb, args.band_passes)
    if workload == "dense32" and mg:
        raise SystemExit("dense32 is a single-GPU workload")

    # ---- the matrix (this rank's swath of it), directly in HBM ----------------------------------------------
    A = None
    shard = None
    if workload == "c5":
        n = 1 << args.c5_scale
        rows = cols = n
        nnz_total = args.c5_edges
        x_seed = G.SEED_C5 + 2
        scaling = "strong"
        desc = (f"C5 R-MAT scale {args.c5_scale}: {n} x {n}, {nnz_total} generated edges (duplicates kept), a,b,c,d = "
                f".57,.19,.19,.05, seed 0x5EED0005, values/x uniform in [-1,1); one matrix independent of the GPU count")
        if not mg:
            A = G.rmat_csr(args.c5_scale, nnz_total, dtype=tdt, device=dev, seed=G.SEED_C5)
        else:
            shard = MG.rmat_shard(args.c5_scale, nnz_total, rank, world, tdt, device=dev, seed=G.SEED_C5, use_dist=True)
    else:
        rows = C2_ROWS_PER_GPU * world
        cols = C2_ROWS_PER_GPU if workload == "c2" else C2_NPR
        nnz_total = rows * C2_NPR
        x_seed = G.SEED_C2 + 2
        scaling = "weak"
        desc = ((f"C2 uniform CSR: {rows} x {cols}, {C2_NPR} nnz/row, {nnz_total} nnz ({C2_ROWS_PER_GPU * C2_NPR} nnz per GPU), "
                 f"uniform random sorted columns, values/x in [-1,1)") if workload == "c2" else
                f"dense {rows} x {C2_NPR} as CSR ({nnz_total} nnz): the streaming variant of C2 (--dense=32 --size=100000000)")
        if not mg:
            A = (G.uniform_csr(rows, cols, C2_NPR, dtype=tdt, device=dev) if workload == "c2"
                 else G.dense_csr(rows, C2_NPR, dtype=tdt, device=dev, ones=False))
        else:
            shard = MG.uniform_shard(rows, cols, C2_NPR, rank, world, tdt, device=dev)
    x = G.uniform_pm1(x_seed, cols, tdt, dev)

    # ---- the operator ------------------------------------------------------------------------------------------
    plan = None
    exchange = None
    if not mg:
        # the plain drop-in call: no shard wrapper, no collective
        local_rows, local_nnz = A.rows, A.nnz
        ws = M.CsrMVWorkspace(A.rows, A.nnz, tdt, device=dev)
        y = torch.empty(A.rows, dtype=tdt, device=dev)

        def op():
            M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=cols, workspace=ws)

        def op_sync():
            torch.cuda.synchronize()
    elif (backend == "nccl" and not one_device) or args.exchange == "ipc" or os.environ.get("MSPMV_BENCH_FORCE_C_OPERATOR") == "1":
        # the C multi-GPU operator, one part per process.  The exchange asked for is tried first; if any rank fails to set it up or
        # to run two trial steps with it, every rank falls back together (rccl -> ipc -> the Python twin over torch.distributed) and
        # the record says so: a scaling run on a node this code has never met should still produce its numbers.
        local_rows, local_nnz = shard.local_rows, shard.local_nnz

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        def try_exchange(kind):
            p, why = None, None
            try:
                if kind == "ipc":
                    # hipIpc peer backend: no collective library in the step (works with all ranks on one device too)
                    p = MG.MgPlan(shard.row_split, shard.nz_split, cols, tdt, [rank], [local_rank], exchange=MG.EXCHANGE_IPC)
                else:
                    # RCCL communicator over the ranks
                    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                    if rank == 0:
                        idt.copy_(torch.frombuffer(bytearray(MG.unique_id()), dtype=torch.uint8))
                    dist.broadcast(idt, 0)
                    p = MG.MgPlan(shard.row_split, shard.nz_split, cols, tdt, [rank], [local_rank], exchange=MG.EXCHANGE_RCCL,
                                  id128=bytes(idt.cpu().numpy().tobytes()))
                p.set_part(0, shard.values, shard.row_offsets, shard.column_indices)
                blob = p.ipc_export() if kind == "ipc" else None
            except Exception as e:  # noqa: BLE001 - any failure means "fall back"
                why = f"setup: {e}"
            if not all_ok(why is None):
                return p, why or "another rank failed during setup"
            try:
                if kind == "ipc":
                    blobs = [None] * world
                    dist.all_gather_object(blobs, blob)
                    p.ipc_import(blobs)
                p.x(0).copy_(x)
                torch.cuda.synchronize()
                for _ in range(2):
                    p.csrmv()
                p.synchronize(); torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                why = f"trial steps: {e}"
            if not all_ok(why is None):
                return p, why or "another rank failed in the trial steps"
            return p, None

        order = [args.exchange] + [k for k in ("rccl", "ipc") if k != args.exchange]
        fallbacks = []
        for kind in order:
            plan, why = try_exchange(kind)
            if why is None:
                break
            fallbacks.append({"exchange": kind, "failed": why[:300]})
            if plan is not None:
                try:
                    plan.close()
                except Exception:  # noqa: BLE001
                    pass
            plan = None
        if plan is not None:
            exchange = plan.info()
            if fallbacks:
                exchange["fallbacks"] = fallbacks

            def op():
                plan.csrmv()

            def op_sync():
                plan.synchronize(); torch.cuda.synchronize()
        else:
            sharded = MG.ShardedCsrMV(shard, group=None)
            exchange = {"exchange": "python twin over " + backend, "carry_bytes_per_step": world * vb, "fallbacks": fallbacks}

            def op():
                sharded(x)

            def op_sync():
                torch.cuda.synchronize()
    else:
        local_rows, local_nnz = shard.local_rows, shard.local_nnz
        sharded = MG.ShardedCsrMV(shard, group=None)
        exchange = {"exchange": "python twin over " + backend, "carry_bytes_per_step": world * vb}

        def op():
            sharded(x)

        def op_sync():
            torch.cuda.synchronize()

    def barrier():
        op_sync()
        if dist is not None:
            dist.barrier()
        op_sync()

    for _ in range(args.warmup):
        op()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        op()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- per-kernel durations: the same K steps again with hipEvents on the launch stream
    M.profile_begin(args.steps)
    for _ in range(args.steps):
        op()
    op_sync()
    prof = M.profile_end()

    # ---- N > 1 through the C operator: what the step's exchange alone takes (events around it, per rank), and the same steps with the
    # parts' hot-column plans switched OFF (they are automatic since round 5: the headline above ran with whatever the plan decided)
    hot = None
    exchange_ms = None
    if plan is not None:
        ex = []
        for _ in range(5):
            op(); op_sync()
            try:
                ex.append(plan.exchange_ms(0))
            except Exception:  # noqa: BLE001
                break
        exchange_ms = sum(ex) / len(ex) if ex else None
    if plan is not None and workload == "c5":
        hot_parts = int(plan.info().get("hot_parts", 0))
        torch.cuda.synchronize(); th0 = time.perf_counter()
        plan.hot_columns(not hot_parts)            # the other setting: off when the plan chose them, on when it did not
        op_sync(); switch_ms = (time.perf_counter() - th0) * 1e3
        for _ in range(max(2, min(args.warmup, 5))):
            op()
        barrier()
        th0 = time.perf_counter()
        for _ in range(args.steps):
            op()
        barrier()
        th = torch.tensor([time.perf_counter() - th0, switch_ms], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
        other_ms = float(th[0].item()) * 1e3 / args.steps
        hot = {"api": "mspmv_mg_plan_hot_columns: AUTOMATIC by default (decided per part when its matrix is attached: x beyond the Infinity Cache and columns "
                      "that come back, mspmv_csrmv_hotcols_skew); x permuted per step, inside the timed loop; y bit for bit the same either way",
               "chosen_by_the_plan_on_rank0": bool(hot_parts), "headline_ran_with_hot_columns": bool(hot_parts),
               "this_record_is": ("the same steps WITHOUT the hot-column plans (switched off for this leg)" if hot_parts
                                  else "the same steps WITH the hot-column plans forced on (the plan had not chosen them)"),
               "ms_per_step": round(other_ms, 5), "value": round(2.0 * nnz_total / (other_ms * 1e-3) / 1e9, 3), "unit": "GFLOP/s",
               ("release_ms" if hot_parts else "setup_ms") + "_max_over_ranks": round(float(th[1].item()), 2)}
        plan.hot_columns(-1)                       # back to the default for what follows
        op_sync()

    # ---- N > 1, c5: rank 0 runs the WHOLE matrix alone on its GPU in the same job -----------------------------------
    single = None
    if mg and workload == "c5" and not args.no_single_gpu_leg:
        y0 = None
        if plan is not None:
            if rank == 0:
                op_sync()
                y0 = plan.y(0).clone()          # rank 0's owned rows of the last step (rows 0 .. of the whole matrix)
            plan.close()
        plan = None; shard = None; sharded = None
        torch.cuda.empty_cache()
        if rank == 0:
            W = G.rmat_csr(args.c5_scale, nnz_total, dtype=tdt, device=dev, seed=G.SEED_C5)
            wws = M.CsrMVWorkspace(W.rows, W.nnz, tdt, device=dev)
            wy = torch.empty(W.rows, dtype=tdt, device=dev)
            k = 5
            for _ in range(2):
                M.csrmv(W.values, W.row_offsets, W.column_indices, x, y=wy, num_cols=cols, workspace=wws)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(k):
                M.csrmv(W.values, W.row_offsets, W.column_indices, x, y=wy, num_cols=cols, workspace=wws)
            torch.cuda.synchronize()
            sms = (time.perf_counter() - t1) * 1e3 / k
            single = {"n_gpus": 1, "steps": k, "ms_per_step": round(sms, 5), "value": round(2.0 * nnz_total / (sms * 1e-3) / 1e9, 3),
                      "unit": "GFLOP/s", "speedup_of_this_job": round(sms / ms_per_step, 3),
                      "parallel_efficiency": round(sms / ms_per_step / world, 4)}
            if y0 is not None:
                # rank 0's tiles are the single-GPU call's first tiles, so its rows must match bit for bit (its last row, completed by
                # the next ranks' carries, to within re-association)
                ref = wy[:y0.numel()]
                differ = int((ref != y0).sum().item())
                part_info = M.launch_info(local_rows, local_nnz, vb)
                whole_info = M.launch_info(W.rows, W.nnz, vb)
                same = (part_info["items_per_thread"] == whole_info["items_per_thread"]
                        and M.band_passes(W.rows, W.cols, W.nnz, vb) <= 1 and M.band_passes(local_rows, cols, local_nnz, vb) <= 1)
                neq = (ref != y0).nonzero()
                single["rank0_rows_vs_single_gpu"] = {"rows": int(y0.numel()), "not_bitwise_equal": differ,
                                                      "first_differing_row": int(neq[0].item()) if differ else None,
                                                      "tile_items": int(part_info["tile_items"]),
                                                      "max_abs_diff": float((ref - y0).abs().max().item()) if y0.numel() else 0.0,
                                                      "same_tiling": bool(same),
                                                      "note": "bit for bit when rank 0's part and the whole matrix run the same tile shape through the same "
                                                              "path (`same_tiling`: true at BASELINE's size), except inside rank 0's LAST tile, which ends where the part "
                                                              "ends (another extent, possibly the other in-tile reduction): rows from `first_differing_row` on; "
                                                              "otherwise the same sums in another association"}
            del W, wws, wy

    # anything native code left in C stdio buffers (RCCL prints a version banner at communicator creation) goes out on
    # EVERY rank before rank 0 prints the result, so that the JSON line is the last line of the job's stdout
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    per_rank = None
    if dist is not None:
        # per-rank kernel times and what is left of a step after them (launch gaps + the carry exchange + the owner's add)
        mine = torch.tensor([prof["search_ms"], prof["tile_ms"], prof["fixup_ms"], elapsed_local * 1e3 / args.steps, float(local_nnz),
                             float("nan") if exchange_ms is None else exchange_ms], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows_ = torch.stack(allr).cpu().numpy()
        kern = rows_[:, 0] + rows_[:, 1] + rows_[:, 2]
        per_rank = {"tile_ms_max": round(float(rows_[:, 1].max()), 5), "tile_ms_min": round(float(rows_[:, 1].min()), 5),
                    "step_ms_max": round(float(rows_[:, 3].max()), 5), "step_ms_min": round(float(rows_[:, 3].min()), 5),
                    "exchange_and_gaps_ms_max": round(float((rows_[:, 3] - kern).max()), 5),
                    "exchange_and_gaps_ms_min": round(float((rows_[:, 3] - kern).min()), 5),
                    "nnz_per_rank_max": int(rows_[:, 4].max()), "nnz_per_rank_min": int(rows_[:, 4].min()),
                    "exchange_ms_max": None if np.isnan(rows_[:, 5]).all() else round(float(np.nanmax(rows_[:, 5])), 5),
                    "exchange_ms_min": None if np.isnan(rows_[:, 5]).all() else round(float(np.nanmin(rows_[:, 5])), 5),
                    "note": "per rank: hipEvent averages of its kernels and its own wall time per step; exchange_and_gaps = step - kernels (inferred); "
                            "exchange_ms = hipEvents on the rank's stream right after its SpMV and right after the all-gather + the owner's add "
                            "(mspmv_mg_plan_exchange_ms, 5 separate steps): the exchange alone, the wait for slower ranks included"}
        dist.barrier()
    if rank == 0:
        gflops = 2.0 * nnz_total / (ms_per_step * 1e-3) / 1e9
        info = M.launch_info(local_rows, local_nnz, vb)
        b_alg = algorithmic_bytes(local_rows, cols, local_nnz, vb)
        tile_s = prof["tile_ms"] * 1e-3
        achieved = b_alg / tile_s / 1e9 if tile_s > 0 else 0.0
        traffic = None
        replayed = None
        traffic_source = "not measured: the counter passes (rocprofv3 --pmc) are run for single-GPU workloads only"
        if not mg:
            label = "c2_f32" if (workload, dtype_name) == ("c2", "f32") else workload
            replayed, src = replayed_traffic(label, dtype_name)
            live, live_src = ((None, "skipped (--no-live-pmc, a tuning override, or config 5: rocprofv3 does not survive generating its 36 GB)")
                              if args.no_live_pmc or args.tune or args.band_passes or workload == "c5" else live_traffic(label))
            if live is not None:
                traffic, traffic_source = live, live_src
            elif replayed is not None:
                traffic, traffic_source = replayed, src + f" [live counters: {live_src}]"
        out = {
            "metric": "CsrMV GFLOP/s", "value": round(gflops, 3), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": desc,
                       "tile": f"{info['block_threads']}x{info['items_per_thread']}",
                       "partition": ("single GPU" if not mg else
                                     f"merge-path diagonal split over {world} GPUs (mspmv_mg_partition): "
                                     f"{local_rows - 1} rows + {local_nnz} nonzeros on rank 0; one exchange of {world} carries per step")},
            "effective_GBs_reference_formula": round(effective_bytes(rows, nnz_total, vb) / (ms_per_step * 1e-3) / 1e9, 2),
            "effective_pct_of_peak": round(100.0 * effective_bytes(rows, nnz_total, vb) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 2),
            "compulsory_GBs": round(algorithmic_bytes(rows, cols, nnz_total, vb) / (ms_per_step * 1e-3) / 1e9, 2),
            "roofline": {"kernel": "tile_kernel_snap (one launch per part)", "achieved": round(achieved, 2), **roofline_bound(M, b_alg, achieved),
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_is": "l2_fabric_bytes_infinity_cache_hits_included",
                         "traffic_over_algorithmic": round(traffic / b_alg, 3) if traffic else None,
                         "traffic_replayed_from_committed_passes": (replayed if not mg else None), "algorithmic_bytes_per_launch": b_alg,
                         "kernel_ms": {"search": round(prof["search_ms"], 5), "tile": round(prof["tile_ms"], 5),
                                       "fixup": round(prof["fixup_ms"], 5)},
                         "events": f"hipEvents on the launch stream, {prof['calls']} launches"},
        }
        if not mg:
            # column-band passes of the stateless call: offered by the sizes, accepted (or not) by the device-side windows
            offered = M.band_passes(local_rows, cols, local_nnz, vb)
            rec = {"offered_by_policy": offered, "passes_run": 0, "windows_spread_of_64": None}
            if offered > 1:
                spread = int(M.debug_band_windows(ws, local_rows, local_nnz, vb).sum())
                rec["windows_spread_of_64"] = spread
                rec["passes_run"] = offered if (spread >= 56 or args.band_passes >= 2) else 0
                rec["note"] = ("the tile kernel streamed the CSR arrays `passes_run` times, each pass gathering one column band of x "
                               "(the slice stays in every XCD's L2); algorithmic bytes count the arrays once")
            out["roofline"]["column_band_passes"] = rec
            # which kernel the figures are for: candidates for the column-band passes run the classic three launches with the BAND
            # tile kernel, every other call ONE launch of tile_kernel_snap (its "search" / "fixup" figures are then just the cost of
            # two back-to-back event records)
            out["roofline"]["kernel"] = "tile_kernel_vec<.., BAND>" if offered > 1 else "tile_kernel_snap (one launch: no coordinate pass, no fix-up)"
        if not mg:
            chk = M.sampled_check(A, x, y)
            out["sampled_worst_ratio"] = chk["worst_ratio"]
            out["sampled_check"] = dict(chk, what="2^16 seeded rows + the first, last and longest recomputed in fp64 with torch gathers against the stated bound "
                                                  "|y - g| <= 2 (ceil(log2(len + 1)) + depth + 8) eps s (SURVEY 8d); < 1 passes; parity proper: tests/ -m gpu")
        if exchange is not None:
            out["exchange"] = {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in exchange.items() if k != "steps"}
            if isinstance(exchange.get("exchange"), int):
                out["exchange"]["backend"] = {1: "RCCL ncclAllGather (1 element per rank) below the C ABI", 2: "peer reads",
                                              3: "hipIpc peer writes into the owner's mailbox, step-tagged (no collective)"}.get(exchange["exchange"])
        if per_rank is not None:
            out["per_rank"] = per_rank
        if hot is not None:
            out["hot_column_plan"] = hot
        if single is not None:
            out["single_gpu_same_workload"] = single
        if not mg and workload == "c2" and not args.no_plan and hasattr(M, "CsrMVPlan"):
            out["prepared_plan"] = M.plan_bench_record(A, x, y, steps=args.steps, warmup=args.warmup, peak_gbs=HBM_PEAK_GBS)
        if not mg and not args.no_vendor:
            out["vendor"] = vendor_record(torch, A, x, y, min(args.steps, 30))
            if "ms_per_step" in out["vendor"]:
                out["vendor"]["vendor_time_over_ours"] = round(out["vendor"]["ms_per_step"] / ms_per_step, 3)
        if not mg and not args.no_cpu_baseline and A.nnz <= 400_000_000:
            out["cpu_baseline"] = cpu_baseline(A, x, "same " + workload.upper() + " matrix")
        if not mg and workload == "c2" and dtype_name == "f32" and not args.no_configs and not args.tune and not args.band_passes:
            del A, ws, y
            torch.cuda.empty_cache()
            out["configs"] = config_records(M, torch, G, dev, min(args.steps, 50), args.warmup, args.configs_budget, vendor=not args.no_vendor, live_pmc=not args.no_live_pmc,
                                            mtx_dir=args.mtx_dir, c5_pmc=not args.no_c5_pmc)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
