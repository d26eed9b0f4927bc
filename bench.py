#!/usr/bin/env python3
"""bench.py -- CsrMV throughput of the MI355X-native merge-based SpMV.

    python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or plainly -- self_launch)

A "step" is one y = A*x through the C ABI (include/mspmv.h); inputs are resident in HBM before timing.

N = 1: BASELINE.json config 2 -- fp32, 3 125 000^2, 32 nnz/row = 100 000 000 nnz, uniform random sorted columns --
       through the stateless drop-in call mspmv_csrmv_f32.
N > 1: the same matrix family, WEAK scaling -- N x 3 125 000 rows over the same 3 125 000 columns, cut by merge-path diagonals into N
       swaths (mspmv_mg_partition: every GPU holds what the N = 1 line's GPU holds), one rank per GPU, ONE RCCL all-gather of the N
       boundary-row carries per step below the C ABI: the per-N values of a scaling series are about one workload.
       Then, as a second leg in the same job (`c5_strong`; --no-c5-leg skips it), BASELINE.json config 5 -- fp64 R-MAT scale 26,
       2 000 000 000 edges, ONE matrix whatever N, strong scaling.  `--workload c5` makes config 5 the headline instead; rank 0 then also runs that whole
       matrix alone (`single_gpu_same_workload`; --no-single-gpu-leg skips it; --c5-single-gpu-leg adds it to the default run's leg).
       `--preflight`: communicator, one all-gather, one step, exit (< 10 s; a failure names the rank).

The LAST stdout line is ONE JSON object under 4 KB, numbers only (what the reference prints is one perf line, gpu_spmv.cu:459-471):
  value        2 * nnz_total / t  (GFLOP/s, whole job; gpu_spmv.cu:451-465)
  roofline     algorithmic bytes of one tile-kernel launch (every array once) / its average duration from hipEvents on
               the launch stream, read against the 8 TB/s HBM3E spec peak (`frac`); arrays that fit the 256 MB Infinity Cache carry `frac_ic`
               as context (against a bare read stream over a buffer of that size measured in this run); `traffic` = L2<->fabric bytes per launch
               (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction; Infinity-Cache hits included) -- replayed from the
               committed rocprofv3 --pmc passes of the same workload (`traffic_src` names the file), measured live with --full
  cpu_baseline the product's OpenMP merge-path kernel (host/merge_csrmv.hpp, what cpu_spmv runs; bit for bit the oracle's,
               tests/test_cpu_product_parity.py) on the same matrix on this box's host cores: bounded sample
  configs      one compact object per remaining single-GPU configuration of BASELINE.json (+ the circuit5M-shaped matrix of the
               reference's one published number); `worst` = sampled_check's worst |y - g| / bound (< 1 passes)
Everything longer -- notes, sources, per-run CPU variants, vendor column, plans -- goes to the side file named in `detail`
(default gpurun_out/bench_detail.json) and, with --full, includes live counters, rocSPARSE, the prepared plans and config 5 at G = 1
(tools/bench_full.py)."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20       # the die-level L3 (same guide)
LINE_LIMIT = 4096                      # the final stdout line stays under this many bytes (tests/test_bench_helpers.py)
WORKLOADS = {"c2": "f32", "dense32": "f32", "c5": "f64"}
C2_ROWS_PER_GPU, C2_NPR = 3_125_000, 32
REFERENCE_PUBLISHED_PCT = 62.96        # circuit5M fp64 on a K40: 181.6 effective GB/s of 288.4 (README.md:116,137-138)
REAL_FILES = {"c3_web": "webbase-1M.mtx", "c3_orkut": "com-Orkut.mtx", "circuit": "circuit5M.mtx"}      # SuiteSparse names (ufl_matrices.txt)
STANDIN_MARK = "STAND-IN written by tools/make_standin_mtx.py"


def algorithmic_bytes(rows, cols, nnz, vb):
    """every CSR array, x and y touched exactly once."""
    return nnz * (vb + 4) + (rows + 1) * 4 + rows * vb + cols * vb


def effective_bytes(rows, nnz, vb):
    """the reference's byte model, gpu_spmv.cu:452-456"""
    return nnz * (2 * vb + 4) + rows * (4 + vb)


_CACHE_RATE = {}


def roofline_bound(M, b_alg, achieved_gbs):
    """What the achieved rate is read against: ALWAYS the HBM3E spec peak (`peak`, `frac`).  Arrays that fit the 256 MB Infinity
    Cache stay there from SpMV to SpMV; for those the rate of a bare 16-byte-per-lane read stream over a buffer of the same size,
    measured in this run (mspmv_probe_read_stream), is kept beside it as context (`ic_resident`, `ic_stream_peak`, `frac_ic`)."""
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4)}
    if b_alg > INFINITY_CACHE_BYTES:
        return out
    key = int(b_alg) >> 20
    if key not in _CACHE_RATE:
        try:
            _CACHE_RATE[key] = M.cache_stream_rate(b_alg)
        except Exception as e:  # noqa: BLE001
            _CACHE_RATE[key] = None
            sys.stderr.write(f"cache stream probe failed: {e}\n")
    peak = _CACHE_RATE[key]
    out.update({"ic_resident": True, "ic_stream_peak": round(peak, 1) if peak else None, "frac_ic": round(achieved_gbs / peak, 4) if peak else None})
    return out


def _cgroup_cpu_stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, _, v = line.partition(" ")
            out[k] = int(v)
    except (OSError, ValueError):
        pass
    return out.get("nr_throttled"), out.get("throttled_usec")


def cpu_baseline(A, x, label, budget_s=12.0, max_iters=200):
    """The product's OpenMP merge-path kernel on the host cores (rank 0, N = 1 only): quota-sized team bound to socket 0, the same
    unbound, and two threads under the quota unbound -- each with the cgroup's throttling counters read around the timed loop.
    `value` is the fastest; all three go to the detail file."""
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
        runs.append({"variant": name, "threads": threads, "bound_to_socket0_cores": bool(pinned.value), "sockets_visible": packages.value,
                     "ms": round(avg.value, 3), "iters": iters.value, "value": round(2.0 * nnz / (avg.value * 1e-3) / 1e9, 3),
                     "cgroup_throttled_periods": None if thr0[0] is None else thr1[0] - thr0[0]})
    good = [r for r in runs if "value" in r]
    if not good:
        return {"error": "; ".join(r.get("error", "?") for r in runs)[:200]}
    best = max(good, key=lambda r: r["value"])
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = "unlimited" if q == "max" else f"{int(q) / int(per):g} CPUs"
    except (OSError, ValueError):
        quota = "?"
    return {"value": best["value"], "unit": "GFLOP/s", "cores": best["threads"], "kind": "port", "ms": best["ms"],
            "sample": f"{label}: {best['iters']} SpMVs after 4 warm-ups", "kernel": "host/merge_csrmv.hpp (OpenMP merge-path)",
            "variant": best["variant"], "runs": runs, "hardware_threads": int(H.mspmv_host_hardware_threads()), "cpu_quota": quota}


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
    """L2 <-> fabric bytes per launch of the tile kernel from the committed rocprofv3 --pmc passes of the same workload
    (profiles/*/pmc_latest.json, written by tools/gpu_profile.sh): (bytes, repo-relative path).  The newest round's file wins."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_latest.json"))):
        try:
            pmc = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if pmc.get("workload") == workload and pmc.get("dtype") == dtype_name and pmc.get("tile_kernel_hbm_bytes_per_launch"):
            best = (int(pmc["tile_kernel_hbm_bytes_per_launch"]), os.path.relpath(path, ROOT))
    return best if best else (None, None)


def config_specs(torch, G, dev, steps):
    """(name, pmc label, description, dtype, steps, make) of every single-GPU configuration besides the headline -- shared with
    tools/run_config.py, which profiles one of them under rocprofv3."""
    f32, f64 = torch.float32, torch.float64
    return [
        ("C1 dense5", "dense5", "BASELINE config 1's matrix: --dense=5 (cpu_spmv.cpp:581-587), fp64, on the GPU; `cpu` = the product's cpu_spmv kernel", f64, max(steps, 100),
         lambda: (G.dense_csr((1 << 24) // 5, 5, dtype=f64, device=dev, ones=True), G.SEED_C2 + 2)),
        ("C2 f64", "c2", "BASELINE config 2's matrix in the reference's default precision (gpu_spmv.cu:727-735)", f64, steps,
         lambda: (G.uniform_csr(C2_ROWS_PER_GPU, C2_ROWS_PER_GPU, C2_NPR, dtype=f64, device=dev), G.SEED_C2 + 2)),
        ("circuit5M-shaped", "circuit", f"stand-in of circuit5M's sizes ({G.CIRCUIT5M_ROWS}^2, {G.CIRCUIT5M_NNZ} nnz; README.md:116,137-138), fp64", f64, steps,
         lambda: (G.circuit_csr(dtype=f64, device=dev), G.SEED_CIRCUIT + 9)),
        ("C3 webbase-sized", "c3_web", f"BASELINE config 3 stand-in: R-MAT scale {G.C3_WEBBASE_SCALE}, {G.C3_WEBBASE_EDGES} edges (ufl_matrices.txt:2379), fp64", f64, max(steps, 200),
         lambda: (G.rmat_csr(G.C3_WEBBASE_SCALE, G.C3_WEBBASE_EDGES, dtype=f64, device=dev, seed=G.SEED_C3), G.SEED_C3 + 2)),
        ("C3 Orkut-sized", "c3_orkut", f"BASELINE config 3 stand-in: R-MAT scale {G.C3_ORKUT_SCALE}, {G.C3_ORKUT_EDGES} entries mirrored, fp64", f64, steps,
         lambda: (G.rmat_symmetric_csr(G.C3_ORKUT_SCALE, G.C3_ORKUT_EDGES, dtype=f64, device=dev, seed=G.SEED_C3), G.SEED_C3 + 2)),
        ("C4", "c4", "BASELINE config 4: 16 777 216 rows, one row of 67 108 864 nonzeros, one nonzero in every 4096-th other row, rest empty, fp32", f32, steps,
         lambda: (G.degenerate_csr(dtype=f32, device=dev, ones=False), G.SEED_C4 + 2)),
        ("dense32", "dense32", "the reference's streaming input --dense=32 --size=100000000 (gpu_spmv.cu:645-650), fp32", f32, steps,
         lambda: (G.dense_csr(C2_ROWS_PER_GPU, C2_NPR, dtype=f32, device=dev, ones=False), G.SEED_C2 + 2)),
        ("C5 G=1", "c5", "BASELINE config 5 on ONE GPU: fp64 R-MAT scale 26, 2 000 000 000 edges (--full only)", f64, 5,
         lambda: (G.rmat_csr(26, 2_000_000_000, dtype=f64, device=dev, seed=G.SEED_C5), G.SEED_C5 + 2)),
    ]


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


def roofline_record(M, A, cols, prof, ms, ws, workload_label, dtype_name):
    """The `roofline` object of one stateless record (full form; compact() keeps the numbers)."""
    vb = A.values.element_size()
    b_alg = algorithmic_bytes(A.rows, cols, A.nnz, vb)
    tile_s = prof["tile_ms"] * 1e-3
    achieved = b_alg / tile_s / 1e9 if tile_s > 0 else 0.0
    offered = M.band_passes(A.rows, cols, A.nnz, vb)
    passes = clocked = 0
    if offered > 1:
        # a candidate for the banded form: the device-side windows decide; since round 6 the form is the one-pass clock-scheduled bands
        # (csrc/mspmv_tdm.hpp) -- `band_passes` stays in the record as the count the passes WOULD have taken (0: the windows refused)
        spread = int(M.debug_band_windows(ws, A.rows, A.nnz, vb).sum())
        passes = offered if spread >= 56 else 0
        clocked = M.clocked_bands(A.rows, cols, A.nnz, vb)[0] if passes else 0
    kernel = "tile_kernel_vec<BAND,TDM>" if clocked else "tile_kernel_vec<BAND>" if offered > 1 else "tile_kernel_snap"
    rec = {"kernel": kernel, "achieved": round(achieved, 1), **roofline_bound(M, b_alg, achieved),
           "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(prof["tile_ms"], 5), "search_ms": round(prof["search_ms"], 5),
           "fixup_ms": round(prof["fixup_ms"], 5), "launches_timed": prof["calls"], "band_passes": passes, "clocked_bands": clocked}
    tr, src = replayed_traffic(workload_label, dtype_name)
    rec["traffic"] = tr
    rec["traffic_over_algorithmic"] = round(tr / b_alg, 3) if tr else None
    rec["traffic_src"] = ("replayed:" + src) if tr else None
    return rec


def config_records(M, torch, G, dev, steps, warmup, budget_s, mtx_dir=None, full=None):
    """The `configs` array (full form): every single-GPU configuration of BASELINE.json that the headline does not cover, through
    the same stateless call.  Generation is on the GPU and not timed; a configuration that would start after `budget_s` is
    reported as skipped.  `full` (tools/bench_full.py, --full) adds live counters, the vendor column, the plans and config 5."""
    out = []
    t_start = time.perf_counter()
    for name, label, desc, tdt, k, make in config_specs(torch, G, dev, steps):
        if label == "c5" and full is None:
            continue
        if time.perf_counter() - t_start > budget_s:
            out.append({"config": name, "skipped": f"configs budget of {budget_s:.0f} s used"})
            continue
        t0 = time.perf_counter()
        try:
            data_label = "synthetic"
            path = os.path.join(mtx_dir, REAL_FILES[label]) if (mtx_dir and label in REAL_FILES) else None
            from_file = bool(path and os.path.exists(path))
            if from_file:
                A, stand_in = load_mtx(torch, G, path, tdt, dev)
                x_seed = G.SEED_C3 + 2
                data_label = ("stand-in file " if stand_in else "suitesparse ") + path
            else:
                A, x_seed = make()
            x = G.uniform_pm1(x_seed, A.cols, tdt, dev)
            torch.cuda.synchronize()
            gen_s = time.perf_counter() - t0
            ms, prof, ws, y = time_stateless(M, torch, A, x, k, min(warmup, 3))
            vb = A.values.element_size()
            dn = "f32" if vb == 4 else "f64"
            info = M.launch_info(A.rows, A.nnz, vb, num_cols=A.cols)
            eff = effective_bytes(A.rows, A.nnz, vb) / (ms * 1e-3) / 1e9
            rec = {"config": name, "label": label, "workload": desc, "data": data_label, "dtype": dn, "rows": A.rows, "cols": A.cols, "nnz": A.nnz,
                   "steps": k, "ms_per_step": round(ms, 5), "value": round(2.0 * A.nnz / (ms * 1e-3) / 1e9, 1), "unit": "GFLOP/s",
                   "tile": f"{info['block_threads']}x{info['items_per_thread']}", "generation_s": round(gen_s, 2),
                   "roofline": roofline_record(M, A, A.cols, prof, ms, ws, label, dn) if not from_file else
                               dict(roofline_record(M, A, A.cols, prof, ms, ws, "-", dn)),
                   "effective_GBs": round(eff, 1), "effective_pct_of_peak": round(100.0 * eff / HBM_PEAK_GBS, 2)}
            if label == "circuit":
                rec["reference_published_pct_of_peak"] = REFERENCE_PUBLISHED_PCT
            chk = M.sampled_check(A, x, y)
            rec["sampled_check"] = chk
            rec["y_finite"] = bool(torch.isfinite(y).all().item())
            if full is not None:
                full.extend_config(rec, M, torch, G, A, x, y, ws, label, x_seed, from_file, k)
            if label == "dense5":
                rec["cpu"] = cpu_baseline(A, x, "C1 --dense=5 fp64", budget_s=3.0, max_iters=60)
            out.append(rec)
            del A, x, ws, y
        except Exception as e:                   # e.g. out of memory on a smaller part: report, keep the headline
            out.append({"config": name, "error": f"{type(e).__name__}: {e}"[:200]})
        torch.cuda.empty_cache()
    return out


# ---- the final line ------------------------------------------------------------------------------------------------------------

_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "ic_resident", "frac_ic", "traffic", "traffic_over_algorithmic", "traffic_src",
              "algorithmic_bytes_per_launch", "kernel_ms", "band_passes", "clocked_bands")


def _compact_config(c):
    if "error" in c or "skipped" in c:
        return {"config": c.get("config"), "error": str(c.get("error", c.get("skipped")))[:60]}
    r = c.get("roofline", {})
    out = {"config": c["config"], "dtype": c["dtype"], "ms_per_step": c["ms_per_step"], "value": c["value"], "bound": r.get("bound"),
           "frac": r.get("frac"), "traffic_over_algorithmic": r.get("traffic_over_algorithmic"), "worst": c.get("sampled_check", {}).get("worst_ratio")}
    if r.get("ic_resident"):
        out["frac_ic"] = r.get("frac_ic")
    if "cpu" in c and "value" in c["cpu"]:
        out["cpu_gflops"] = c["cpu"]["value"]; out["cpu_cores"] = c["cpu"]["cores"]
    if "data" in c and c["data"] != "synthetic":
        out["data"] = c["data"][:60]
    return out


def compact_line(detail, detail_path=None):
    """The driver's line: the contract's keys, `roofline`, `cpu_baseline`, one compact object per config -- numbers, short tags,
    nothing else; strictly JSON and under LINE_LIMIT bytes whatever `detail` holds (optional parts are dropped until it is)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: detail.get(k) for k in keep}
    cfg = detail.get("config", {})
    line["config"] = {"workload": str(cfg.get("workload", ""))[:160], "tile": cfg.get("tile"), "partition": str(cfg.get("partition", ""))[:120]}
    r = detail.get("roofline", {})
    line["roofline"] = {k: r[k] for k in _ROOF_KEYS if k in r}
    cb = detail.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = ({"error": str(cb["error"])[:80]} if "error" in cb else
                                {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "ms", "sample", "kernel")})
    for k in ("effective_pct_of_peak", "sampled_worst_ratio"):
        if k in detail:
            line[k] = detail[k]
    optional = []
    if "exchange" in detail:
        ex = detail["exchange"]
        line["exchange"] = {"backend": str(ex.get("backend", ex.get("exchange")))[:40], "carry_bytes_per_step": ex.get("carry_bytes_per_step"),
                            "hot_parts": ex.get("hot_parts"), "fallbacks": len(ex.get("fallbacks", []))}
    if "per_rank" in detail:
        line["per_rank"] = {k: v for k, v in detail["per_rank"].items() if k != "note"}
    if "single_gpu_same_workload" in detail:
        s = detail["single_gpu_same_workload"]
        line["single_gpu_same_workload"] = {k: s.get(k) for k in ("n_gpus", "steps", "ms_per_step", "value") if k in s}
        optional.append("single_gpu_same_workload")
    if "preflight" in detail:
        line["preflight"] = detail["preflight"]
    if detail.get("c5_strong"):
        c = detail["c5_strong"]
        if "error" in c:
            line["c5_strong"] = {"error": str(c["error"])[:120]}
        else:
            line["c5_strong"] = {"n_gpus": c.get("n_gpus"), "scaling": c.get("scaling"), "dtype": c.get("dtype"), "steps": c.get("steps"),
                                 "ms_per_step": c.get("ms_per_step"), "value": c.get("value"), "frac": c.get("roofline", {}).get("frac"),
                                 "backend": str(c.get("exchange", {}).get("backend", c.get("exchange", {}).get("exchange")))[:40],
                                 "hot_parts": c.get("exchange", {}).get("hot_parts"),
                                 "tile_ms_max": c.get("per_rank", {}).get("tile_ms_max"), "exchange_ms_max": c.get("per_rank", {}).get("exchange_ms_max"),
                                 "single_gpu_value": c.get("single_gpu_same_workload", {}).get("value")}
    if "prepared_plan" in detail and "ms_per_step" in detail["prepared_plan"]:
        p = detail["prepared_plan"]
        line["prepared_plan"] = {"ms_per_step": p["ms_per_step"], "value": p["value"], "frac": p["roofline"]["frac"], "setup_ms": p["setup_ms"]}
        optional.append("prepared_plan")
    if "vendor" in detail and "ms_per_step" in detail["vendor"]:
        line["vendor"] = {"library": "rocSPARSE csrmv", "ms_per_step": detail["vendor"]["ms_per_step"], "analysis_ms": detail["vendor"]["analysis_ms"]}
        optional.append("vendor")
    if detail.get("configs"):
        line["configs"] = [_compact_config(c) for c in detail["configs"]]
        optional.append("configs")
    if detail_path:
        line["detail"] = detail_path
    s = json.dumps(line, separators=(",", ":"), allow_nan=False)
    while len(s) >= LINE_LIMIT and optional:
        line.pop(optional.pop(0), None)
        s = json.dumps(line, separators=(",", ":"), allow_nan=False)
    if len(s) >= LINE_LIMIT:
        line["config"] = {"workload": line["config"]["workload"][:60]}
        s = json.dumps(line, separators=(",", ":"), allow_nan=False)
    return s


def _finite(o):
    """NaN / inf -> None, recursively (the line is strict JSON)."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def emit(detail, detail_path):
    """Side file first (best effort), then the line -- the LAST thing on stdout."""
    detail = _finite(detail)
    written = None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(detail, f, indent=1)
            written = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
        except OSError as e:
            sys.stderr.write(f"bench detail file not written: {e}\n")
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stderr.flush()
    print(compact_line(detail, written), flush=True)


# ---- N > 1 ---------------------------------------------------------------------------------------------------------------------

def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run (one rank per GPU) and
    hand its output through, rank 0's JSON line LAST.  MSPMV_BENCH_ONE_DEVICE=1 (a one-GPU box: every rank on cuda:0) defaults the
    process group to gloo -- RCCL admits one rank per device -- and the ranks still go through the C operator (rccl -> hipIpc)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
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
        print(lines[last_json], flush=True)
    return r.returncode if (r.returncode != 0 or last_json is not None) else 1


def preflight(torch, dist, MG, rank, world, local_rank, dev, backend, one_device):
    """--preflight: what a scaling run needs, checked in seconds and with the failing rank named -- process group, a torch
    all-reduce, the C operator's communicator (RCCL id shipped through the group), one all-gather of carries inside one step of a
    small sharded SpMV whose y is compared with the single-part result on every rank."""
    import numpy as np
    import merge_spmv_amd as M
    t0 = time.perf_counter()
    stage, why = "start", None
    try:
        stage = "torch all_reduce"
        t = torch.ones(1, device=dev if backend == "nccl" else "cpu"); dist.all_reduce(t)
        assert int(t.item()) == world, f"all_reduce gave {t.item()}, expected {world}"
        stage = "shard"
        scale, edges = 16, 1 << 20
        shard = MG.rmat_shard(scale, edges, rank, world, torch.float64, device=dev, seed=0x5EED0005, use_dist=True)
        x = torch.linspace(-1, 1, 1 << scale, dtype=torch.float64, device=dev)
        stage = "communicator"
        kind = MG.EXCHANGE_RCCL if (backend == "nccl" and not one_device) else MG.EXCHANGE_IPC
        if kind == MG.EXCHANGE_RCCL:
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(MG.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            plan = MG.MgPlan(shard.row_split, shard.nz_split, 1 << scale, torch.float64, [rank], [local_rank], exchange=kind, id128=bytes(idt.cpu().numpy().tobytes()))
        else:
            plan = MG.MgPlan(shard.row_split, shard.nz_split, 1 << scale, torch.float64, [rank], [local_rank], exchange=kind)
        plan.set_part(0, shard.values, shard.row_offsets, shard.column_indices)
        if kind == MG.EXCHANGE_IPC:
            blobs = [None] * world
            dist.all_gather_object(blobs, plan.ipc_export())
            plan.ipc_import(blobs)
        stage = "one step (SpMV + all-gather of carries)"
        plan.x(0).copy_(x)
        plan.csrmv(); plan.synchronize(); torch.cuda.synchronize()
        stage = "result check"
        yo = plan.y(0).clone()
        # the same rows by the single-GPU call on this rank's swath + the carries it is owed are what plan.y holds; check against a
        # gather-based fp64 recomputation of the owned rows
        off = shard.row_offsets.to(torch.int64); n_owned = yo.numel()
        seg = torch.repeat_interleave(torch.arange(off.numel() - 1, device=dev), off[1:] - off[:-1])
        p = shard.values * x[shard.column_indices.to(torch.int64)]
        g = torch.zeros(off.numel() - 1, dtype=torch.float64, device=dev).index_add_(0, seg, p)
        # interior rows (not the first: a row cut by the left boundary receives carries) must match to rounding
        if n_owned > 2:
            err = float((yo[1:n_owned - 1] - g[1:n_owned - 1]).abs().max().item())
            assert err < 1e-9, f"owned rows differ from the recomputation by {err}"
        plan.close()
    except Exception as e:  # noqa: BLE001
        why = f"rank {rank} failed at '{stage}': {type(e).__name__}: {e}"[:300]
        sys.stderr.write(why + "\n")
    flags = [None] * world
    try:
        dist.all_gather_object(flags, why)
    except Exception as e:  # noqa: BLE001
        flags = [why or f"rank {rank}: all_gather_object failed: {e}"]
    bad = [f for f in flags if f]
    np.seterr(all="ignore")
    return {"ok": not bad, "seconds": round(time.perf_counter() - t0, 2), "ranks": world, "backend": backend,
            "exchange": "rccl" if (backend == "nccl" and not one_device) else "ipc", "failed": bad[:4]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: c2 (N > 1: weak scaling, the same per-GPU matrix, then config 5 as a second leg); c5 = one R-MAT matrix cut N ways, strong scaling")
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--c5-scale", type=int, default=26)
    ap.add_argument("--c5-edges", type=int, default=2_000_000_000)
    ap.add_argument("--full", action="store_true", help="N = 1: add live rocprofv3 --pmc traffic, the rocSPARSE column, the prepared plans and config 5 at G = 1 (minutes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the `configs` sub-records")
    ap.add_argument("--configs-budget", type=float, default=60.0, help="seconds the `configs` leg may take before it stops starting new ones")
    ap.add_argument("--mtx-dir", default=os.environ.get("MSPMV_C3_DIR"),
                    help="N = 1: a directory that may hold webbase-1M.mtx, com-Orkut.mtx, circuit5M.mtx (default: $MSPMV_C3_DIR); a file found replaces the generated stand-in")
    ap.add_argument("--detail", default=os.environ.get("MSPMV_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json")),
                    help="side file for everything that is not in the line ('' = none)")
    ap.add_argument("--no-c5-leg", action="store_true", help="N > 1, default workload: skip the config-5 leg (one R-MAT matrix cut N ways, strong scaling) after the headline")
    ap.add_argument("--c5-single-gpu-leg", action="store_true", help="N > 1, default workload: after the config-5 leg rank 0 also runs the WHOLE R-MAT matrix alone (another ~80 s)")
    ap.add_argument("--no-single-gpu-leg", action="store_true", help="N > 1, c5: skip rank 0's run of the WHOLE matrix alone afterwards")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "ipc"], help="N > 1: RCCL all-gather per step (default) or the hipIpc peer backend")
    ap.add_argument("--preflight", action="store_true", help="N > 1: communicator init, one all-gather, one step; exits in seconds naming a failing rank")
    ap.add_argument("--dist-timeout", type=int, default=900, help="N > 1: seconds a collective may block before the job aborts")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
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
    # MSPMV_BENCH_ONE_DEVICE=1: the multi-rank path on a single-GPU box (all ranks on cuda:0): a functional check, not a measurement
    one_device = os.environ.get("MSPMV_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    M.load_library()
    dist = None
    backend = os.environ.get("MSPMV_BENCH_BACKEND", "nccl")
    # MSPMV_BENCH_FORCE_MG=1 with ONE rank: take the N > 1 code path anyway (process group, shipped RCCL id, the C operator's
    # multi-process form with its all-gather in the timed loop)
    mg = world > 1 or os.environ.get("MSPMV_BENCH_FORCE_MG") == "1"
    if mg:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)          # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, timeout=tmo)
    detail_path = args.detail or None

    if args.preflight:
        if not mg:
            raise SystemExit("--preflight is for --gpus N > 1")
        res = preflight(torch, dist, MG, rank, world, local_rank, dev, backend, one_device)
        if rank == 0:
            emit({"metric": "preflight", "value": 1.0 if res["ok"] else 0.0, "unit": "ok", "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": None,
                  "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                  "config": {"workload": "preflight: R-MAT scale 16, one step", "tile": None, "partition": f"{world} ranks"}, "preflight": res}, None)
        dist.barrier(); dist.destroy_process_group()
        raise SystemExit(0 if res["ok"] else 3)

    def measure(workload, steps, warmup, single_leg=True):
        """One workload through the whole protocol (every rank calls it); rank 0 gets the record, the others None."""
        dtype_name = args.dtype or WORKLOADS[workload]
        tdt = torch.float32 if dtype_name == "f32" else torch.float64
        vb = 4 if dtype_name == "f32" else 8
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
            desc = f"C5 R-MAT scale {args.c5_scale}: {n}^2, {nnz_total} edges (duplicates kept), a,b,c,d=.57,.19,.19,.05, seed 0x5EED0005; one matrix whatever N"
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
            desc = (f"C2 uniform CSR: {rows} x {cols}, {C2_NPR} nnz/row, {nnz_total} nnz, uniform random sorted columns, values/x in [-1,1)" if workload == "c2" else
                    f"dense {rows} x {C2_NPR} as CSR ({nnz_total} nnz): C2's streaming variant (--dense=32 --size=100000000)")
            if not mg:
                A = (G.uniform_csr(rows, cols, C2_NPR, dtype=tdt, device=dev) if workload == "c2"
                     else G.dense_csr(rows, C2_NPR, dtype=tdt, device=dev, ones=False))
            else:
                shard = MG.uniform_shard(rows, cols, C2_NPR, rank, world, tdt, device=dev)
        x = G.uniform_pm1(x_seed, cols, tdt, dev)

        # ---- the operator ------------------------------------------------------------------------------------------
        plan = None
        sharded = None
        exchange = None
        ws = y = None
        if not mg:
            local_rows, local_nnz = A.rows, A.nnz
            ws = M.CsrMVWorkspace(A.rows, A.nnz, tdt, device=dev)
            y = torch.empty(A.rows, dtype=tdt, device=dev)

            def op():
                M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=cols, workspace=ws)

            def op_sync():
                torch.cuda.synchronize()
        elif (backend == "nccl" and not one_device) or args.exchange == "ipc" or os.environ.get("MSPMV_BENCH_FORCE_C_OPERATOR") == "1":
            # the C multi-GPU operator, one part per process.  The exchange asked for is tried first; if any rank fails to set it up or
            # to run two trial steps with it, every rank falls back together (rccl -> ipc -> the Python twin over torch.distributed)
            local_rows, local_nnz = shard.local_rows, shard.local_nnz

            def all_ok(ok):
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return bool(t.item())

            def try_exchange(kind):
                p, why = None, None
                try:
                    if kind == "ipc":
                        p = MG.MgPlan(shard.row_split, shard.nz_split, cols, tdt, [rank], [local_rank], exchange=MG.EXCHANGE_IPC)
                    else:
                        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                        if rank == 0:
                            idt.copy_(torch.frombuffer(bytearray(MG.unique_id()), dtype=torch.uint8))
                        dist.broadcast(idt, 0)
                        p = MG.MgPlan(shard.row_split, shard.nz_split, cols, tdt, [rank], [local_rank], exchange=MG.EXCHANGE_RCCL,
                                      id128=bytes(idt.cpu().numpy().tobytes()))
                    p.set_part(0, shard.values, shard.row_offsets, shard.column_indices)
                    blob = p.ipc_export() if kind == "ipc" else None
                except Exception as e:  # noqa: BLE001 - any failure means "fall back"
                    why = f"rank {rank} setup: {e}"
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
                    why = f"rank {rank} trial steps: {e}"
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
                sys.stderr.write(f"[bench] exchange {kind} not usable: {why[:300]}\n")
                if plan is not None:
                    try:
                        plan.close()
                    except Exception:  # noqa: BLE001
                        pass
                plan = None
            if plan is not None:
                exchange = plan.info()
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

        for _ in range(warmup):
            op()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            op()
        barrier()
        elapsed = time.perf_counter() - t0
        elapsed_local = elapsed
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ms_per_step = elapsed * 1e3 / steps

        # ---- per-kernel durations: the same K steps again with hipEvents on the launch stream
        M.profile_begin(steps)
        for _ in range(steps):
            op()
        op_sync()
        prof = M.profile_end()

        # ---- N > 1 through the C operator: what the step's exchange alone takes (events around it, per rank)
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

        # ---- N > 1, c5: rank 0 runs the WHOLE matrix alone on its GPU in the same job -----------------------------------
        single = None
        if mg and workload == "c5" and single_leg and not args.no_single_gpu_leg:
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
                single = {"n_gpus": 1, "steps": k, "ms_per_step": round(sms, 5), "value": round(2.0 * nnz_total / (sms * 1e-3) / 1e9, 3), "unit": "GFLOP/s"}
                if y0 is not None:
                    # (detail file) rank 0's tiles are the single-GPU call's first tiles, so its rows must match bit for bit when both run the
                    # same tile shape through the same path -- except inside rank 0's LAST tile, which ends where the part ends
                    ref = wy[:y0.numel()]
                    neq = (ref != y0).nonzero()
                    part_info = M.launch_info(local_rows, local_nnz, vb)
                    whole_info = M.launch_info(W.rows, W.nnz, vb)
                    same = (part_info["items_per_thread"] == whole_info["items_per_thread"] and part_info["snap_head_max"] == whole_info["snap_head_max"]
                            and M.band_passes(W.rows, W.cols, W.nnz, vb) <= 1 and M.band_passes(local_rows, cols, local_nnz, vb) <= 1)
                    single["rank0_rows_vs_single_gpu"] = {"rows": int(y0.numel()), "not_bitwise_equal": int(neq.numel()),
                                                          "first_differing_row": int(neq[0].item()) if neq.numel() else None,
                                                          "tile_items": int(part_info["tile_items"]), "same_tiling": bool(same),
                                                          "max_abs_diff": float((ref - y0).abs().max().item()) if y0.numel() else 0.0}
                del W, wws, wy

        sys.stdout.flush()
        per_rank = None
        if dist is not None:
            mine = torch.tensor([prof["search_ms"], prof["tile_ms"], prof["fixup_ms"], elapsed_local * 1e3 / steps, float(local_nnz),
                                 float("nan") if exchange_ms is None else exchange_ms], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            rows_ = torch.stack(allr).cpu().numpy()
            nanmax = lambda a: None if np.isnan(a).all() else round(float(np.nanmax(a)), 5)
            nanmin = lambda a: None if np.isnan(a).all() else round(float(np.nanmin(a)), 5)
            per_rank = {"tile_ms_max": round(float(rows_[:, 1].max()), 5), "tile_ms_min": round(float(rows_[:, 1].min()), 5),
                        "step_ms_max": round(float(rows_[:, 3].max()), 5), "step_ms_min": round(float(rows_[:, 3].min()), 5),
                        "nnz_per_rank_max": int(rows_[:, 4].max()), "nnz_per_rank_min": int(rows_[:, 4].min()),
                        "exchange_ms_max": nanmax(rows_[:, 5]), "exchange_ms_min": nanmin(rows_[:, 5]),
                        "note": "per rank: hipEvent averages of its kernels, its own wall time per step; exchange_ms = events right after the SpMV and right after "
                                "the all-gather + the owner's add (mspmv_mg_plan_exchange_ms, 5 separate steps), the wait for slower ranks included"}
            dist.barrier()
        if plan is not None:                 # (a second leg follows in the default N > 1 run: the part's buffers go back first)
            try:
                plan.close()
            except Exception:  # noqa: BLE001
                pass
            plan = None
        shard = None; sharded = None
        if mg:
            torch.cuda.empty_cache()
        if rank == 0:
            gflops = 2.0 * nnz_total / (ms_per_step * 1e-3) / 1e9
            info = M.launch_info(local_rows, local_nnz, vb)
            eff = effective_bytes(rows, nnz_total, vb) / (ms_per_step * 1e-3) / 1e9
            out = {
                "metric": "CsrMV GFLOP/s", "value": round(gflops, 3), "unit": "GFLOP/s",
                "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": scaling,
                "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
                "config": {"workload": desc, "tile": f"{info['block_threads']}x{info['items_per_thread']}",
                           "partition": ("single GPU" if not mg else
                                         f"merge-path diagonal split over {world} GPUs: {local_rows - 1} rows + {local_nnz} nnz on rank 0; one exchange of {world} carries per step")},
                "effective_GBs_reference_formula": round(eff, 1), "effective_pct_of_peak": round(100.0 * eff / HBM_PEAK_GBS, 2),
            }
            if not mg:
                label = "c2_f32" if (workload, dtype_name) == ("c2", "f32") else workload
                out["roofline"] = roofline_record(M, A, cols, prof, ms_per_step, ws, label, dtype_name)
                chk = M.sampled_check(A, x, y)
                out["sampled_worst_ratio"] = chk["worst_ratio"]
                out["sampled_check"] = chk
            else:
                b_alg = algorithmic_bytes(local_rows, cols, local_nnz, vb)
                tile_s = prof["tile_ms"] * 1e-3
                achieved = b_alg / tile_s / 1e9 if tile_s > 0 else 0.0
                out["roofline"] = {"kernel": "tile kernel of rank 0's part", "achieved": round(achieved, 1), **roofline_bound(M, b_alg, achieved),
                                   "algorithmic_bytes_per_launch": b_alg, "kernel_ms": round(prof["tile_ms"], 5), "traffic": None,
                                   "traffic_over_algorithmic": None, "traffic_src": None}
            if exchange is not None:
                ex = {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in exchange.items() if k != "steps"}
                if isinstance(exchange.get("exchange"), int):
                    ex["backend"] = {1: "rccl all-gather", 2: "peer reads", 3: "hipIpc mailbox"}.get(exchange["exchange"])
                out["exchange"] = ex
            if per_rank is not None:
                out["per_rank"] = per_rank
            if single is not None:
                out["single_gpu_same_workload"] = single
            full = None
            if args.full and not mg:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_full as full
                full.extend_headline(out, M, torch, G, A, x, y, ws, workload, dtype_name, args)
            if not mg and not args.no_cpu_baseline and A.nnz <= 400_000_000:
                out["cpu_baseline"] = cpu_baseline(A, x, "same " + workload.upper() + " matrix")
            if not mg and workload == "c2" and dtype_name == "f32" and not args.no_configs:
                del A, ws, y
                torch.cuda.empty_cache()
                out["configs"] = config_records(M, torch, G, dev, min(steps, 50), warmup, args.configs_budget if not args.full else 600.0,
                                                mtx_dir=args.mtx_dir, full=full)
            return out
        return None
    # N > 1 (default): config 2's matrix family, weak scaling -- the SAME per-GPU matrix as the N = 1 line, so the per-N values
    # of a scaling series are comparable -- and then BASELINE config 5 (one R-MAT matrix cut N ways, strong scaling) as a second leg
    primary = args.workload or "c2"
    out = measure(primary, args.steps, args.warmup)
    if mg and args.workload is None and not args.no_c5_leg:
        c5 = None
        try:
            c5 = measure("c5", min(args.steps, 20), min(args.warmup, 3), single_leg=args.c5_single_gpu_leg)
        except Exception as e:  # noqa: BLE001 - the headline is already measured: the leg must not cost it
            c5 = {"error": f"{type(e).__name__}: {e}"[:200]}
            sys.stderr.write(f"[bench] config 5 leg failed on rank {rank}: {c5['error']}\n")
        if rank == 0:
            out["c5_strong"] = c5
    if rank == 0:
        emit(out, detail_path)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
