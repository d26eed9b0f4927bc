"""bench.py --full: the legs that take minutes and do not belong in the driver's default run -- live L2<->fabric counters by two
rocprofv3 --pmc child runs per record, the rocSPARSE comparison column, the opt-in plans (band-major, hot columns) and config 5 on
one GPU.  Everything lands in the detail file; bench.compact_line keeps a few numbers of it."""
import csv
import glob
import os
import shutil
import signal
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0
_LIVE = {"ok": True, "why": ""}       # one failure (no rocprofv3, a crash, a time-out) turns the live passes off for the rest of the run


def live_traffic(label, steps=24, timeout_s=90, extra_args=()):
    """L2 <-> fabric bytes (Infinity-Cache hits included) per launch of the tile kernel, measured now on this box: two separate
    `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, then WRITE_SIZE: never combined with other trace domains) over
    `tools/run_config.py <label>`; corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes (KB -> bytes; gfx950's FETCH_SIZE
    tallies 128-byte requests at 64 bytes: doubled).  (bytes, source) or (None, why)."""
    if not _LIVE["ok"]:
        return None, "live counters off after an earlier failure: " + _LIVE["why"]
    tr, why = _live_traffic(label, steps, timeout_s, extra_args)
    if tr is None:
        _LIVE["ok"] = False; _LIVE["why"] = why
    return tr, why


def _live_traffic(label, steps, timeout_s, extra_args):
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="mspmv_pmc_", dir="/tmp")
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    got = {}
    try:
        for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, pmc)
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
                return None, f"rocprofv3 --pmc {pmc} took longer than {timeout_s} s"
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
    return int((2.0 * fetch_kb + write_kb) * 1024), f"live: FETCH_SIZE {fetch_kb:.0f} KB x 2 + WRITE_SIZE {write_kb:.0f} KB, {nf}/{nw} dispatches"


def vendor_record(torch, A, x, y_ours, iters):
    """rocSPARSE csrmv on the same device arrays -- the column the reference prints beside its own (cuSPARSE there,
    gpu_spmv.cu:262-364,565-578): analysis time apart, average SpMV time."""
    try:
        import rocsparse_ref
        ana, avg, yr = rocsparse_ref.time_csrmv(A, x, iters=iters)
        rec = {"library": "rocSPARSE csrmv after csrmv_analysis (tools/rocsparse_ref.py)", "analysis_ms": round(ana, 3), "ms_per_step": round(avg, 5),
               "steps": iters, "value": round(2.0 * A.nnz / (avg * 1e-3) / 1e9, 3), "unit": "GFLOP/s"}
        if y_ours is not None:
            rec["max_abs_diff_vs_ours"] = float((yr.double() - y_ours.double()).abs().max().item())
        return rec
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def _apply_live(roof, tr, src):
    if tr is not None:
        roof["traffic_replayed"] = roof.get("traffic")
        roof["traffic"] = tr
        roof["traffic_over_algorithmic"] = round(tr / roof["algorithmic_bytes_per_launch"], 3)
        roof["traffic_src"] = src
    else:
        roof["traffic_live_failed"] = src


def extend_headline(out, M, torch, G, A, x, y, ws, workload, dtype_name, args):
    label = "c2_f32" if (workload, dtype_name) == ("c2", "f32") else workload
    if workload != "c5":
        _apply_live(out["roofline"], *live_traffic(label))
    if workload == "c2":
        out["prepared_plan"] = M.plan_bench_record(A, x, y, steps=min(args.steps, 50), warmup=5, peak_gbs=HBM_PEAK_GBS)
    out["vendor"] = vendor_record(torch, A, x, y, min(args.steps, 30))


def extend_config(rec, M, torch, G, A, x, y, ws, label, x_seed, from_file, k):
    if not from_file:
        if label == "c5":
            # rocprofv3 --pmc dies while a child GENERATES the 2e9 edges: the matrix is parked as a raw image in /dev/shm for the two passes
            img = f"/dev/shm/mspmv_bench_c5_{os.getpid()}.img"
            try:
                vb = A.values.element_size()
                if shutil.disk_usage("/dev/shm").free > A.nnz * (vb + 4) + 4 * (A.rows + 1) + (1 << 30):
                    G.save_csr_image(A, x_seed, img)
                    _apply_live(rec["roofline"], *live_traffic(label, steps=5, timeout_s=240, extra_args=("--load", img)))
            except Exception as e:  # noqa: BLE001
                rec["roofline"]["traffic_live_failed"] = f"{type(e).__name__}: {e}"[:200]
            finally:
                try:
                    os.remove(img)
                except OSError:
                    pass
        else:
            _apply_live(rec["roofline"], *live_traffic(label))
    if label in ("c5", "c3_orkut"):
        rec["hot_column_plan"] = M.hotcols_bench_record(A, x, y, steps=k, warmup=2, peak_gbs=HBM_PEAK_GBS)
    rec["vendor"] = vendor_record(torch, A, x, y, 3 if label == "c5" else 5 if label == "c4" else min(k, 30))
    if "ms_per_step" in rec["vendor"]:
        rec["vendor"]["vendor_time_over_ours"] = round(rec["vendor"]["ms_per_step"] / rec["ms_per_step"], 3)
