#!/usr/bin/env python3
"""tools/corpus_summary.py <dir with corpus_fp64.csv / corpus_fp32.csv / corpus_checks.txt> -- what the reference reports for its corpus
sweep (README.md:152-160, the paper's Fig. 10 / Table 1 rows in BASELINE.md): per decade of nonzeros the harmonic mean of the GFLOP/s
of each method, plus -- this project's bar -- the distribution of the algorithmic-bytes roofline fraction (B_alg / t / 8 TB/s),
the files where the merge-based CsrMV is more than 10 % slower than the vendor column, and the strict-check verdicts."""
import math
import os
import sys

PEAK = 8000.0e9


def is_num(s):
    try:
        float(s); return True
    except ValueError:
        return False


def parse(path, vb):
    rows = []
    for line in open(path):
        f = [t.strip() for t in line.strip().rstrip(",").split(",")]
        if len(f) < 13 or not is_num(f[1]):
            continue
        rec = {"file": os.path.basename(f[0]), "rows": int(f[1]), "cols": int(f[2]), "nnz": int(f[3]), "mean": float(f[4]), "cv": float(f[6]), "skew": float(f[7]),
               "methods": {}}
        i = 8
        while i < len(f):
            if not is_num(f[i]) and i + 4 < len(f) + 1 and all(is_num(t) for t in f[i + 1:i + 5]) and len(f[i + 1:i + 5]) == 4:
                rec["methods"][f[i]] = {"setup_ms": float(f[i + 1]), "ms": float(f[i + 2]), "gflops": float(f[i + 3]), "eff_gbs": float(f[i + 4])}
                i += 5
            else:
                i += 1
        b_alg = rec["nnz"] * (vb + 4) + (rec["rows"] + 1) * 4 + rec["rows"] * vb + rec["cols"] * vb
        ours = rec["methods"].get("Merge-based CsrMV")
        if ours and ours["ms"] > 0:
            rec["frac"] = b_alg / (ours["ms"] * 1e-3) / PEAK
            rows.append(rec)
    return rows


def hmean(v):
    v = [x for x in v if x > 0]
    return len(v) / sum(1.0 / x for x in v) if v else float("nan")


def quantiles(v, qs=(0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0)):
    v = sorted(v)
    return [v[min(int(q * (len(v) - 1) + 0.5), len(v) - 1)] for q in qs] if v else []


def main():
    d = sys.argv[1]
    print("# corpus sweep summary (tools/corpus_sweep.sh; files: tools/make_corpus.py --list) -- MI355X, reference protocol (gpu_spmv.cu:401-434), x = 1")
    for prec, vb in (("fp64", 8), ("fp32", 4)):
        path = os.path.join(d, f"corpus_{prec}.csv")
        if not os.path.exists(path):
            continue
        recs = parse(path, vb)
        if not recs:
            print(f"## {prec}: no lines"); continue
        vendor_name = next((m for m in recs[0]["methods"] if "rocSPARSE" in m and "Csr" in m), None)
        print(f"\n## {prec}: {len(recs)} files, {sum(r['nnz'] for r in recs) / 1e9:.2f} G nonzeros; row-length CV {min(r['cv'] for r in recs):.2f} ... {max(r['cv'] for r in recs):.1f}, "
              f"nonzeros {min(r['nnz'] for r in recs)} ... {max(r['nnz'] for r in recs)}")
        print("nonzeros (decade)   files   hmean GFLOP/s ours   hmean GFLOP/s rocSPARSE csrmv   median roofline frac   min frac   max frac")
        for lo in range(4, 9):                  # (files are generated at 1 and 3 x 10^k nonzeros, give or take rounding: decades cut at 5 x 10^(k-1))
            sel = [r for r in recs if 0.5 * 10 ** lo <= r["nnz"] < 5 * 10 ** lo]
            if not sel:
                continue
            ven = [r["methods"][vendor_name]["gflops"] for r in sel if vendor_name in r["methods"]] if vendor_name else []
            fr = sorted(r["frac"] for r in sel)
            print(f"~1e{lo} .. 3e{lo}       {len(sel):5d}   {hmean([r['methods']['Merge-based CsrMV']['gflops'] for r in sel]):18.1f}   {hmean(ven):29.1f}   "
                  f"{fr[len(fr) // 2]:20.3f}   {fr[0]:8.3f}   {fr[-1]:8.3f}")
        allg = [r["methods"]["Merge-based CsrMV"]["gflops"] for r in recs]
        print(f"all files: harmonic mean {hmean(allg):.1f} GFLOP/s ours" + (f", {hmean([r['methods'][vendor_name]['gflops'] for r in recs if vendor_name in r['methods']]):.1f} rocSPARSE csrmv" if vendor_name else ""))
        big = [r for r in recs if r["nnz"] >= 10_000_000]
        q = quantiles([r["frac"] for r in big])
        if q:
            print(f"roofline fraction (B_alg / t / 8 TB/s) of the {len(big)} files with >= 10 M nonzeros: min {q[0]:.3f}  p10 {q[1]:.3f}  p25 {q[2]:.3f}  median {q[3]:.3f}  p75 {q[4]:.3f}  p90 {q[5]:.3f}  max {q[6]:.3f}")
        # by row-length variation (the paper's axis): does the rate depend on it?
        print("row-length CV      files (>= 1 M nnz)   median frac   hmean GFLOP/s ours   hmean rocSPARSE")
        for lo, hi in ((0, 0.1), (0.1, 0.5), (0.5, 2), (2, 10), (10, 1e9)):
            sel = [r for r in recs if lo <= r["cv"] < hi and r["nnz"] >= 1_000_000]
            if sel:
                fr = sorted(r["frac"] for r in sel)
                ven = hmean([r["methods"][vendor_name]["gflops"] for r in sel if vendor_name in r["methods"]]) if vendor_name else float("nan")
                print(f"{lo:5g} .. {hi:<8g} {len(sel):8d}             {fr[len(fr) // 2]:10.3f}   {hmean([r['methods']['Merge-based CsrMV']['gflops'] for r in sel]):18.1f}   {ven:15.1f}")
        if vendor_name:
            slow = [(r["methods"]["Merge-based CsrMV"]["ms"] / r["methods"][vendor_name]["ms"], r) for r in recs if vendor_name in r["methods"] and r["methods"][vendor_name]["ms"] > 0]
            worse = sorted([s for s in slow if s[0] > 1.0 / 0.9], key=lambda s: -s[0])
            ahead = sum(1 for s in slow if s[0] < 1.0)
            print(f"against {vendor_name} (its analysis not counted): ours faster on {ahead} of {len(slow)} files; below 0.9 x the vendor's rate on {len(worse)}:")
            for ratio, r in worse[:25]:
                print(f"    {r['file']:<28s} nnz {r['nnz']:>10d}  CV {r['cv']:7.2f}  ours {r['methods']['Merge-based CsrMV']['ms']:.5f} ms  vendor {r['methods'][vendor_name]['ms']:.5f} ms  (x{ratio:.2f})")
            best = sorted(slow, key=lambda s: s[0])[:5]
            print("    largest advantages: " + "; ".join(f"{r['file']} x{1.0 / ratio:.1f}" for ratio, r in best))
    chk = os.path.join(d, "corpus_checks.txt")
    if os.path.exists(chk):
        lines = [l.split(",") for l in open(chk) if l.startswith("strict-check")]
        fails = [l for l in lines if len(l) > 3 and l[3].strip() != "PASS"]
        worst = max((float(l[5]) for l in lines if len(l) > 5 and is_num(l[5].strip())), default=float("nan"))
        print(f"\n## strict check (|y - g| <= 2 (ceil(log2(len + 1)) + 16 + 8) eps s per row, empty rows exactly 0): {len(lines) - len(fails)} of {len(lines)} PASS, worst |error| / bound {worst:.3g}")
        for l in fails[:20]:
            print("    FAIL " + ",".join(l).strip())
        other = [l.strip() for l in open(chk) if not l.startswith("strict-check") and l.strip() and "amdgpu.ids" not in l]
        if other:
            print(f"    ({len(other)} other stderr lines, first: {other[0][:160]})")


if __name__ == "__main__":
    main()
