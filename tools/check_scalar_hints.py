#!/usr/bin/env python3
"""tools/check_scalar_hints.py -- the one-launch kernel requests its tile hints with two scalar loads in one asm statement and
waits for them in a later one (merge_spmv_amd/csrc/mspmv_kernels.hpp, tile_kernel_snap).  Nothing may read the destination
registers in between; if the compiler ever copied them there the hints would be garbage (harmless -- they are verified -- but
every tile would then search its boundaries).  This compiles the device code to assembly and checks every such kernel.
No GPU needed: python tools/check_scalar_hints.py"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "api.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                    os.path.join(root, "merge_spmv_amd", "csrc", "mspmv_api.hip")], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
cur, checked, bad, i = None, 0, 0, 0
while i < len(lines):
    m = re.match(r"^(_ZN5mspmv16tile_kernel_snap\S+):", lines[i])
    if m: cur = m.group(1)
    if cur and "ASMSTART" in lines[i] and i + 2 < len(lines) and "s_load_dwordx4" in lines[i + 1] and "s_load_dwordx2" in lines[i + 2]:
        d4 = re.search(r"s_load_dwordx4 s\[(\d+):(\d+)\]", lines[i + 1]); d2 = re.search(r"s_load_dwordx2 s\[(\d+):(\d+)\]", lines[i + 2])
        regs = set(range(int(d4.group(1)), int(d4.group(2)) + 1)) | set(range(int(d2.group(1)), int(d2.group(2)) + 1))
        j, touched, waited = i + 3, [], False
        while j < len(lines) and not ("ASMSTART" in lines[j] and "s_waitcnt lgkmcnt(0)" in lines[j + 1]):
            t = lines[j]
            if "s_waitcnt" in t and "lgkmcnt(0)" in t: waited = True          # (a wait the compiler needed anyway also lands them)
            if not waited and not t.strip().startswith(";"):
                used = {int(r) for r in re.findall(r"\bs(\d+)\b", t)}
                for a, b in re.findall(r"s\[(\d+):(\d+)\]", t): used |= set(range(int(a), int(b) + 1))
                if used & regs: touched.append(t.strip())
            j += 1
        checked += 1
        if touched:
            bad += 1; print(cur, "reads the hint registers before they have landed:", touched[:3])
        i = j
    i += 1
print(f"{checked} scalar-hint kernels checked, {bad} with the destination registers read before a wait")
sys.exit(1 if bad or checked == 0 else 0)
