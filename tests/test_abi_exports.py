"""CPU-side checks of the drop-in boundary: libmspmv.so loads and exports
every symbol include/mspmv.h declares; the host-only entry points (size
query, launch info, multi-GPU partitioner) behave.  No kernel is launched."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
import merge_spmv_amd as M
from oracle import oracle as O


def declared_symbols(header="mspmv.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef MSPMV_DEV\n.*?#endif\n", "", text, flags=re.S)          # (the -DMSPMV_DEV experiment build's one extra entry point)
    return sorted(set(re.findall(r"\b(mspmv_[a-z0-9_]+)\s*\(", text)))


def exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if " T mspmv_" in l)


def test_library_exports_every_declared_symbol():
    lib = M.load_library()
    names = declared_symbols()
    assert "mspmv_csrmv_f32" in names and "mspmv_csrmv_f64" in names and len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mspmv_version() == 102
    # ... and nothing else: what is exported is what include/mspmv.h declares
    assert exported(M.library_path("product")) == names


def test_the_product_library_has_no_setters_and_reads_no_environment():
    """the forcing / re-tuning state lives in libmspmv_dev.so (include/mspmv_dev.h) only.  The product exports no
    mspmv_set_* symbol, does not import getenv, and holds no MSPMV_* variable name; the development library exports exactly the
    product's symbols plus the four setters."""
    import subprocess
    prod, dev = M.library_path("product"), M.library_path("dev")
    names = exported(prod)
    assert not [n for n in names if n.startswith("mspmv_set_") or n.startswith("mspmv_dev_")]
    undefined = subprocess.run(["nm", "-D", "--undefined-only", prod], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", dev], capture_output=True, text=True, check=True).stdout
    strings = subprocess.run(["strings", prod], capture_output=True, text=True, check=True).stdout
    assert not re.findall(r"^MSPMV_[A-Z0-9_]+$", strings, flags=re.M)
    setters = ["mspmv_set_band_passes", "mspmv_set_compact_tiles", "mspmv_set_record_polls", "mspmv_set_tdm", "mspmv_set_tuning"]
    assert exported(dev) == sorted(names + setters)
    assert sorted(set(declared_symbols("mspmv_dev.h")) - set(names)) == setters


def test_setters_switch_to_the_development_library_and_defaults_do_not():
    assert M.active_library() == "product"
    M.set_tuning(4); M.set_band_passes(8, 0); M.set_tdm(4, 0); M.set_record_polls(0); M.set_compact_tiles(0)        # defaults: no-ops on the product
    assert M.active_library() == "product"
    M.set_tuning(4, 256, 11)
    assert M.active_library() == "dev" and M.launch_info(10, 10, 4)["items_per_thread"] == 11
    M.set_tuning(4)
    assert M.use_library("product") == "dev"
    assert M.launch_info(10, 10, 4)["items_per_thread"] == 7
    with pytest.raises(ValueError):
        M.use_library("nope")


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(M, "_libs", {})
    monkeypatch.setattr(M, "_LIB_NAMES", {"product": "libmspmv_does_not_exist.so", "dev": "libmspmv_dev.so"})
    monkeypatch.delenv("MSPMV_LIB", raising=False)
    with pytest.raises(M.MspmvError):
        M.load_library()


def test_size_query_two_phase_convention():
    """d_temp == NULL -> size only, success, no work (dispatch_spmv_orig.cuh:651-655);
    too small -> hipErrorInvalidValue (util_device.cuh:90-93)."""
    lib = M.load_library()
    for fn, vb in ((lib.mspmv_csrmv_f32, 4), (lib.mspmv_csrmv_f64, 8)):
        size = ctypes.c_size_t(0)
        st = fn(None, ctypes.byref(size), None, None, None, None, None, 1000, 1000, 50000, None, 0)
        assert st == 0 and size.value > 0
        info = M.launch_info(1000, 50000, vb)
        assert info["temp_bytes"] == size.value
        assert info["tile_items"] == info["block_threads"] * info["items_per_thread"]
        assert info["num_tiles"] == -(-51000 // info["tile_items"])
        small = ctypes.c_size_t(size.value - 1)
        st = fn(ctypes.c_void_p(256), ctypes.byref(small), None, None, None, None, None, 1000, 1000, 50000, None, 0)
        assert st == 1  # hipErrorInvalidValue
        # temp storage must be 16-byte aligned (64-bit atomic records, scalar loads): refused before anything is touched
        big = ctypes.c_size_t(size.value + 64)
        fake = ctypes.c_void_p(4096)
        for misaligned in (4096 + 1, 4096 + 4, 4096 + 8):
            st = fn(ctypes.c_void_p(misaligned), ctypes.byref(big), fake, fake, fake, fake, fake, 1000, 1000, 50000, None, 0)
            assert st == 1
        st = fn(None, ctypes.byref(size), None, None, None, None, None, -1, 5, 5, None, 0)
        assert st == 1
        st = fn(None, ctypes.byref(size), None, None, None, None, None, 2**30, 5, 2**30 + 5, None, 0)
        assert st == 1  # rows + nnz must stay below 2^31
        st = fn(None, ctypes.byref(size), None, None, None, None, None, 1000, 5, 2**31 - 1 - 65536 - 1000 + 1, None, 0)
        assert st == 1  # ... with one tile of slack for the int32 chunk arithmetic of the staging
        st = fn(None, ctypes.byref(size), None, None, None, None, None, 1000, 5, 2**31 - 1 - 65536 - 1000, None, 0)
        assert st == 0


def test_tuning_rejects_unknown_shapes():
    """(the development library: the setters switch to it)"""
    M.set_tuning(4, 256, 11)
    assert M.launch_info(10, 10, 4)["items_per_thread"] == 11
    M.set_tuning(4)
    assert M.launch_info(10, 10, 4)["items_per_thread"] == 7
    # default shapes: fp32 256x7 while that keeps the problem within 2304 tiles (1024 until round 5), 256x11 beyond; fp64 256x7 up to
    # 8 M path items, 256x11 beyond.  Always ONE launch of row-snapped tiles (tile_kernel_snap, no fix-up) unless
    # MSPMV_TUNE_TWO_LAUNCH asks for the classic three
    assert M.launch_info(300_000, 1_000_000, 4)["items_per_thread"] == 7         # 1.3M items / 1792 = 726 tiles
    assert M.launch_info(300_000, 1_500_000, 4)["items_per_thread"] == 7         # 1005 tiles
    assert M.launch_info(300_000, 2_200_000, 4)["items_per_thread"] == 7         # 1396 tiles
    assert M.launch_info(300_000, 3_700_000, 4)["items_per_thread"] == 7         # 2233 tiles
    assert M.launch_info(300_000, 4_000_000, 4)["items_per_thread"] == 11        # 2400 tiles of 256x7: the large-problem shape
    info = M.launch_info(1_000_000, 3_500_000, 4)
    assert info["items_per_thread"] == 11 and info["fixup_levels"] == 0 and info["snap_head_max"] == 192
    M.set_tuning(4, 0, 0, 0x40000000)                                            # the classic three launches: one fix-up launch
    info = M.launch_info(1_000_000, 3_500_000, 4)
    assert info["fixup_levels"] == 1 and info["snap_head_max"] == 0
    M.set_tuning(4, 0, 0, 128)                                                   # the chunked multi-level variant
    assert M.launch_info(1_000_000, 3_500_000, 4)["fixup_levels"] == 2           # 1599 carries / 512 per block -> 2 launches
    M.set_tuning(4, 0, 0, 16)                                                    # the large-problem shape whatever the size
    assert M.launch_info(1000, 5000, 4)["items_per_thread"] == 11
    M.set_tuning(4)
    # the override is per host thread: another thread sees the defaults
    import threading
    M.set_tuning(4, 256, 11)
    seen = {}
    th = threading.Thread(target=lambda: seen.update(ipt=M.launch_info(10, 10, 4)["items_per_thread"]))
    th.start(); th.join()
    assert seen["ipt"] == 7 and M.launch_info(10, 10, 4)["items_per_thread"] == 11
    M.set_tuning(4)
    assert M.launch_info(1_000_000, 5_000_000, 8)["items_per_thread"] == 7       # fp64 up to 8 M path items: 256x7
    assert M.launch_info(1000, 5000, 8)["items_per_thread"] == 7
    assert M.launch_info(4_000_000, 30_000_000, 8)["items_per_thread"] == 11
    assert M.launch_info(3_125_000, 100_000_000, 4)["items_per_thread"] == 11
    assert M.launch_info(3_125_000, 100_000_000, 8)["items_per_thread"] == 11
    with pytest.raises(M.MspmvError):
        M.set_tuning(4, 250, 7)
    # the other shapes of the tuning sweeps exist only in the development build
    for vb, b, i in ((4, 128, 7), (4, 512, 7), (4, 256, 5), (4, 256, 9), (4, 256, 15), (8, 128, 5), (8, 512, 5), (8, 256, 3), (8, 256, 5), (8, 256, 9)):
        with pytest.raises(M.MspmvError):
            M.set_tuning(vb, b, i)
    # the product library has no timing-experiment kernels: their flag bits (persistent grid, staging-only
    # ablation that returns wrong y, cycle stamps, XCD remap; include/mspmv_dev.h) are rejected
    for dev_bits in (1, 0x100, 0x10000, 0x60000, 0x70000, 0x100000):
        with pytest.raises(M.MspmvError):
            M.set_tuning(4, 0, 0, dev_bits)
        with pytest.raises(M.MspmvError):
            M.set_tuning(8, 256, 11, dev_bits | 16)
    assert not hasattr(M.load_library(), "mspmv_dev_set_trace")
    assert M.launch_info(10, 10, 4)["flags"] == 0
    M.use_library("product")
    assert M.launch_info(10, 10, 4)["flags"] == 0


@pytest.mark.parametrize("parts", [1, 2, 3, 4, 8])
def test_mg_partition_matches_oracle_search(parts):
    """The multi-GPU cut points are merge-path coordinates of equally spaced
    diagonals (64-bit MergePathSearch), and the shards tile the matrix."""
    from merge_spmv_amd import multi_gpu as MG
    rng = np.random.default_rng(parts)
    lens = rng.integers(0, 9, size=200)
    lens[50] = 700                        # a row spanning several parts
    lens[120:140] = 0
    off = np.zeros(201, dtype=np.int64); np.cumsum(lens, out=off[1:])
    rows, nnz = 200, int(off[-1])
    row_split, nz_split = MG.partition(off, parts)
    per = -(-(rows + nnz) // parts)
    for g in range(parts + 1):
        d = min(per * g, rows + nnz)
        assert (row_split[g], nz_split[g]) == O.merge_path_search_i64(d, off[1:], rows, nnz)
    assert (row_split[0], nz_split[0]) == (0, 0) and (row_split[-1], nz_split[-1]) == (rows, nnz)
    # local CSR of every part: offsets rebased, one extra open row
    for g in range(parts):
        lo = MG.local_offsets(off, row_split[g], row_split[g + 1], nz_split[g], nz_split[g + 1])
        assert lo[0] == 0 and lo[-1] == nz_split[g + 1] - nz_split[g]
        assert lo.size == row_split[g + 1] - row_split[g] + 2
        assert np.all(np.diff(lo) >= 0)


def test_extension_entry_points_follow_the_same_conventions():
    """mspmv_csrmv_prepare / mspmv_csrmm_*: NULL temp -> size query and nothing else; bad arguments ->
    hipErrorInvalidValue; prepare asks for exactly the CsrMV temp size (the two share the buffer)."""
    lib = M.load_library()
    size = ctypes.c_size_t(0)
    for vb in (4, 8):
        assert lib.mspmv_csrmv_prepare(None, ctypes.byref(size), None, 1000, 50000, vb, None, 0) == 0
        assert size.value == M.launch_info(1000, 50000, vb)["temp_bytes"]
    assert lib.mspmv_csrmv_prepare(None, ctypes.byref(size), None, 1000, 50000, 2, None, 0) == 1
    for fn in (lib.mspmv_csrmm_f32, lib.mspmv_csrmm_f64):
        sizes = []
        for k, cols in ((2, 1000), (4, 1000), (16, 1000), (16, 10_000_000)):      # narrow packs ... 64-byte packs (X > 1 MiB)
            st = fn(None, ctypes.byref(size), None, None, None, None, k, None, k, 1000, cols, 50000, k, 1.0, 0.0, None, 0)
            assert st == 0 and size.value > 0
            sizes.append(size.value)
        assert sizes[1] >= sizes[0] and sizes[3] > sizes[2]          # more / wider carries, smaller tiles
        # 8 M path items and more: groups of 8 / 16 columns run the slot form on its own 256 x 11 tiles whatever X's footprint
        big = []
        for cols in (1000, 10_000_000):
            assert fn(None, ctypes.byref(size), None, None, None, None, 16, None, 16, 1_000_000, cols, 9_000_000, 16, 1.0, 0.0, None, 0) == 0
            big.append(size.value)
        assert big[0] == big[1]
        assert fn(None, ctypes.byref(size), None, None, None, None, 3, None, 4, 1000, 1000, 50000, 4, 1.0, 0.0, None, 0) == 1   # ldx < k
        assert fn(None, ctypes.byref(size), None, None, None, None, 4, None, 4, -1, 1000, 5, 4, 1.0, 0.0, None, 0) == 1
        small = ctypes.c_size_t(16)
        assert fn(ctypes.c_void_p(256), ctypes.byref(small), None, None, None, None, 4, None, 4, 1000, 1000, 50000, 4, 1.0, 0.0, None, 0) == 1
    assert lib.mspmv_csrmv_prepared_f32(None, ctypes.byref(size), None, None, None, None, None, 10, 10, 10, 1.0, 0.0, None, 0) == 1   # needs a prepared buffer


def test_headers_are_plain_c(tmp_path):
    """include/mspmv.h (and the development header) must compile as C99 with no HIP or C++ in sight: that is what a cgo /
    JNI / ctypes-generator binding of the boundary consumes."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "include/mspmv.h"\n#include "include/mspmv_dev.h"\n'
                   "int use(void) { size_t b = 0; int32_t n = 0; mspmv_mg_info_t info; mspmv_launch_info_t li;\n"
                   "  int st = mspmv_csrmv_f64(0, &b, 0, 0, 0, 0, 0, 10, 10, 10, 0, 0);\n"
                   "  st |= mspmv_csrmv_plan_size(10, 10, 10, 4, 0, &b, &n); st |= mspmv_mg_plan_info(0, &info);\n"
                   "  st |= mspmv_get_launch_info(10, 10, 4, &li); return st; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-o", str(tmp_path / "hdr.o"), "-I", ROOT],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
