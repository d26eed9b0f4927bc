"""oracle.py -- TEST INFRASTRUCTURE ONLY (the parity oracle's Python face).

ctypes access to oracle/liboracle.so (merge_oracle.c) plus numpy restatements
of the reference's host data model.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module; the product
(merge_spmv_amd/) never does.

Each function cites the reference lines it follows (paths relative to the
reference tree).  Pinning: tests/test_oracle_golden.py checks every function
here against tests/golden/*.json, which oracle/make_golden.py generated from
the reference's own code (oracle/_ref/ref_host = sparse_matrix.h + utils.h,
oracle/_ref/ref_search = cub::MergePathSearch) in the authoring container.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so (gcc) if missing; returns its path."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _ip(a):  # int32 pointer
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _suffix(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"unsupported value type {dtype}")


# ---------------------------------------------------------------------------
# CSR container -- the three-array layout of sparse_matrix.h:645-650
# ---------------------------------------------------------------------------
@dataclass
class Csr:
    rows: int
    cols: int
    row_offsets: np.ndarray      # int32 [rows+1], [0]=0, [rows]=nnz
    column_indices: np.ndarray   # int32 [nnz]
    values: np.ndarray           # f32/f64 [nnz]

    @property
    def nnz(self) -> int:
        return int(self.row_offsets[-1]) if self.row_offsets.size else 0

    @property
    def row_end_offsets(self) -> np.ndarray:
        """row_offsets + 1 (device_spmv.cuh:148, cpu_spmv.cpp:381)."""
        return self.row_offsets[1:]

    def astype(self, dtype) -> "Csr":
        return Csr(self.rows, self.cols, self.row_offsets, self.column_indices,
                   self.values.astype(dtype))


# ---------------------------------------------------------------------------
# COO generators -- sparse_matrix.h:386-617 (all values = default 1.0).
# Each returns (rows, cols, coo_row, coo_col, coo_val) in EMISSION order.
# ---------------------------------------------------------------------------
def coo_dense(rows: int, cols: int, dtype=np.float64):
    """InitDense, sparse_matrix.h:386-413: row-major full matrix."""
    r = np.repeat(np.arange(rows, dtype=np.int64), cols)
    c = np.tile(np.arange(cols, dtype=np.int64), rows)
    return rows, cols, r, c, np.ones(rows * cols, dtype=dtype)


def coo_wheel(spokes: int, dtype=np.float64):
    """InitWheel, sparse_matrix.h:419-452: hub row 0 -> 1..spokes, then rim
    vertex i+1 -> ((i+1) % spokes)+1."""
    i = np.arange(spokes, dtype=np.int64)
    r = np.concatenate([np.zeros(spokes, dtype=np.int64), i + 1])
    c = np.concatenate([i + 1, ((i + 1) % spokes) + 1])
    return spokes + 1, spokes + 1, r, c, np.ones(2 * spokes, dtype=dtype)


def coo_grid2d(width: int, self_loop: bool = False, dtype=np.float64):
    """InitGrid2d, sparse_matrix.h:461-526: neighbours emitted W, E, N, S
    (then self) for vertex me = j*width + k."""
    j, k = np.meshgrid(np.arange(width, dtype=np.int64),
                       np.arange(width, dtype=np.int64), indexing="ij")
    me = (j * width + k).ravel()
    jj, kk = j.ravel(), k.ravel()
    cand = [
        (kk - 1 >= 0, jj * width + (kk - 1)),
        (kk + 1 < width, jj * width + (kk + 1)),
        (jj - 1 >= 0, (jj - 1) * width + kk),
        (jj + 1 < width, (jj + 1) * width + kk),
    ]
    if self_loop:
        cand.append((np.ones_like(me, dtype=bool), me))
    return _interleave(width * width, me, cand, dtype)


def coo_grid3d(width: int, self_loop: bool = False, dtype=np.float64):
    """InitGrid3d, sparse_matrix.h:533-617: neighbours k-1, k+1, j-1, j+1,
    i-1, i+1 (then self) for me = i*w*w + j*w + k."""
    w = width
    i, j, k = np.meshgrid(*(np.arange(w, dtype=np.int64),) * 3, indexing="ij")
    ii, jj, kk = i.ravel(), j.ravel(), k.ravel()
    me = ii * w * w + jj * w + kk
    cand = [
        (kk - 1 >= 0, me - 1),
        (kk + 1 < w, me + 1),
        (jj - 1 >= 0, me - w),
        (jj + 1 < w, me + w),
        (ii - 1 >= 0, me - w * w),
        (ii + 1 < w, me + w * w),
    ]
    if self_loop:
        cand.append((np.ones_like(me, dtype=bool), me))
    return _interleave(w * w * w, me, cand, dtype)


def _interleave(n, me, cand, dtype):
    """Emit, vertex by vertex, the candidate neighbours whose guard holds."""
    mask = np.stack([m for m, _ in cand], axis=1)
    nbr = np.stack([v for _, v in cand], axis=1)
    rows = np.broadcast_to(me[:, None], mask.shape)[mask]
    cols = nbr[mask]
    return n, n, rows.astype(np.int64), cols.astype(np.int64), np.ones(rows.size, dtype=dtype)


# ---------------------------------------------------------------------------
# Matrix Market parser -- CooMatrix::InitMarket, sparse_matrix.h:217-380,
# with its quirks kept: banner flags by substring (symmetric/skew/array only,
# :265-267); lines capped at 1023 chars by getline(line,1024) (:244, a longer
# line sets failbit and ends parsing); indices via strtol base 0 (:330-345);
# missing value -> default (:351-355); 1-based -> 0-based (:357); symmetric
# mirrors off-diagonal entries, negated if skew (:362-368); array format is
# column-major (:321-324); nnz = entries actually produced (:373).
# ---------------------------------------------------------------------------
class MarketError(Exception):
    pass


def _strtol0(s: str, pos: int):
    """C strtol(..., base 0): skip whitespace, sign, 0x/0 prefixes."""
    n = len(s)
    i = pos
    while i < n and s[i] in " \t\n\v\f\r":
        i += 1
    j = i
    neg = False
    if j < n and s[j] in "+-":
        neg = s[j] == "-"
        j += 1
    base = 10
    if j + 1 < n and s[j] == "0" and s[j + 1] in "xX" and j + 2 < n and s[j + 2] in "0123456789abcdefABCDEF":
        base = 16
        j += 2
    elif j < n and s[j] == "0":
        base = 8
    digits = "0123456789abcdef"[:base]
    k = j
    while k < n and s[k].lower() in digits:
        k += 1
    if k == j:
        return None, pos
    v = int(s[j:k], base)
    return (-v if neg else v), k


def _sscanf_ints(s: str, n: int):
    """sscanf(s, "%d %d ...") : leading whitespace, optional sign, base 10."""
    import re
    out, pos = [], 0
    pat = re.compile(r"[ \t\n\v\f\r]*([+-]?\d+)")
    for _ in range(n):
        m = pat.match(s, pos)
        if not m:
            break
        out.append(int(m.group(1)))
        pos = m.end()
    return out


def _strtod(s: str, pos: int):
    import re
    m = re.compile(r"[ \t\n\v\f\r]*([+-]?(?:inf(?:inity)?|nan|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?))",
                   re.IGNORECASE).match(s, pos)
    if not m:
        return None, pos
    return float(m.group(1)), m.end()


def coo_market(path: str, default_value: float = 1.0, dtype=np.float64):
    with open(path, "r", newline="") as f:
        text = f.read()
    array = symmetric = skew = False
    current_nz = -1
    num_rows = num_cols = num_nonzeros = 0
    rr, cc, vv = [], [], []
    for raw in text.split("\n")[:-1]:
        # ifs.getline(line, 1024): a line of >= 1024 chars sets failbit -> stop.
        # A final line without '\n' sets eofbit -> !good() -> NOT processed
        # (:245-250), hence the [:-1] in both cases.
        if len(raw) >= 1024:
            break
        line = raw
        if line[:1] == "%":
            if line[1:2] == "%":
                symmetric = "symmetric" in line
                skew = "skew" in line
                array = "array" in line
            continue
        if current_nz == -1:
            ints = _sscanf_ints(line, 3)            # sscanf("%d %d %d") (:277)
            if (not array) and len(ints) == 3:
                num_rows, num_cols, num_nonzeros = ints
                if symmetric:
                    num_nonzeros *= 2
                current_nz = 0
            elif array and len(ints) == 2:
                num_rows, num_cols = ints
                num_nonzeros = num_rows * num_cols
                current_nz = 0
            else:
                raise MarketError(f"invalid problem description: {line}")
            continue
        if current_nz >= num_nonzeros:
            raise MarketError(f"encountered more than {num_nonzeros} num_nonzeros")
        if array:
            val, _ = _strtod(line, 0)
            if val is None:
                raise MarketError(f"badly formed current_nz: '{line}' at edge {current_nz}")
            col = current_nz // num_rows
            row = current_nz - num_rows * col
            rr.append(row); cc.append(col); vv.append(val)
            r1, c1 = row, col          # array entries are NOT shifted (:324)
        else:
            row, p = _strtol0(line, 0)
            if row is None:
                raise MarketError(f"badly formed row at edge {current_nz}")
            col, p = _strtol0(line, p)
            if col is None:
                raise MarketError(f"badly formed col at edge {current_nz}")
            val, p2 = _strtod(line, p)
            if val is None:
                val = default_value
            rr.append(row - 1); cc.append(col - 1); vv.append(val)
            r1, c1 = row, col
        current_nz += 1
        if symmetric and r1 != c1:
            rr.append(cc[-1]); cc.append(rr[-2]); vv.append(vv[-1] * (-1 if skew else 1))
            current_nz += 1
    if current_nz < 0:
        current_nz = 0
    return (num_rows, num_cols, np.asarray(rr, dtype=np.int64),
            np.asarray(cc, dtype=np.int64), np.asarray(vv, dtype=np.float64).astype(dtype))


# ---------------------------------------------------------------------------
# COO -> CSR -- CsrMatrix::Init, sparse_matrix.h:666-728: stable sort by
# (row, col) (:636-643,676), duplicates kept, offsets filled for empty rows
# including trailing ones (:707-727).
# ---------------------------------------------------------------------------
def csr_from_coo(rows, cols, coo_row, coo_col, coo_val) -> Csr:
    order = np.lexsort((coo_col, coo_row))          # stable, row-major keys
    r = coo_row[order]
    counts = np.bincount(r, minlength=rows).astype(np.int64) if r.size else np.zeros(rows, dtype=np.int64)
    off = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    return Csr(rows, cols, off.astype(np.int32), coo_col[order].astype(np.int32),
               np.ascontiguousarray(coo_val[order]))


def make(kind: str, *args, dtype=np.float64) -> Csr:
    gen = {"dense": coo_dense, "wheel": coo_wheel, "grid2d": coo_grid2d,
           "grid3d": coo_grid3d, "mtx": coo_market}[kind]
    return csr_from_coo(*gen(*args, dtype=dtype))


# ---------------------------------------------------------------------------
# Row-length statistics -- CsrMatrix::Stats, sparse_matrix.h:897-910, and the
# log-histogram buckets of DisplayHistogram, :919-956.
# ---------------------------------------------------------------------------
def stats(csr: Csr) -> dict:
    lengths = np.diff(csr.row_offsets.astype(np.int64)).astype(np.float64)
    mean = float(csr.nnz) / csr.rows
    delta = lengths - mean
    variance = float(np.sum(delta * delta)) / csr.rows
    std = float(np.sqrt(variance))
    with np.errstate(divide="ignore", invalid="ignore"):
        skew = float(np.float64(np.sum(delta * delta * delta) / csr.rows) / np.float64(std) ** 3)
        variation = float(np.float64(std) / np.float64(mean))
    return {"row_length_mean": mean, "row_length_std_dev": std,
            "row_length_variation": variation, "row_length_skewness": skew}


def _c_f5(v: float) -> str:
    """printf("%.5f") as glibc prints it, including the sign of a NaN
    (0.0/0.0 on x86 is the negative 'real indefinite' -> "-nan")."""
    if np.isnan(v):
        return "-nan" if np.signbit(v) else "nan"
    return "%.5f" % v


def stats_csv(csr: Csr) -> str:
    """GraphStats::Display(false), sparse_matrix.h:91-105."""
    st = stats(csr)
    return "%d, %d, %d, %s, %s, %s, %s, " % (
        csr.rows, csr.cols, csr.nnz, _c_f5(st["row_length_mean"]), _c_f5(st["row_length_std_dev"]),
        _c_f5(st["row_length_variation"]), _c_f5(st["row_length_skewness"]))


def histogram_text(csr: Csr) -> str:
    lengths = np.diff(csr.row_offsets.astype(np.int64))
    log_counts = [0] * 9
    max_log = -1
    max_len = -1
    for L in lengths.tolist():
        max_len = max(max_len, L)
        lg = -1
        while L > 0:
            L //= 10
            lg += 1
        max_log = max(max_log, lg)
        log_counts[lg + 1] += 1
    out = "CSR matrix (%d rows, %d columns, %d non-zeros, max-length %d):\n" % (
        csr.rows, csr.cols, csr.nnz, max_len)
    for i in range(-1, max_log + 1):
        pct = np.float32(log_counts[i + 1]) * 100.0 / csr.cols   # (float)count*100.0/num_cols (:953)
        out += "\tDegree 1e%d: \t%d (%.2f%%)\n" % (i, log_counts[i + 1], pct)
    return out


# ---------------------------------------------------------------------------
# ctypes faces of merge_oracle.c
# ---------------------------------------------------------------------------
def merge_path_search(diagonal: int, row_end: np.ndarray, rows: int, nnz: int):
    x = ctypes.c_int(); y = ctypes.c_int()
    row_end = np.ascontiguousarray(row_end, dtype=np.int32)
    keep = row_end if row_end.size else np.zeros(1, dtype=np.int32)
    lib().oracle_merge_path_search(int(diagonal), _ip(keep), int(rows), int(nnz),
                                   ctypes.byref(x), ctypes.byref(y))
    return x.value, y.value


def merge_path_search_i64(diagonal: int, row_end: np.ndarray, rows: int, nnz: int):
    x = ctypes.c_int64(); y = ctypes.c_int64()
    row_end = np.ascontiguousarray(row_end, dtype=np.int64)
    keep = row_end if row_end.size else np.zeros(1, dtype=np.int64)
    lib().oracle_merge_path_search_i64(ctypes.c_int64(diagonal), _vp(keep), ctypes.c_int64(rows),
                                       ctypes.c_int64(nnz), ctypes.byref(x), ctypes.byref(y))
    return x.value, y.value


def tile_coords(csr: Csr, step: int) -> np.ndarray:
    total = csr.rows + csr.nnz
    count = (total + step - 1) // step + 1
    out = np.zeros((count, 2), dtype=np.int32)
    re_ = np.ascontiguousarray(csr.row_end_offsets) if csr.rows else np.zeros(1, dtype=np.int32)
    lib().oracle_tile_coords(csr.rows, csr.nnz, _ip(re_), int(step), int(count), _ip(out))
    return out


def spmv_gold(csr: Csr, x: np.ndarray, y_in=None, alpha=1.0, beta=0.0) -> np.ndarray:
    sfx = _suffix(csr.values.dtype)
    T = csr.values.dtype
    x = np.ascontiguousarray(x, dtype=T)
    y_in = np.ones(csr.rows, dtype=T) if y_in is None else np.ascontiguousarray(y_in, dtype=T)
    y = np.empty(csr.rows, dtype=T)
    ct = ctypes.c_float if sfx == "f32" else ctypes.c_double
    getattr(lib(), f"oracle_spmv_gold_{sfx}")(
        csr.rows, _ip(csr.row_offsets), _ip(csr.column_indices), _vp(csr.values),
        _vp(x), _vp(y_in), _vp(y), ct(alpha), ct(beta))
    return y


def spmv_gold_acc64(csr: Csr, x: np.ndarray):
    """(g, s): fp64-accumulated y and sum |val*x| per row (strict check)."""
    sfx = _suffix(csr.values.dtype)
    x = np.ascontiguousarray(x, dtype=csr.values.dtype)
    g = np.empty(csr.rows, dtype=np.float64)
    s = np.empty(csr.rows, dtype=np.float64)
    getattr(lib(), f"oracle_spmv_gold_acc64_{sfx}")(
        csr.rows, _ip(csr.row_offsets), _ip(csr.column_indices), _vp(csr.values),
        _vp(x), _vp(g), _vp(s))
    return g, s


def omp_merge_csrmv(csr: Csr, x: np.ndarray, num_threads: int, y_fill=np.nan) -> np.ndarray:
    sfx = _suffix(csr.values.dtype)
    x = np.ascontiguousarray(x, dtype=csr.values.dtype)
    y = np.full(csr.rows, y_fill, dtype=csr.values.dtype)   # NaN sentinel, cf. memset(-1) cpu_spmv.cpp:380
    re_ = np.ascontiguousarray(csr.row_end_offsets) if csr.rows else np.zeros(1, dtype=np.int32)
    rc = getattr(lib(), f"oracle_omp_merge_csrmv_{sfx}")(
        int(num_threads), csr.rows, csr.nnz, _ip(re_), _ip(csr.column_indices),
        _vp(csr.values), _vp(x), _vp(y))
    if rc != 0:
        raise RuntimeError(f"oracle_omp_merge_csrmv failed rc={rc}")
    return y


def tiled_csrmv(csr: Csr, x: np.ndarray, tile_items: int):
    """(y, carry_keys, carry_vals) of the sequential tile emulation."""
    sfx = _suffix(csr.values.dtype)
    T = csr.values.dtype
    x = np.ascontiguousarray(x, dtype=T)
    y = np.full(csr.rows, np.nan, dtype=T)
    total = csr.rows + csr.nnz
    nt = max((total + tile_items - 1) // tile_items, 0)
    ck = np.zeros(max(nt, 1), dtype=np.int32)
    cv = np.zeros(max(nt, 1), dtype=T)
    re_ = np.ascontiguousarray(csr.row_end_offsets) if csr.rows else np.zeros(1, dtype=np.int32)
    rc = getattr(lib(), f"oracle_tiled_csrmv_{sfx}")(
        csr.rows, csr.nnz, _ip(re_), _ip(csr.column_indices), _vp(csr.values),
        _vp(x), _vp(y), int(tile_items), _ip(ck), _vp(cv))
    if rc < 0:
        raise RuntimeError(f"oracle_tiled_csrmv failed rc={rc}")
    return y, ck[:nt], cv[:nt]


def compare_results(computed: np.ndarray, reference: np.ndarray) -> int:
    """The reference's weak CompareResults (utils.h:692-742): 0 = PASS."""
    sfx = _suffix(computed.dtype)
    computed = np.ascontiguousarray(computed)
    reference = np.ascontiguousarray(reference, dtype=computed.dtype)
    bad = ctypes.c_int()
    return int(getattr(lib(), f"oracle_compare_results_{sfx}")(
        _vp(computed), _vp(reference), int(computed.size), ctypes.byref(bad)))


def max_threads() -> int:
    """OpenMP thread count worth using: the processor count capped by the container's CPU
    quota (cgroup cpu.max) -- a team larger than the quota gets throttled as a whole."""
    n = int(lib().oracle_max_threads())
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# ---------------------------------------------------------------------------
# Strict tolerance of SURVEY 8(d) / BASELINE.md 2: PASS iff for every row
# |y - g| <= c*eps*s with c = 2*(ceil(log2(len+1)) + items_per_thread + 8),
# eps = 2^-24 (fp32) / 2^-53 (fp64); empty rows exactly 0.
# Returns (ok, worst_ratio) where ratio = |y-g| / (c*eps*s) over rows with s>0.
# ---------------------------------------------------------------------------
def strict_check(csr: Csr, y: np.ndarray, g: np.ndarray, s: np.ndarray,
                 items_per_thread: int = 16):
    eps = 2.0 ** -24 if csr.values.dtype == np.float32 else 2.0 ** -53
    lens = np.diff(csr.row_offsets.astype(np.int64))
    c = 2.0 * (np.ceil(np.log2(lens + 1.0)) + items_per_thread + 8)
    bound = c * eps * s
    err = np.abs(y.astype(np.float64) - g)
    empty = lens == 0
    ok_empty = bool(np.all(y[empty] == 0)) if empty.any() else True
    nz = ~empty
    finite = bool(np.all(np.isfinite(y[nz]))) if nz.any() else True
    ok = ok_empty and finite and bool(np.all(err[nz] <= bound[nz]))
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(bound > 0, err / bound, np.where(err == 0, 0.0, np.inf))
    worst = float(ratio[nz].max()) if nz.any() else 0.0
    return ok, worst
