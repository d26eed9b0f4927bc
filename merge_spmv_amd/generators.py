"""Synthetic CSR workloads of BASELINE.json / SURVEY.md 8(d), generated with
torch ops so they can be produced directly in HBM (or on the CPU for tests)
from a counter-based RNG: element i of stream `seed` is splitmix64(seed + i),
so a matrix is independent of device, chunking and GPU count.

The reference has no random generators (its --dense/--grid2d/--grid3d/--wheel
matrices, sparse_matrix.h:386-617, all carry 1.0 values and are reproduced in
merge_spmv_amd/host/sparse_matrix.hpp); these are the "new" inputs SURVEY.md
7(2) asks for: C2 uniform, C3 power-law / R-MAT stand-ins, C4 degenerate,
C5 R-MAT.  Output layout is the reference's CSR (sparse_matrix.h:645-650):
int32 row_offsets[rows+1], int32 column_indices[nnz] sorted by (row, col) with
duplicates kept, values[nnz].
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

SEED_C2 = 0x5EED0002
SEED_C3 = 0x5EED0003
SEED_C4 = 0x5EED0004
SEED_C5 = 0x5EED0005

_M64 = (1 << 64) - 1


def _i64(v: int) -> int:
    """two's-complement int64 view of a 64-bit constant"""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    """logical shift right on int64 lanes"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def splitmix64(seed: int, index: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser of (seed + (index+1)*golden) as int64 bit patterns.
    index: int64 tensor of counters."""
    z = index * _i64(0x9E3779B97F4A7C15) + _i64(seed + 0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _i64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def uniform01(seed: int, index: torch.Tensor) -> torch.Tensor:
    """double in [0,1): top 53 bits of splitmix64"""
    return _lsr(splitmix64(seed, index), 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def uniform_pm1(seed: int, count: int, dtype, device, start: int = 0, chunk: int = 1 << 26) -> torch.Tensor:
    """values uniform in [-1, 1) (cast from double), counters start..start+count"""
    out = torch.empty(count, dtype=dtype, device=device)
    for a in range(0, count, chunk):
        b = min(a + chunk, count)
        idx = torch.arange(start + a, start + b, dtype=torch.int64, device=device)
        out[a:b] = (uniform01(seed, idx) * 2.0 - 1.0).to(dtype)
    return out


@dataclass
class DeviceCsr:
    rows: int
    cols: int
    row_offsets: torch.Tensor
    column_indices: torch.Tensor
    values: torch.Tensor

    @property
    def nnz(self) -> int:
        return int(self.values.numel())


def uniform_csr(rows: int, cols: int, nnz_per_row: int, dtype=torch.float32, device="cuda",
                seed: int = SEED_C2, chunk_rows: int = 1 << 19, row_lo: int = 0,
                row_hi: Optional[int] = None) -> DeviceCsr:
    """C2: exactly nnz_per_row entries per row, columns i.i.d. uniform over
    [0, cols) sorted within the row (duplicates kept), values uniform [-1,1)
    from stream seed+1.  (x for the run: uniform_pm1(seed + 2, cols, ...).)
    row_lo/row_hi restrict the output to a row range of the SAME global matrix
    (counters are global), so ranks can build their swaths independently."""
    row_hi = rows if row_hi is None else row_hi
    n_rows = row_hi - row_lo
    nnz = n_rows * nnz_per_row
    cols_out = torch.empty(nnz, dtype=torch.int32, device=device)
    for r0 in range(row_lo, row_hi, chunk_rows):
        r1 = min(r0 + chunk_rows, row_hi)
        idx = torch.arange(r0 * nnz_per_row, r1 * nnz_per_row, dtype=torch.int64, device=device)
        c = (uniform01(seed, idx) * cols).to(torch.int64).clamp_(max=cols - 1)
        c = c.view(r1 - r0, nnz_per_row).sort(dim=1).values
        cols_out[(r0 - row_lo) * nnz_per_row: (r1 - row_lo) * nnz_per_row] = c.reshape(-1).to(torch.int32)
    offsets = (torch.arange(n_rows + 1, dtype=torch.int64, device=device) * nnz_per_row).to(torch.int32)
    vals = uniform_pm1(seed + 1, nnz, dtype, device, start=row_lo * nnz_per_row)
    return DeviceCsr(n_rows, cols, offsets, cols_out, vals)


def _grid_csr(n: int, me: torch.Tensor, cand, dtype, device) -> DeviceCsr:
    """CSR of a lattice: for vertex `me` the neighbours of `cand` (ascending column order) whose guard holds; values 1."""
    mask = torch.stack([m for m, _ in cand], dim=1)
    nbr = torch.stack([v for _, v in cand], dim=1)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(mask.sum(dim=1), 0, out=offsets[1:])
    cols = nbr[mask].to(torch.int32)
    return DeviceCsr(n, n, offsets.to(torch.int32), cols, torch.ones(cols.numel(), dtype=dtype, device=device))


def grid2d_csr(width: int, dtype=torch.float64, device="cuda") -> DeviceCsr:
    """The reference's --grid2d=W input as CSR (InitGrid2d without self loops, sparse_matrix.h:461-526, after its COO -> CSR
    conversion: columns ascending within a row): W*W vertices, 4-neighbour lattice, every value 1."""
    j = torch.arange(width, dtype=torch.int64, device=device).repeat_interleave(width)
    k = torch.arange(width, dtype=torch.int64, device=device).repeat(width)
    me = j * width + k
    cand = [(j - 1 >= 0, me - width), (k - 1 >= 0, me - 1), (k + 1 < width, me + 1), (j + 1 < width, me + width)]
    return _grid_csr(width * width, me, cand, dtype, device)


def grid3d_csr(width: int, dtype=torch.float64, device="cuda") -> DeviceCsr:
    """The reference's --grid3d=W input as CSR (InitGrid3d without self loops, sparse_matrix.h:533-617): W^3 vertices,
    6-neighbour lattice, columns ascending within a row, every value 1."""
    w = width
    a = torch.arange(w, dtype=torch.int64, device=device)
    i = a.repeat_interleave(w * w); j = a.repeat_interleave(w).repeat(w); k = a.repeat(w * w)
    me = i * w * w + j * w + k
    cand = [(i - 1 >= 0, me - w * w), (j - 1 >= 0, me - w), (k - 1 >= 0, me - 1), (k + 1 < w, me + 1), (j + 1 < w, me + w), (i + 1 < w, me + w * w)]
    return _grid_csr(w * w * w, me, cand, dtype, device)


def dense_csr(rows: int, cols: int, dtype=torch.float32, device="cuda", ones: bool = True,
              seed: int = SEED_C2) -> DeviceCsr:
    """The reference's --dense=<cols> matrix (InitDense, sparse_matrix.h:386-413)
    built directly in CSR: every row holds columns 0..cols-1."""
    nnz = rows * cols
    offsets = (torch.arange(rows + 1, dtype=torch.int64, device=device) * cols).to(torch.int32)
    c = torch.arange(cols, dtype=torch.int32, device=device).repeat(rows)
    v = torch.ones(nnz, dtype=dtype, device=device) if ones else uniform_pm1(seed + 1, nnz, dtype, device)
    return DeviceCsr(rows, cols, offsets, c, v)


def degenerate_csr(rows: int = 1 << 24, giant_nnz: int = 1 << 26, every: int = 4096, dtype=torch.float32,
                   device="cuda", ones: bool = True, seed: int = SEED_C4) -> DeviceCsr:
    """C4: square `rows`; row rows/2 holds giant_nnz entries (columns k mod
    cols); every `every`-th other row holds one entry (column = row); all
    remaining rows are empty."""
    cols = rows
    giant = rows // 2
    lens = torch.zeros(rows, dtype=torch.int64, device=device)
    lens[::every] = 1
    lens[giant] = giant_nnz
    offsets64 = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=offsets64[1:])
    nnz = int(offsets64[-1].item())
    col = torch.empty(nnz, dtype=torch.int32, device=device)
    single_rows = torch.arange(0, rows, every, dtype=torch.int64, device=device)
    single_rows = single_rows[single_rows != giant]
    col[offsets64[single_rows]] = single_rows.to(torch.int32)
    g0 = int(offsets64[giant].item())
    # within the giant row the reference's CSR order is sorted by column; with
    # k mod cols and giant_nnz a multiple of cols each column appears
    # giant_nnz/cols times consecutively
    k = torch.arange(giant_nnz, dtype=torch.int64, device=device)
    col[g0:g0 + giant_nnz] = ((k % cols).sort().values).to(torch.int32)
    v = torch.ones(nnz, dtype=dtype, device=device) if ones else uniform_pm1(seed + 1, nnz, dtype, device)
    return DeviceCsr(rows, cols, offsets64.to(torch.int32), col, v)


def rmat_edges(scale: int, edge_begin: int, edge_end: int, device, seed: int,
               a: float = 0.57, b: float = 0.19, c: float = 0.19):
    """(row, col) int64 of R-MAT edges [edge_begin, edge_end): per level one
    uniform draw from counter edge*scale+level picks the quadrant."""
    n = edge_end - edge_begin
    e = torch.arange(edge_begin, edge_end, dtype=torch.int64, device=device)
    row = torch.zeros(n, dtype=torch.int64, device=device)
    col = torch.zeros(n, dtype=torch.int64, device=device)
    for level in range(scale):
        u = uniform01(seed, e * scale + level)
        rbit = (u >= a + b).to(torch.int64)                       # quadrants c, d
        cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)   # quadrants b, d
        row = row * 2 + rbit
        col = col * 2 + cbit
    return row, col


def rmat_row_counts(scale: int, edges: int, device, seed: int = SEED_C5, chunk: int = 1 << 25, edge_begin: int = 0) -> torch.Tensor:
    """int64 [2^scale]: how many of the R-MAT edges [edge_begin, edge_begin + edges) fall in each row (one hashing pass,
    nothing kept)."""
    n = 1 << scale
    counts = torch.zeros(n, dtype=torch.int64, device=device)
    for e0 in range(edge_begin, edge_begin + edges, chunk):
        r, _ = rmat_edges(scale, e0, min(e0 + chunk, edge_begin + edges), device, seed)
        counts += torch.bincount(r, minlength=n)
    return counts


def rmat_csr(scale: int, edges: int, dtype=torch.float64, device="cuda", seed: int = SEED_C5,
             row_lo: int = 0, row_hi: Optional[int] = None, chunk: int = 1 << 25,
             return_edge_ids: bool = False):
    """R-MAT (a,b,c,d = .57,.19,.19,.05), duplicates kept, restricted to rows
    [row_lo, row_hi) (default all): CSR sorted by (row, col, edge id); value of
    edge e = uniform [-1,1) from stream seed+1 counter e.  rows = cols =
    2^scale.  Generating a row range needs a full pass over all edge ids
    (cheap hashing) but only keeps that range's edges, so ranks can build
    disjoint row ranges independently of the GPU count."""
    n = 1 << scale
    row_hi = n if row_hi is None else row_hi
    keep_r, keep_c, keep_e = [], [], []
    for e0 in range(0, edges, chunk):
        e1 = min(e0 + chunk, edges)
        r, c = rmat_edges(scale, e0, e1, device, seed)
        m = (r >= row_lo) & (r < row_hi)
        if row_lo == 0 and row_hi == n:
            keep_r.append(r); keep_c.append(c)
            keep_e.append(torch.arange(e0, e1, dtype=torch.int64, device=device))
        else:
            keep_r.append(r[m]); keep_c.append(c[m])
            keep_e.append(torch.arange(e0, e1, dtype=torch.int64, device=device)[m])
    r = torch.cat(keep_r); c = torch.cat(keep_c); e = torch.cat(keep_e)
    del keep_r, keep_c, keep_e
    # edge ids are already increasing, so a stable sort by (row, col) keeps them ordered
    key = (r - row_lo) * n + c
    order = torch.sort(key, stable=True).indices
    r = r[order]; c = c[order]; e = e[order]
    lens = torch.bincount(r - row_lo, minlength=row_hi - row_lo)
    offsets = torch.zeros(row_hi - row_lo + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=offsets[1:])
    vals = (uniform01(seed + 1, e) * 2.0 - 1.0).to(dtype)
    csr = DeviceCsr(row_hi - row_lo, n, offsets.to(torch.int32) if offsets[-1] < 2**31 else offsets,
                    c.to(torch.int32), vals)
    return (csr, e) if return_edge_ids else csr


# BASELINE config 3 names two SuiteSparse matrices that cannot be fetched offline; these are their sizes
# (ufl_matrices.txt:2379 webbase-1M; SNAP com-Orkut: 117 185 083 stored entries of a symmetric pattern matrix)
C3_WEBBASE_SCALE, C3_WEBBASE_EDGES = 20, 3_105_536
C3_ORKUT_SCALE, C3_ORKUT_EDGES = 22, 117_185_083


def rmat_symmetric_csr(scale: int, edges: int, dtype=torch.float64, device="cuda", seed: int = SEED_C3) -> DeviceCsr:
    """The com-Orkut-sized stand-in of config 3: `edges` R-MAT entries read as the stored half of a SYMMETRIC matrix,
    i.e. mirrored the way InitMarket mirrors a `symmetric` Matrix Market file (sparse_matrix.h:362-368: (c, r) is added
    for every stored (r, c) with r != c).  CSR sorted by (row, col), duplicates kept; the value of a stored entry e is
    uniform [-1, 1) from stream seed + 1 counter e, and its mirror image carries the same value."""
    n = 1 << scale
    r, c = rmat_edges(scale, 0, edges, device, seed)
    e = torch.arange(edges, dtype=torch.int64, device=device)
    off_diag = r != c
    rr = torch.cat([r, c[off_diag]]); cc = torch.cat([c, r[off_diag]]); ee = torch.cat([e, e[off_diag]])
    del r, c, e, off_diag
    order = torch.sort(rr * n + cc, stable=True).indices
    cols = cc[order].to(torch.int32); ee = ee[order]
    lens = torch.bincount(rr, minlength=n)
    del rr, cc, order
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=offsets[1:])
    vals = (uniform01(seed + 1, ee) * 2.0 - 1.0).to(dtype)
    return DeviceCsr(n, n, offsets.to(torch.int32), cols, vals)


# The one matrix the reference publishes a number for (README.md:116,137-138: circuit5M, 5 558 326 x 5 558 326,
# 59 524 291 nonzeros, fp64: 181.6 effective GB/s = 62.96 % of the K40's peak) cannot be fetched offline either.
SEED_CIRCUIT = 0x5EED00C5
CIRCUIT5M_ROWS, CIRCUIT5M_NNZ = 5_558_326, 59_524_291
# its longest rows as fractions of the nonzero count (the largest is circuit5M's 1 290 501-entry row), and where they sit
_CIRCUIT_GIANTS = ((1_290_501 / 59_524_291, 0.2), (645_000 / 59_524_291, 0.4), (320_000 / 59_524_291, 0.6), (160_000 / 59_524_291, 0.8))


def circuit_csr(rows: int = CIRCUIT5M_ROWS, nnz: int = CIRCUIT5M_NNZ, dtype=torch.float64, device="cuda",
                seed: int = SEED_CIRCUIT) -> DeviceCsr:
    """A circuit5M-SHAPED stand-in (a seeded synthetic, not the SuiteSparse matrix): square, exactly `nnz` entries, ~10.7
    per row on average with the spread of a circuit-simulation matrix -- 70 % of the rows hold 2-8 entries, 27.4 % 8-30,
    2.6 % 30-90, and four "supply net" rows hold 2.2 %, 1.1 %, 0.5 % and 0.3 % of all nonzeros -- the diagonal present in
    every row, 80 % of the other entries within +-2000 columns of it (cubic fall-off), 15 % within +-200 000, 3 % anywhere,
    2 % on four hub columns; the giant rows reference columns uniformly.  CSR sorted by (row, column), duplicates kept,
    values uniform in [-1, 1).  Element k of every random stream is a pure function of (seed, k): the same matrix on any device."""
    cols = rows
    r = torch.arange(rows, dtype=torch.int64, device=device)
    u, v = uniform01(seed, r), uniform01(seed + 1, r)
    lens = torch.where(u < 0.70, 2 + (7 * v).to(torch.int64),
                       torch.where(u < 0.9738, 8 + (23 * v).to(torch.int64), 30 + (61 * v).to(torch.int64)))
    giant_rows = [min(int(rows * pos), rows - 1) for _, pos in _CIRCUIT_GIANTS]
    giant_lens = [max(int(round(nnz * frac)), 1) for frac, _ in _CIRCUIT_GIANTS]
    if rows >= 64 and nnz >= 16 * rows // 2:
        for gr, gl in zip(giant_rows, giant_lens):
            lens[gr] = gl
    # the exact count: +-1 on rows spread evenly over the matrix (never a giant row, never below one entry)
    diff = nnz - int(lens.sum().item())
    if diff != 0:
        ok = torch.ones(rows, dtype=torch.bool, device=device)
        ok[torch.tensor(giant_rows, device=device)] = False
        if diff < 0:
            ok &= lens > 1
        cand = torch.nonzero(ok).flatten()
        reps, rest = divmod(abs(diff), int(cand.numel()))
        step = 1 if diff > 0 else -1
        lens[cand] += step * reps
        if rest:
            pick = cand[(torch.arange(rest, dtype=torch.int64, device=device) * int(cand.numel())) // rest]
            lens[pick] += step
        assert int(lens.min().item()) >= 1 or diff > 0
    offsets = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=offsets[1:])
    assert int(offsets[-1].item()) == nnz
    row_of = torch.repeat_interleave(r, lens)
    k = torch.arange(nnz, dtype=torch.int64, device=device)
    first = k == offsets[row_of]                                    # the row's first generated entry: the diagonal
    a, b = uniform01(seed + 2, k), uniform01(seed + 3, k)
    s = 2.0 * b - 1.0
    near = row_of + torch.round(s * s * s * 2000.0).to(torch.int64)
    mid = row_of + torch.round(s * 200_000.0).to(torch.int64)
    anywhere = (b * cols).to(torch.int64)
    hub = (((b * 4).to(torch.int64).clamp_(max=3) * 2 + 1) * cols) // 8
    col = torch.where(a < 0.80, near, torch.where(a < 0.95, mid, torch.where(a < 0.98, anywhere, hub)))
    is_giant = torch.zeros(rows, dtype=torch.bool, device=device)
    if rows >= 64 and nnz >= 16 * rows // 2:
        is_giant[torch.tensor(giant_rows, device=device)] = True
    col = torch.where(is_giant[row_of], anywhere, col)
    col = torch.where(first, row_of, col).clamp_(0, cols - 1)
    del a, b, s, near, mid, anywhere, hub, first
    order = torch.sort(row_of * cols + col).indices                 # (equal keys are equal entries: stability is moot)
    col = col[order].to(torch.int32)
    vals = (uniform01(seed + 4, k[order]) * 2.0 - 1.0).to(dtype)
    return DeviceCsr(rows, cols, offsets.to(torch.int32), col, vals)


def save_csr_image(A, x_seed, path):
    """A DeviceCsr as a raw image (a JSON header, then row offsets, column indices, values) -- what lets a profiled child process load
    BASELINE config 5 instead of generating it (rocprofv3 --pmc dies in the generation of its 2e9 edges): tools/run_config.py --save /
    --load, bench.py's live counter passes.  RAM-backed /dev/shm is the place for the 24 GB."""
    import json, numpy as np
    with open(path, "wb") as f:
        head = json.dumps({"rows": A.rows, "cols": A.cols, "nnz": A.nnz, "dtype": str(A.values.dtype), "x_seed": x_seed}).encode()
        f.write(len(head).to_bytes(8, "little")); f.write(head)
        for t in (A.row_offsets, A.column_indices, A.values):
            for a in range(0, t.numel(), 1 << 28):                      # 1-2 GB at a time through the host
                f.write(t[a:a + (1 << 28)].cpu().numpy().tobytes())


def load_csr_image(path, dev):
    """(DeviceCsr, x_seed) from save_csr_image's file: host reads and host-to-device copies only, no kernels."""
    import json, numpy as np
    with open(path, "rb") as f:
        n = int.from_bytes(f.read(8), "little"); head = json.loads(f.read(n)); base = 8 + n
    tdt = {"torch.float32": torch.float32, "torch.float64": torch.float64}[head["dtype"]]
    ndt = np.float32 if tdt == torch.float32 else np.float64

    def read(count, dtype_np, dtype_t, offset):
        out = torch.empty(count, dtype=dtype_t, device=dev)
        mm = np.memmap(path, dtype=dtype_np, mode="r", offset=offset, shape=(count,))
        for a in range(0, count, 1 << 28):
            out[a:a + (1 << 28)].copy_(torch.from_numpy(np.ascontiguousarray(mm[a:a + (1 << 28)])))
        return out
    off = read(head["rows"] + 1, np.int32, torch.int32, base)
    col = read(head["nnz"], np.int32, torch.int32, base + 4 * (head["rows"] + 1))
    val = read(head["nnz"], ndt, tdt, base + 4 * (head["rows"] + 1) + 4 * head["nnz"])
    return DeviceCsr(head["rows"], head["cols"], off, col, val), head["x_seed"]
