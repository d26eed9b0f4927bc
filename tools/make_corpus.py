#!/usr/bin/env python3
"""tools/make_corpus.py -- a corpus-SHAPED directory of Matrix Market stand-ins for the reference's sweep (eval_csrmv.sh:8-17 runs a
driver over every .mtx of the SuiteSparse collection, README.md:152-160; the collection cannot be fetched here: no network).

    python tools/make_corpus.py --list                               the corpus: index, name, family, nonzero target
    python tools/make_corpus.py --dir DIR --from I --budget-nnz N    write files I, I+1, ... while their nonzeros sum to <= N;
                                                                     prints `next <index>` (= len(corpus) when done)

The corpus spans the axes of the paper's evaluation (Merrill & Garland SC16, Table 1 / Fig. 10 in BASELINE.md): 10 k ... 200 M nonzeros,
row-length coefficient of variation ~0 ... > 100, and the structural families the collection is made of -- 2-D / 3-D stencils, bands,
FEM-like block rows, uniformly random and variable-length rows, R-MAT graphs (general and symmetric), circuit-shaped matrices, Pareto
(power-law) row lengths, wheels / arrows (one row and column touching everything), dense slabs, degenerate (mostly empty + one giant
row).  Every file is seeded, carries the STAND-IN comment line of tools/make_standin_mtx.py right after its banner, and real values in
[-1, 1) (pattern files for the graphs: the reader assigns 1.0, sparse_matrix.h:351-355).  Needs a GPU (the generators run there);
writing is parallel on the host (mspmv_host_write_mtx).  tools/corpus_sweep.sh drives it chunk by chunk so that the disk never holds
more than one chunk."""
import argparse, ctypes, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MARK = "STAND-IN written by tools/make_standin_mtx.py"
SIZES = [10_000, 30_000, 100_000, 300_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000, 200_000_000]
# (family, parameter, largest size index it is generated at)
FAMILIES = [
    ("grid2d", None, 9), ("grid3d", None, 8), ("band", 5, 8), ("band", 27, 7), ("femblock", 3, 7), ("femblock", 6, 8),
    ("uniform", 8, 8), ("uniform", 32, 9), ("uniform", 128, 7), ("varlen", 16, 7),
    ("rmat", 4, 7), ("rmat", 16, 8), ("rmat", 64, 7), ("rmatsym", 16, 8),
    ("circuit", None, 8), ("pareto", 1.3, 7), ("pareto", 1.8, 7), ("pareto", 2.5, 6),
    ("wheel", None, 6), ("arrow", None, 6), ("dense", 5, 7), ("dense", 512, 7), ("degenerate", None, 8),
]


def corpus():
    out = []
    for fam, par, top in FAMILIES:
        for si, nnz in enumerate(SIZES):
            if si <= top:
                tag = f"{nnz // 1000}k" if nnz < 1_000_000 else f"{nnz // 1_000_000}M"
                name = fam + ("" if par is None else f"{par}".replace(".", "p")) + "_" + tag
                out.append((name, fam, par, nnz))
    out.sort(key=lambda t: (t[3], t[0]))            # small files first: a chunk holds many of them
    return out


def build(fam, par, nnz, seed, dev="cuda"):
    """-> (rows, cols, row index, column index, values or None, symmetric)"""
    import torch
    from merge_spmv_amd import generators as G
    f64 = torch.float64

    def coo(A, pattern=False):
        r = torch.repeat_interleave(torch.arange(A.rows, dtype=torch.int64, device=dev), (A.row_offsets[1:] - A.row_offsets[:-1]).long())
        return A.rows, A.cols, r, A.column_indices.long(), None if pattern else A.values.double(), False

    def from_lens(rows, cols, lens, col_of):
        """rows with the given lengths; col_of(row_of, k) -> column of global entry k"""
        lens = lens.clamp_(min=0)
        row_of = torch.repeat_interleave(torch.arange(rows, dtype=torch.int64, device=dev), lens)
        k = torch.arange(int(row_of.numel()), dtype=torch.int64, device=dev)
        c = col_of(row_of, k).clamp_(0, cols - 1)
        v = G.uniform01(seed + 7, k) * 2.0 - 1.0
        return rows, cols, row_of, c, v, False

    if fam == "grid2d":
        return coo(G.grid2d_csr(max(int(math.sqrt(nnz / 4.0)), 3), dtype=f64, device=dev))
    if fam == "grid3d":
        return coo(G.grid3d_csr(max(int(round((nnz / 6.0) ** (1.0 / 3.0))), 3), dtype=f64, device=dev))
    if fam == "band":
        k = int(par); rows = max(nnz // k, k + 1)
        offs = torch.arange(k, dtype=torch.int64, device=dev) - k // 2
        if k > 9:                                    # 27 diagonals: three groups of nine, a 3-D stencil's bandwidth apart
            w = max(int(round(rows ** (1.0 / 3.0))), 4)
            offs = torch.cat([torch.arange(9, device=dev) - 4 + d for d in (-w * w, 0, w * w)]).long()
        lens = torch.full((rows,), k, dtype=torch.int64, device=dev)
        return from_lens(rows, rows, lens, lambda r, j: r + offs[j % k])
    if fam == "femblock":
        b = int(par); nb = 9; per = b * nb           # nb blocks of b x b per block row: b * nb entries per row
        rows = max(nnz // per, per) // b * b
        lens = torch.full((rows,), per, dtype=torch.int64, device=dev)
        w = max(int(math.sqrt(rows / b)), 3)
        boffs = torch.tensor([dy * w + dx for dy in (-1, 0, 1) for dx in (-1, 0, 1)], dtype=torch.int64, device=dev)
        return from_lens(rows, rows, lens, lambda r, j: ((r // b) + boffs[(j % per) // b]) * b + (j % b))
    if fam == "uniform":
        k = int(par); rows = max(nnz // k, 4)
        return coo(G.uniform_csr(rows, rows, k, dtype=f64, device=dev, seed=seed))
    if fam == "varlen":
        k = int(par); rows = max(nnz // k, 4)
        lens = (G.uniform01(seed, torch.arange(rows, dtype=torch.int64, device=dev)) * (2 * k + 1)).long()
        return from_lens(rows, rows, lens, lambda r, j: (G.uniform01(seed + 1, j) * rows).long())
    if fam in ("rmat", "rmatsym"):
        avg = int(par); edges = nnz if fam == "rmat" else nnz // 2
        scale = max(int(round(math.log2(max(nnz // avg, 16)))), 4)
        r, c = G.rmat_edges(scale, 0, edges, dev, seed)
        if fam == "rmatsym":                         # stored lower triangle, `pattern symmetric` like the SNAP graphs
            lo, hi = torch.minimum(r, c), torch.maximum(r, c)
            return 1 << scale, 1 << scale, hi.long(), lo.long(), None, True
        return 1 << scale, 1 << scale, r.long(), c.long(), None, False
    if fam == "circuit":
        return coo(G.circuit_csr(rows=max(int(nnz / 10.7), 64), nnz=nnz, dtype=f64, device=dev, seed=seed))
    if fam == "pareto":
        alpha = float(par); rows = max(nnz // 12, 16)
        u = G.uniform01(seed, torch.arange(rows, dtype=torch.int64, device=dev)).clamp_(min=1e-12)
        lens = (u ** (-1.0 / alpha)).clamp_(max=float(max(nnz // 4, 8)))
        lens = (lens * (nnz / float(lens.sum().item()))).long()
        return from_lens(rows, rows, lens, lambda r, j: (G.uniform01(seed + 1, j) * rows).long())
    if fam == "wheel":                               # InitWheel's shape (sparse_matrix.h): a hub row of all spokes + a rim
        n = max(nnz // 3, 8)
        lens = torch.full((n,), 2, dtype=torch.int64, device=dev); lens[0] = n - 1
        first = lens.cumsum(0) - lens
        def col_of(r, j):
            local = j - first[r]
            return torch.where(r == 0, local + 1, torch.where(local == 0, torch.zeros_like(r), (r % (n - 1)) + 1))
        return from_lens(n, n, lens, col_of)
    if fam == "arrow":                               # diagonal + a dense first row and first column
        n = max(nnz // 3, 8)
        lens = torch.full((n,), 2, dtype=torch.int64, device=dev); lens[0] = n
        first = lens.cumsum(0) - lens
        return from_lens(n, n, lens, lambda r, j: torch.where(r == 0, j - first[r], torch.where(j - first[r] == 0, torch.zeros_like(r), r)))
    if fam == "dense":
        c = int(par); rows = max(nnz // c, 4)
        return coo(G.dense_csr(rows, c, dtype=f64, device=dev, ones=False, seed=seed))
    if fam == "degenerate":                          # BASELINE config 4's shape at every size: 4/5 of the nonzeros in ONE row, mostly empty rows
        rows = max(nnz // 4, 64); every = 64
        giant = max(nnz - rows // every, 8)
        return coo(G.degenerate_csr(rows=rows, giant_nnz=giant, every=every, dtype=f64, device=dev, ones=False, seed=seed))
    raise SystemExit(f"unknown family {fam}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--dir")
    ap.add_argument("--from", dest="first", type=int, default=0)
    ap.add_argument("--budget-nnz", type=float, default=1.5e8)
    ap.add_argument("--max-nnz", type=float, default=2.0e8, help="leave out files above this many nonzeros (a quick pass)")
    ap.add_argument("--device", default="cuda", help="where the generators run (cpu: small files, a functional check)")
    args = ap.parse_args()
    items = [t for t in corpus() if t[3] <= args.max_nnz]
    if args.list:
        for i, (name, fam, par, nnz) in enumerate(items):
            print(i, name, fam, par, nnz)
        print(f"{len(items)} files, {sum(t[3] for t in items) / 1e9:.2f} G nonzeros in all")
        return
    import numpy as np
    import torch
    os.makedirs(args.dir, exist_ok=True)
    H = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
    H.mspmv_host_write_mtx.restype = ctypes.c_int
    H.mspmv_host_write_mtx.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_char_p]
    used, i = 0.0, args.first
    while i < len(items):
        name, fam, par, nnz = items[i]
        if used > 0 and used + nnz > args.budget_nnz:
            break
        t0 = time.time()
        rows, cols, r, c, v, sym = build(fam, par, nnz, 0x5EED0600 + i, args.device)
        rh = r.to(torch.int32).cpu().numpy(); ch = c.to(torch.int32).cpu().numpy()
        vh = None if v is None else np.ascontiguousarray(v.double().cpu().numpy())
        path = os.path.join(args.dir, f"{i:03d}_{name}.mtx")
        st = H.mspmv_host_write_mtx(path.encode(), rows, cols, int(rh.size), rh.ctypes.data, ch.ctypes.data, None if vh is None else vh.ctypes.data,
                                    1 if sym else 0, f"{MARK}: corpus family {fam} {'' if par is None else par} target {nnz} nonzeros; NOT a SuiteSparse matrix".encode())
        if st != 0:
            raise SystemExit(f"{path}: write failed ({st})")
        sys.stderr.write(f"{path}: {rows} x {cols}, {rh.size} entries, {os.path.getsize(path) / 1e6:.0f} MB in {time.time() - t0:.1f} s\n")
        del r, c, v, rh, ch, vh
        if args.device == "cuda":
            torch.cuda.empty_cache()
        used += nnz; i += 1
    print(f"next {i}")


if __name__ == "__main__":
    main()
