/*
 * mspmv_dev.h -- development-only additions of libmspmv_dev.so (the same sources as libmspmv.so built with -DMSPMV_TUNING by
 * `make -C merge_spmv_amd`; the tests that FORCE a code path and the A/B tools load it).  None of this is in the product library,
 * which has no setters and reads nothing from the environment:
 *   - per-host-thread overrides: mspmv_set_tuning, mspmv_set_band_passes, mspmv_set_record_polls, mspmv_set_compact_tiles;
 *   - environment variables read once per process: MSPMV_SMALL_MAX_TILES, MSPMV_COMPACT_MAX_TILES, MSPMV_COMPACT_MAP, MSPMV_LEAN_AVG,
 *     MSPMV_SNAP_MAP (re-tuning aids), MSPMV_FAKE_L2_MIB, MSPMV_FAKE_XCDS, MSPMV_FAKE_INFINITY_CACHE_MIB (tests of the cache policies).
 * Every combination accepted by the setters computes correct results with the product's kernels.
 *
 * A further build, -DMSPMV_DEV (`make -C merge_spmv_amd exp` -> libmspmv_exp.so; tools/trace_tiles.py, the tuning sweeps), adds five
 * more tile shapes per precision and kernels that exist to TIME parts of the tile kernel -- one of them deliberately computes WRONG
 * results.  Its extra mspmv_set_tuning bits:
 *   bit 0        one contiguous tile range per XCD in the persistent form
 *   bits  8..15  persistent form of tile_kernel_vec: resident blocks per CU (grid = that x CUs)
 *   bits 20..23  persistent form: tiles per block
 *   bits 16..18  1 = staging only (wrong results, timing ablation), 6 = per-phase cycle stamps written to the buffer given to
 *                mspmv_dev_set_trace, 7 = the reference's per-thread merge-path walk inside the vectorised tile
 */
#ifndef MSPMV_DEV_H_
#define MSPMV_DEV_H_
#include "mspmv.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Tuning override for experiments (per HOST THREAD: it affects the calls the same thread makes afterwards, including
 * the size queries, and nothing else -- the plans of mspmv_csrmv_plan_* and mspmv_mg_plan_* always run the library
 * defaults; 0 = library default): selects one of the
 * compiled tile shapes for value_bytes and/or the option bits below.  Returns 0, or
 * hipErrorInvalidValue if that shape was not compiled in or a bit is not one of these.  Every
 * combination accepted here computes correct results; kernels that exist only for timing
 * experiments are not part of this library either (the -DMSPMV_DEV build above). */
#define MSPMV_TUNE_NO_XLDS    0x80000 /* never gather a tiny x (<= 4 KB) from LDS */
#define MSPMV_TUNE_ATOMIC_FIX 2   /* single-launch atomicAdd fix-up (non-deterministic) */
#define MSPMV_TUNE_NO_VEC     4   /* force the dword-per-lane kernel with the reference's per-thread path walk (the path taken for unaligned arrays) */
#define MSPMV_TUNE_BINARY_SEARCH 8 /* tile coordinates by a 64-ary wave search per boundary */
#define MSPMV_TUNE_SCATTER_COORDS 0x10000000 /* ... always by one coalesced pass over all row offsets (the default below 10 M rows) */
#define MSPMV_TUNE_INTERP_COORDS  0x20000000 /* ... always by one thread per boundary, interpolation search (the default from 10 M rows up) */
#define MSPMV_TUNE_NO_FUSED   16  /* small problems take the large-problem tile shape too (256x11; fp64 up to 8 M path items: 256x7) */
#define MSPMV_TUNE_TWO_LAUNCH 0x40000000 /* the classic three launches (coordinate pass, tile_kernel_vec with one carry per tile, fix-up) instead of ONE launch of
                                           row-snapped tiles on verified coordinate hints (tile_kernel_snap) */
#define MSPMV_TUNE_FORCE_NT   32  /* CSR streams always read with non-temporal loads */
#define MSPMV_TUNE_FORCE_TEMPORAL 64 /* ... always with ordinary loads (default: by matrix size vs the 256 MB Infinity Cache) */
#define MSPMV_TUNE_NO_LEAN ((int32_t) 0x80000000u) /* one-launch kernel: closed tiles of short rows take the general flag/segmented-scan reduction too (default: the
                                           row-by-row reduction, consume_tile_rows) */
#define MSPMV_TUNE_MULTILEVEL_FIX 128 /* carry fix-up in two/three chunked levels (one launch each) instead of the one-launch owner-computes kernel */
/* bits 24..27: block -> tile mapping of the tile kernel: 0 = default (runs of 64 consecutive tiles per XCD),
 * 15 = plain round-robin, else log2 of the run length. */
int mspmv_set_tuning(int32_t value_bytes, int32_t block_threads,
                     int32_t items_per_thread, int32_t flags);

/* Column-band passes (extension).  A large matrix whose columns are spread uniformly over an x of
 * 1.375-10 x one XCD's L2 (fp32; 1.75-9 x in fp64: 5.5-40 / 7-36 MiB on MI355X) is gather-bound at the Infinity-Cache rate; streaming it 2-4 times, each
 * pass multiplying the nonzeros of one column band (an x slice that stays in every XCD's L2), is 10-29 %
 * faster.  The call stays stateless, asynchronous and three launches: 64 blocks added to the coordinate
 * launch sample 64 windows of 2048 consecutive column indices, and the tile kernel reads their verdicts and
 * runs either its ordinary body or the passes.  Results stay within the strict bound and are bitwise
 * reproducible; rounding differs from the one-sweep result in the last bits (a re-association).
 *   passes = 0  automatic (default): by the sizes of the call (csrc/mspmv_api.hip: band_passes_for) and the verdicts
 *   passes < 0  never
 *   passes >= 2 always that many passes, on any call that takes the 256x11 tile or, in fp64, the 256x7 tile (tests, tuning). */
int mspmv_set_band_passes(int32_t value_bytes, int32_t passes);
/* Clock-scheduled column bands (csrc/mspmv_tdm.hpp): the ONE-pass form of the same organisation, offered to the same calls --
 * a block sorts its tile's nonzeros by column band in LDS and gathers band by band, the band "on air" being read off the
 * chip-wide 100 MHz clock, so every XCD's L2 holds a band or two of x while the CSR stream is read once.  y is bit for bit
 * the one-sweep result (mspmv_set_band_passes(vb, -1)).
 *   policy = 0 the library's rule (default), < 0 never (the passes as before), > 0 always where the passes are offered;
 *   slot_permille: the on-air time of a band in per mille of the computed one (0 = 1000); lookahead_plus_1: bands after the
 *   one on air a block may take, plus one (0 = the default, 2 bands); band_shift: log2 of the columns per band (0 = 18). */
int mspmv_set_tdm(int32_t value_bytes, int32_t policy, int32_t slot_permille, int32_t lookahead_plus_1, int32_t band_shift);
/* Testing aid (per HOST THREAD, like mspmv_set_tuning): how often a tile of the one-launch kernel in which a long row ENDS
 * looks for the partial sum another workgroup publishes before it computes that sum itself from the matrix (0 = the
 * library default, ~0.1 s of polling; 1 = one look; < 0 = never look, which sends every such tile down the recomputing path).  The
 * result is correct for any value: nothing in a call depends on another workgroup making progress; only the time and, by a
 * re-association, the last bits of such a row do. */
int mspmv_set_record_polls(int32_t polls);
/* Testing / tuning aid (per HOST THREAD): up to how many tiles a call of the small tile shape runs the one-launch kernel behind its
 * COMPACT FRONT END (csrc/mspmv_kernels.hpp: compact_front -- small problems, one contiguous tile range per XCD; closed lean tiles on good hints take
 * ~200 instructions per wave of straight-line code at the head of the kernel, every other tile the general body of the same kernel).
 * 0 = the library default (2304 tiles: the sizes at which the matrix stays in the XCDs' L2s between calls), > 0 = that many, < 0 = never.  y is bit for bit the same
 * either way (tests/test_gpu_parity.py: the `compact` / `no_compact` paths).  Matches the reference's special case for small
 * problems (dispatch_spmv_orig.cuh:674-679, agent_spmv_orig.cuh:867-891). */
int mspmv_set_compact_tiles(int32_t max_tiles);
#ifdef MSPMV_DEV
/* -DMSPMV_DEV only: device buffer (16 x 8 uint64 per block) receiving the clock64() stamps; NULL turns it off */
int mspmv_dev_set_trace(void *d_buf);
#endif
#ifdef __cplusplus
}
#endif
#endif
