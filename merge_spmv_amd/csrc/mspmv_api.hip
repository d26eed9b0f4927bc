// mspmv_api.hip -- the C ABI of include/mspmv.h: host dispatch for the
// merge-based CsrMV kernels.  Replaces the reference's DeviceSpmv::CsrMV
// (cub/device/device_spmv.cuh:129-164) and DispatchSpmv::Dispatch
// (cub/device/dispatch/dispatch_spmv_orig.cuh:544-752) -- without that
// dispatcher's per-call device-attribute / occupancy queries and texture
// bind/unbind (all inside the reference's timed loop): grid shapes here are
// pure arithmetic on (rows, nnz), so a call is one to three kernel launches
// and nothing else.  Also here: the prepared-call and SpMM entry points.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <vector>

#include "../../include/mspmv.h"
#include "mspmv_internal.hpp"
#include "mspmv_kernels.hpp"
#include "mspmv_spmm.hpp"

namespace mspmv {

constexpr int SEARCH_BLOCK = 256;
constexpr int INTERP_MIN_ROWS = 10000000;       // coordinate pass: interpolation search from here up, scatter pass below
#ifndef MSPMV_MM_LANE_BLOCK
#define MSPMV_MM_LANE_BLOCK 256
#endif
constexpr int MM_LANE_BLOCK = MSPMV_MM_LANE_BLOCK;      // threads per block of the slot form of SpMM (its tile stays 256 x 7 path items)
constexpr int MM_CHUNK_LOG2 = 6;         // XCD-chunked mapping of the SpMM tiles: 0-3 %
constexpr int FIX_BLOCK = 256;
constexpr int FIX_IPT = 2;              // little serial work per thread: the fix-up is latency-bound (256x8: 14 us, 256x2: 9.5 us, 1024x16: 40 us)
constexpr int FIX_CHUNK = FIX_BLOCK * FIX_IPT;
constexpr int DEV_FLAG_MASK = 1 | 0xff00 | 0x70000 | 0xf00000;   // selectors of the -DMSPMV_DEV kernel variants
constexpr int SMALL_MAX_TILES_DEFAULT = 2304;    // fp32: up to this many 256x7 tiles a problem takes that shape, 256x11 beyond (1024 until the compact front end, 1408 until its
                                                 // launches took one contiguous tile range per XCD: grid2d-900 fp32, 2258 tiles, 7.5 -> 6.7 us)
// The PRODUCT library reads nothing from the environment and has no setters: the re-tuning aids below (an environment variable read
// once, the per-thread overrides of include/mspmv_dev.h) exist only in the -DMSPMV_TUNING build, libmspmv_dev.so.
#ifdef MSPMV_TUNING
static int env_int(const char *name, int fallback) { const char *e = getenv(name); return e ? atoi(e) : fallback; }
#else
static constexpr int env_int(const char *, int fallback) { return fallback; }
#endif
// (dev library: MSPMV_SMALL_MAX_TILES in the environment overrides it, read once: an aid for re-tuning the threshold on other parts)
static int small_max_tiles()
{
    static const int v = [] { const int n = env_int("MSPMV_SMALL_MAX_TILES", 0); return n > 0 ? n : SMALL_MAX_TILES_DEFAULT; }();
    return v;
}
#define SMALL_MAX_TILES (small_max_tiles())
// Closed tiles of the one-launch kernel whose rows average at most this many nonzeros take the lean row-by-row reduction
// (mspmv_kernels.hpp: consume_tile_rows); MSPMV_LEAN_AVG in the environment overrides it (read once: a re-tuning aid), 0 = never
// (8 until round 5.  12: the reference's --dense=9 ... 12 inputs at 16 M nonzeros in fp64 0.0350 / 0.0347 / 0.0340 -> 0.0318 / 0.0317 / 0.0314 ms --
//  rocSPARSE 0.0334 / 0.0329 / 0.0328 --, fp32 0.0240 -> 0.0231, a 1 M-nonzero --dense=10 5.5 -> 3.7 us per call (its tiles reach the compact
//  front end's fast lane), nothing else in the sweep moves; 16: rows of 16 lose -- 0.0331 -> 0.0422; profiles/r05_lean_avg.txt)
constexpr int LEAN_AVG_DEFAULT = 12;
// The compact front end serves problems of up to 2304 tiles of the small shape (both precisions): about the sizes at which the matrix
// stays in the XCDs' L2s from one call to the next -- each XCD works on ONE CONTIGUOUS RANGE of the tiles (compact_tile), the same range in
// every call, 6 (fp64) / 8 (fp32) blocks per CU.  Measured with the limit lifted (tools/ab_driver, profiles/r05_compact_limit.txt): 5-point
// grids in fp64 at 1366 / 1784 / 2258 / 2788 / 4014 tiles 6.1 / 8.4 / 10.8 / 12.8 / 16.3 us against the general kernel's 7.1 / 8.9 / 11.0 /
// 12.7 / 16.8; fp32 at 1784 / 2258 / 2516 / 2788 tiles 5.3 / 6.7 / 7.8 / 9.1 against 5.9 / 7.0 / 7.8 / 9.1; small R-MAT and circuit-shaped
// matrices, most of whose tiles run the general body behind the front end, within +-3 % (tools/compact_other_matrices.py).  With tile =
// block index (the first form, round 5) the limit was one block generation.  MSPMV_COMPACT_MAX_TILES in the environment overrides it (read
// once: a re-tuning aid), 0 = never
static int compact_max_tiles(int value_bytes)
{
    static const int v = env_int("MSPMV_COMPACT_MAX_TILES", -1);
    (void) value_bytes;
    return v >= 0 ? v : 2304;
}
// block -> tile map of the compact launches (mspmv_kernels.hpp: compact_tile): 1 = one contiguous tile range per XCD, 0 = tile = block;
// MSPMV_COMPACT_MAP in the environment overrides it (read once: a re-tuning aid)
static int compact_tile_map()
{
    static const int v = env_int("MSPMV_COMPACT_MAP", 1) != 0;
    return v;
}
static int lean_avg_default()
{
    static const int v = [] { const int n = env_int("MSPMV_LEAN_AVG", -1); return n >= 0 ? n : LEAN_AVG_DEFAULT; }();
    return v;
}

static_assert(TILE_MAP_CONTIGUOUS_CODE == TILE_MAP_CONTIGUOUS, "mapping code shared with the kernels");

// Tile shapes compiled in.  The product library holds the two shapes per precision that pick_shape chooses from -- a sweep
// of every shape over 40 k ... 100 M nonzeros (tools/small_shapes.py, tools/sweep.py; profiles/r03_small_shapes.txt) found
// no size at which another one wins by more than noise --; the further shapes of the tuning sweeps exist only in the
// -DMSPMV_DEV build.
struct Shape { int block, ipt; };
#ifdef MSPMV_DEV
static const Shape kShapesF32[] = {{256, 7}, {256, 11}, {256, 9}, {256, 15}, {256, 5}, {128, 7}, {512, 7}};
static const Shape kShapesF64[] = {{256, 7}, {256, 11}, {256, 5}, {256, 9}, {256, 3}, {128, 5}, {512, 5}};
#elif defined(MSPMV_TUNING)
static const Shape kShapesF32[] = {{256, 7}, {256, 11}};
static const Shape kShapesF64[] = {{256, 7}, {256, 11}};
#endif

// The development override of mspmv_set_tuning / mspmv_set_band_passes (include/mspmv_dev.h, libmspmv_dev.so only): per HOST THREAD,
// read once per call at the C entry points.  [0] = 4-byte values, [1] = 8-byte values.  The product library has no such state:
// every call runs the defaults.
#ifdef MSPMV_TUNING
static thread_local Tune t_tune[2];
static inline const Tune &thread_tune(int value_bytes) { return t_tune[value_bytes == 8]; }
#else
static inline const Tune &thread_tune(int) { static const Tune defaults; return defaults; }
#endif

// Default shape (measured on MI355X with the one-launch kernel; profiles/r03_small_shapes.txt, r03_sweep_vs_rocsparse.txt):
//  * fp32: 256x7 while that cuts the problem into at most SMALL_MAX_TILES (2304; 1024 until round 5) tiles -- more, smaller tiles keep more CUs
//    busy on a small matrix: 4.4-5.6 us per call up to 1.4 M nonzeros where 256x11 takes 4.9-5.9 --, 256x11 beyond (the
//    fastest or within 1 % of the fastest shape on every larger workload tried);
//  * fp64: 256x7 (7 resident blocks per CU instead of 5) up to 8 M path items, 256x11 beyond (24 M until the large shape
//    began to read its hints through the scalar cache: 5-point grids of 10-20 M items 3-6 % faster in 256x11 since);
//  * MSPMV_TUNE_NO_FUSED / _NO_VEC: the large-problem choice whatever the size (tests).
static Shape pick_shape(int value_bytes, long long items, const Tune &t)
{
    if (t.block > 0) return Shape{t.block, t.ipt};
    const bool small_ok = !(t.flags & (MSPMV_TUNE_NO_VEC | MSPMV_TUNE_NO_FUSED));
    if (value_bytes == 8) return items <= 8000000LL ? Shape{256, 7} : Shape{256, 11};
    if (small_ok && (items + 256LL * 7 - 1) / (256LL * 7) <= SMALL_MAX_TILES) return Shape{256, 7};
    return Shape{256, 11};
}

// tile shapes whose vectorised kernel is also compiled with the column-band passes: the large-problem shape, and the fp64
// shape of smaller problems (forced band passes, and what the policy offered while that shape went up to 24 M path items)
static constexpr bool band_shape(int block, int ipt, int value_bytes) { return block == 256 && (ipt == 11 || (value_bytes == 8 && ipt == 7)); }

// room the row-snapped tiles have for the nonzeros they adopt (kernels: snap_head_max)
static constexpr int snap_head_max_host(int block, int ipt)
{
    const int slack = (ipt / 4 + 1) * block * 4 - block * ipt - 16;
    return slack < 192 ? slack : 192;
}

struct Layout {
    Shape shape; int flags;
    int num_tiles;
    int fix_n[3];          // pairs entering fix-up level i (fix_n[0] == num_tiles)
    int fix_levels;
    uint64_t coords_off, carries_off, fix_off[2], pub_off, band_off, band_next_off, rstart_off, err_off, total;
    bool snap;             // ONE launch of tile_kernel_snap when the call allows it (aligned arrays, default block -> tile map, no column-band
                           // passes on offer); the regions of the classic pipeline are laid out all the same
};

// LARGE MATRICES WHOSE ROWS ARE ALL LONG (an average of 240 nonzeros or more: longer than the 192 a tile boundary can snap back over)
// run the classic three launches: with most tile ends inside a row nearly every tile publishes a record and nearly every tile
// takes one, and a record costs its consumer a round trip through the memory side (the records are agent-scope atomics: they pass
// the XCDs' L2s) at the END of the tile's life -- +3...4 us on a block that lives 8-10 us, where a carry per tile and one fix-up
// launch cost 15 us per CALL.  Measured (round 6, tools/long_rows_probe.py, profiles/r06_long_rows_probe.txt; one launch | classic,
// us per SpMV): rows of 512 over a tiny x, 100 M nonzeros, fp64 312 | 209 (rocSPARSE 218), fp32 208 | 129 (172); 30 M: 95 | 67 (69),
// 59 | 42 (48); rows of 2048, 50 M: 186 | 116, 141 | 81.  Crossovers: rows of 384+ from ~8 M path items, rows of 240-384 from ~16 M
// (fp32 rows of 256 at 20 M: 39.1 | 38.4); rows of <= 224 and anything smaller stay with the one launch (rows of 200, 50 M: 178 | 180).
// Judged on what the host knows -- rows and nonzeros -- so a matrix with a FEW long rows among short ones (R-MAT hubs, BASELINE
// config 4's giant row: average 4) keeps the one launch and its records.  Same rule for stateless, prepared and plan-internal calls
// (it is a function of the sizes), so whatever is bitwise equal between them stays so.
static bool long_rows_rule(int rows, int nnz)
{
    const long long items = (long long) rows + nnz;
    if (rows <= 0) return false;
    return ((long long) nnz >= 384LL * rows && items >= 8000000LL) || ((long long) nnz >= 240LL * rows && items >= 16000000LL);
}

static Layout make_layout(int rows, int nnz, int value_bytes, const Tune &tune, bool small_shape = false)
{
    Layout L; memset(&L, 0, sizeof(L));
    const long long total = (long long) rows + nnz;
    L.flags = tune.flags;
    L.shape = small_shape ? Shape{COMPACT_BLOCK, COMPACT_IPT} : pick_shape(value_bytes, total, tune);
    const int tile = L.shape.block * L.shape.ipt;
    L.num_tiles = (int) ((total + tile - 1) / tile);
    const uint64_t pair = value_bytes == 8 ? 16 : 8;
    uint64_t off = 0;
    L.coords_off = off; off = align256(off + uint64_t(L.num_tiles + 1) * sizeof(Coord));
    L.carries_off = off; off = align256(off + uint64_t(L.num_tiles > 0 ? L.num_tiles : 1) * pair);
    // ONE launch (tile_kernel_snap) unless the tuning asks for a piece of the classic pipeline (coordinate pass, tile_kernel_vec or the
    // dword-per-lane tile_kernel, fix-up)
    constexpr int classic = MSPMV_TUNE_TWO_LAUNCH | MSPMV_TUNE_ATOMIC_FIX | MSPMV_TUNE_MULTILEVEL_FIX;
    L.snap = L.num_tiles >= 1 && !(L.flags & (classic | MSPMV_TUNE_NO_VEC | MSPMV_TUNE_BINARY_SEARCH | DEV_FLAG_MASK)) && !long_rows_rule(rows, nnz);
    // published carries of rows longer than the snap limit (16 bytes per tile)
    // (one record per tile + one per group of LB_GROUP tiles: kernels, "GROUP RECORDS")
    L.pub_off = off; off = align256(off + (uint64_t(L.num_tiles > 0 ? L.num_tiles : 1) + uint64_t(L.num_tiles) / LB_GROUP + 1) * 16);
    // column-band passes: the window verdicts and 8 claim counters (always laid out: a buffer sized under one tuning stays
    // large enough under MSPMV_TUNE_NO_FUSED), and one int per tile of the large-problem shape (the chain of tiles each block ran)
    L.band_off = off; off = align256(off + uint64_t(BAND_WINDOWS + 8 * BAND_COUNTER_STRIDE) * sizeof(int));
    L.band_next_off = off; off = align256(off + uint64_t(L.num_tiles > 0 ? L.num_tiles : 1) * sizeof(int));
    // row start of every boundary (4 bytes): with the coordinates, the hints of tile_kernel_snap
    L.rstart_off = off; off = align256(off + uint64_t(L.num_tiles + 1) * sizeof(int));
    L.err_off = off; off = align256(off + 8);           // [0]: receives the call's tag when a bounded poll ran out and the consumer computed the sum itself (diagnostic), [1]: counts such episodes (until round 5: an epoch mixed into the record tags)
    // fix-up levels: n -> 2*ceil(n/CHUNK) until one block suffices
    L.fix_n[0] = L.num_tiles; L.fix_levels = 0;
    if (L.num_tiles > 1) {
        int n = L.num_tiles; int lvl = 0;
        for (;;) {
            const int blocks = (n + FIX_CHUNK - 1) / FIX_CHUNK;
            ++lvl;
            if (blocks == 1 || (L.flags & MSPMV_TUNE_ATOMIC_FIX) || !(L.flags & MSPMV_TUNE_MULTILEVEL_FIX)) break;
            n = 2 * blocks;
            L.fix_n[lvl] = n;
            L.fix_off[lvl - 1] = off; off = align256(off + uint64_t(n) * pair);
        }
        L.fix_levels = lvl;    // <= 3 for rows+nnz < 2^31 (2^31/1280 tiles -> 1640 -> 2 -> done)
    }
    L.total = off > 0 ? off : 256;
    return L;
}

// What the column-band policy is derived from: the L2 a gather can hit in -- one XCD's, as the runtime reports it -- and
// how many of them the device has; queried once per device (never on the hot path again).  Dev library only: MSPMV_FAKE_L2_MIB /
// MSPMV_FAKE_XCDS in the environment override the query (read once: tests of the policy).
struct DeviceCaches { long long l2_bytes; int xcds; };
static DeviceCaches device_caches()
{
    // (lock-free after the first call per device: this sits on the path of every large-problem call.  Two threads racing through
    //  the first call compute the same values.)
    static std::atomic<long long> cached_l2[64];
    static std::atomic<int> cached_xcds[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void) hipGetLastError(); dev = 0; }
    DeviceCaches c; c.l2_bytes = cached_l2[dev].load(std::memory_order_acquire); c.xcds = cached_xcds[dev].load(std::memory_order_relaxed);
    if (c.l2_bytes == 0) {
        c.l2_bytes = 4LL << 20; c.xcds = 8;                                      // MI355X in SPX mode: 8 XCDs x 4 MiB
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && v > 0) c.xcds = v; else (void) hipGetLastError();
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeL2CacheSize, dev) == hipSuccess && v > 0) {
            // (the runtime reports one XCD's L2 on this part; a figure of 16 MiB or more can only be an aggregate)
            c.l2_bytes = v >= (16 << 20) && c.xcds > 1 ? (long long) v / c.xcds : (long long) v;
        } else (void) hipGetLastError();
#ifdef MSPMV_TUNING
        if (const char *e = getenv("MSPMV_FAKE_L2_MIB")) { const double m = atof(e); if (m > 0) c.l2_bytes = (long long) (m * 1048576.0); }
        if (const char *e = getenv("MSPMV_FAKE_XCDS")) { const int n = atoi(e); if (n > 0) c.xcds = n; }
#endif
        cached_xcds[dev].store(c.xcds, std::memory_order_relaxed);
        cached_l2[dev].store(c.l2_bytes, std::memory_order_release);
    }
    return c;
}

// Column-band passes (run_band_passes): how many, for a call the host can only describe by its sizes.  0 = none.
// The choice is a function of x_bytes / (one XCD's L2): a band's slice of x must (nearly) stay in the L2 of every XCD that
// gathers from it, and every further pass costs another CSR stream.  The ratios below were read from ms per SpMV on MI355X
// (L2 = 4 MiB), 96 M uniformly spread nonzeros (tools/band_passes_bench.py; profiles/r02_band_passes.txt, and
// profiles/r04_band_passes_5to8.txt for 5, 6 and 8 bands: every one of them slower than 4 up to x = 24 MiB, 5 bands 1 % ahead
// at 32 MiB fp32 only -- a further pass costs 0.10-0.15 ms, more than the hits it buys), x in MiB:
//   fp32   one pass   2 bands   3 bands   4 bands        fp64   one pass   2 bands   3 bands   4 bands
//    4      0.549      0.622     0.767     0.920            8     0.993      0.849     0.980     1.137
//    6      0.706      0.636     0.774     0.926           12     1.234      0.939     1.020     1.145
//    8      0.933      0.706     0.783     0.934           16     1.362      1.143     1.065     1.170
//   12      1.178      0.868     0.847     0.956           24     1.502      1.379     1.289     1.248
//   16      1.308      1.085     0.939     1.000           32     1.614      1.501     1.462     1.418
//   24      1.457      1.323     1.225     1.170
//   32      1.585      1.445     1.397     1.371
// i.e. fp32: 2 bands from x = 1.375 L2, 3 from 2.625 L2, 4 from 5 L2, none beyond 10 L2 (the slices miss again and the passes
// only cost); fp64: 1.75 / 3.5 / 5 / 9 L2.  Only for the large-problem shape, from a CSR stream of 5 x the device's total L2
// (160 MiB here: a matrix the windows refuse pays ~2-4 us for having been asked, 5 % of a 41 us banded SpMV at 24 M
// nonzeros, 2.5 % from 80 M) and at least 8 nonzeros per row, so that a pass is the CSR stream and little else; whether the
// columns are in fact spread is decided on the device.  Below 256 MB of stream the passes use ordinary loads like the
// one-sweep kernel (the matrix then stays in the Infinity Cache from pass to pass: 24 M uniformly spread nonzeros over
// 7.6 / 11.4 MiB of x: 242 -> 174 us, 301 -> 214 us).
static int band_passes_for(const Layout &L, long long x_bytes, int value_bytes, int rows, int nnz, const CallExtra &ex, int *force)
{
    *force = 0;
    if (ex.no_bands || ex.tile_map != 0 || !band_shape(L.shape.block, L.shape.ipt, value_bytes)) return 0;
    if (L.flags & (MSPMV_TUNE_NO_VEC | DEV_FLAG_MASK)) return 0;
    const int policy = ex.tune.band_passes;
    if (policy < 0) return 0;
    if (policy >= 2) { *force = 1; return x_bytes / value_bytes >= policy ? policy : 0; }
    if ((long long) nnz < 8LL * rows) return 0;
    const DeviceCaches dc = device_caches();
    const unsigned long long stream_bytes = (unsigned long long) nnz * (value_bytes + 4) + 4ull * rows;
    if (stream_bytes < 5ull * (unsigned long long) dc.xcds * (unsigned long long) dc.l2_bytes) return 0;
    const double r = (double) x_bytes / (double) dc.l2_bytes;
    // (The clock-scheduled one-pass form that serves the offered calls since round 6, mspmv_tdm.hpp, would still win beyond 10 / 9 L2 --
    //  48 MiB of fp32 x 1.68 -> 1.16 ms, 64 MiB 1.73 -> 1.53, 48 MiB of fp64 x 1.72 -> 1.57 -- but being a candidate costs every
    //  matrix the windows then refuse the one-launch kernel: a circuit-shaped matrix with 44 MB of x 0.269 -> 0.28 ms.  Not widened.)
    if (value_bytes == 4) return r < 1.375 ? 0 : r < 2.625 ? 2 : r < 5 ? 3 : r <= 10 ? 4 : 0;
    return r < 1.75 ? 0 : r < 3.5 ? 2 : r < 5 ? 3 : r <= 9 ? 4 : 0;
}

// Clock-scheduled column bands (mspmv_tdm.hpp) instead of the passes, for a call band_passes_for offers them to: the band width
// (columns per band = 2^shift: 1 MiB of x, the best of 0.5 / 1 / 2 MiB on C2 in both precisions) -- widened until 32 bands cover x.
// 0 = the passes.  Measured on C2 (tools/tdm_spmv.hip): fp32 0.64 ms against the passes' 0.83, fp64 1.02 against 1.30.
static int tdm_shift_for(long long cols, int value_bytes, int band_passes, const CallExtra &ex)
{
    if (band_passes <= 1 || ex.tune.tdm < 0) return 0;
    int shift = ex.tune.tdm_band_shift > 0 ? ex.tune.tdm_band_shift : value_bytes == 8 ? 17 : 18;      // bands of 1 MiB
    while (shift < TDM_SLOT_SHIFT && ((cols + (1LL << shift) - 1) >> shift) > TDM_MAX_BANDS) ++shift;
    if (((cols + (1LL << shift) - 1) >> shift) > TDM_MAX_BANDS || shift > TDM_SLOT_SHIFT) return 0;
    return shift;
}

// CU count of the current device, queried once per device (never on the hot path again).
static int device_cus()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// ticks per microsecond of the counter s_memrealtime reads (hipDeviceAttributeWallClockRate, kHz; 100 MHz on MI300 / MI355X), queried
// once per device: the slots of the clock-scheduled column bands are lengths of time, not tick counts
static double device_wall_clock_ticks_per_us()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 100.0;
    int khz = cached[dev].load(std::memory_order_relaxed);
    if (khz == 0) {
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void) hipGetLastError(); khz = 100000; }
        cached[dev].store(khz, std::memory_order_relaxed);
    }
    return khz * 1e-3;
}

// opt-in per-kernel event timing (mspmv_profile_begin/_end).  Process-global and meant for ONE
// measuring host thread (mspmv.h says so); the mutex only keeps concurrent callers from corrupting it.
constexpr int PROF_EVENTS = 4;
struct Profiler {
    std::vector<hipEvent_t> events;   // PROF_EVENTS per profiled call
    int capacity = 0, calls = 0;
    bool active = false;
    std::mutex lock;
};
static Profiler g_prof;

static inline int prof_take_slot()
{
    if (!g_prof.active) return -1;
    std::lock_guard<std::mutex> g(g_prof.lock);
    return (g_prof.active && g_prof.calls < g_prof.capacity) ? g_prof.calls++ : -1;
}
static inline void prof_mark(hipStream_t stream, int slot, int which)
{
    if (slot >= 0) (void) hipEventRecord(g_prof.events[size_t(slot) * PROF_EVENTS + which], stream);
}

#define MSPMV_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

// per-call tag of the published carries (LookBack, mspmv_kernels.hpp): a counter from a random start through a
// 64-bit mixer, so that neither a previous call nor whatever another process left in recycled memory can look current
static unsigned long long next_call_tag()
{
    static std::atomic<unsigned long long> counter{[] {
        std::random_device rd;
        return ((unsigned long long) rd() << 32) ^ (unsigned long long) rd() ^ (unsigned long long) (uintptr_t) &rd;
    }()};
    unsigned long long z = counter.fetch_add(0x9E3779B97F4A7C15ull, std::memory_order_relaxed) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static hipError_t after_launch(hipStream_t stream, int debug_sync, const char *name, unsigned grid, unsigned block, int *d_error = nullptr, unsigned error_tag = 0)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (debug_sync) {
        printf("mspmv: %s<<<%u, %u>>>\n", name, grid, block); fflush(stdout);
        e = hipStreamSynchronize(stream);
        if (e == hipSuccess && d_error) {
            // word 0 of the error region holds this call's tag when a block gave up waiting for another block's record and
            // computed the sum from the matrix instead (temp storage is not initialised, so any other value means nothing):
            // the result is complete and correct, the call was slow -- worth a line in a debug run, not an error
            unsigned h = 0;
            e = hipMemcpy(&h, d_error, sizeof(h), hipMemcpyDeviceToHost);
            if (e == hipSuccess && error_tag != 0 && h == error_tag)
                fprintf(stderr, "mspmv: %s: a bounded wait between workgroups ran out; the sum was recomputed from the matrix (fewer resident workgroups than assumed?)\n", name);
        }
    }
    return e;
}

// Resident blocks of a kernel on the current device (blocks per CU from the occupancy calculator x CUs), computed once per
// kernel instantiation and device: what bounds the waits between workgroups of the one-launch kernels.
template <typename K>
static int resident_blocks(K kernel, int block, std::atomic<int> (&cache)[64])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) { (void) hipGetLastError(); n = 2; }
        v = std::min(n, 2048 / block) * device_cus();
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
// Largest run length 2^k (<= the wanted one) of the XCD-chunked block -> tile mapping under which a tile that waits for a
// lower-numbered TILE's record always finds that tile's block dispatched: inside a group of 8 * 2^k blocks a predecessor
// can sit on a block up to 8 * 2^k - 1 later, and workgroups are dispatched in order, so one group must fit the resident
// blocks -- twice over, to leave room for whatever else shares the device.
static int safe_chunk_log2(int wanted_log2, int resident)
{
    int k = wanted_log2;
    while (k > 0 && 16LL * (1LL << k) > resident) --k;
    return k;
}

#ifdef MSPMV_DEV
// ---- development variants of the tile kernel (libmspmv_dev.so only; `make -C merge_spmv_amd dev`) ----
// tuning bits, all rejected by the product library:
//   bit 0        one contiguous tile range per XCD, persistent form (the old XCD_REMAP experiment)
//   bits  8..15  persistent form: resident blocks per CU (grid = that x CUs)
//   bits 20..23  persistent form: tiles per block
//   bits 16..18  1 = staging only (WRONG results, timing ablation), 6 = per-phase cycle stamps written to the
//                buffer given to mspmv_dev_set_trace, 7 = the reference's per-thread merge-path walk in the tile
constexpr int MSPMV_DEV_FLAG_BITS = DEV_FLAG_MASK;
template <typename V, int BLOCK, int IPT>
static bool launch_dev_variant(const Layout &L, const Params<V> &p, bool axpby, bool nt, const Coord *coords, Carry<V> *carries,
                               int chunk_log2, hipStream_t stream)
{
    const int forced = (L.flags >> 8) & 0xff;
    const int tpb_flag = (L.flags >> 20) & 0xf;
    const int tpb = tpb_flag ? tpb_flag : 1;
    const bool remap = !axpby && (L.flags & 1) != 0;
    const int ablate = axpby ? 0 : (L.flags >> 16) & 7;
    if (!(tpb_flag || forced || remap || ablate)) return false;
#define MSPMV_LAUNCH_P(...)                                                                                        \
    do {                                                                                                   \
        auto kernel = tile_kernel_vec<V, BLOCK, IPT, __VA_ARGS__>;                                          \
        static std::atomic<int> resident{0};      /* one per kernel variant */                              \
        int per_cu = forced ? forced : resident.load(std::memory_order_relaxed);                            \
        if (per_cu == 0) {                                                                                 \
            int n = 0;                                                                                     \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, BLOCK, 0) != hipSuccess || n < 1) n = 4; \
            per_cu = std::min(n, 2048 / BLOCK);                                                            \
            resident.store(per_cu, std::memory_order_relaxed);                                             \
        }                                                                                                  \
        long long want = (long long) per_cu * device_cus();                                                \
        if (!forced) want = std::max<long long>(want, (L.num_tiles + tpb - 1) / tpb);                      \
        const unsigned pgrid = (unsigned) std::min<long long>(L.num_tiles, want);                          \
        hipLaunchKernelGGL(kernel, dim3(pgrid), dim3(BLOCK), (size_t) p.x_lds * sizeof(V), stream, p, coords, carries, L.num_tiles, chunk_log2, BandArgs{nullptr, nullptr, nullptr, 0, 0, 0, 0, TdmArgs{0, 0, 0.f, 0}});   \
    } while (0)
    if (ablate == 1) MSPMV_LAUNCH_P(false, false, true, 1, true);
    else if (ablate == 6) MSPMV_LAUNCH_P(false, false, true, 6, true);
    else if (ablate == 7) MSPMV_LAUNCH_P(false, false, true, 7, true);
    else if (remap) { if (nt) MSPMV_LAUNCH_P(false, true, true, 0, true); else MSPMV_LAUNCH_P(false, true, false, 0, true); }
    else if (axpby) { if (nt) MSPMV_LAUNCH_P(true, false, true, 0, true); else MSPMV_LAUNCH_P(true, false, false, 0, true); }
    else if (nt) MSPMV_LAUNCH_P(false, false, true, 0, true);
    else MSPMV_LAUNCH_P(false, false, false, 0, true);
#undef MSPMV_LAUNCH_P
    return true;
}
#endif

template <typename V, int BLOCK, int IPT>
static hipError_t run_shape(const Layout &L, void *d_temp, const Params<V> &p, bool axpby, hipStream_t stream,
                            int debug_sync, const CallExtra &ex)
{
    char *base = static_cast<char *>(d_temp);
    Coord *coords = reinterpret_cast<Coord *>(base + L.coords_off);
    Carry<V> *carries = reinterpret_cast<Carry<V> *>(base + L.carries_off);
    const int tile_items = BLOCK * IPT;
    int phase = ex.phase;

    // 16-byte streaming needs 16-byte aligned array bases (hipMalloc gives 256) and at
    // least one full 4-element chunk in each array
    const bool vec = !(L.flags & MSPMV_TUNE_NO_VEC) && p.nnz >= 4 && p.rows >= 3 &&
                     ((reinterpret_cast<uintptr_t>(p.values) | reinterpret_cast<uintptr_t>(p.cols) |
                       reinterpret_cast<uintptr_t>(p.row_end - 1)) & 15) == 0;
    // a profiler slot is taken only by calls that run all the passes it brackets
    const int slot = phase == PHASE_COORDS_ONLY ? -1 : prof_take_slot();
    // (the band-major plan's contiguous tile ranges, ex.tile_map: through the one-launch kernel in fp64 -- C2 plan 0.669 -> 0.640 ms --,
    //  through the classic launches in fp32, where the one-launch form measured 3 % slower: 0.508 -> 0.523)
    if (L.snap && vec && (ex.tile_map == 0 || sizeof(V) == 8) && ex.band_passes <= 1) {
        // ---- ONE launch: row-snapped tiles on verified coordinate hints (tile_kernel_snap) ----
        int *rstart = reinterpret_cast<int *>(base + L.rstart_off);
        if (phase == PHASE_COORDS_ONLY) {
            // mspmv_csrmv_prepare: fill in the hints (what the first call would otherwise find tile by tile)
            BoundaryOut bo; bo.coords = coords; bo.rstart = rstart;
            const bool interp = (L.flags & MSPMV_TUNE_INTERP_COORDS) || (!(L.flags & MSPMV_TUNE_SCATTER_COORDS) && p.rows >= INTERP_MIN_ROWS);
            const unsigned cgrid = interp ? (unsigned) ((L.num_tiles + 1 + SEARCH_BLOCK - 1) / SEARCH_BLOCK)
                                          : (unsigned) ((((long long) p.rows + 1 + 3) / 4 + SEARCH_BLOCK - 1) / SEARCH_BLOCK);
            if (interp) hipLaunchKernelGGL((coords_interp_kernel<SEARCH_BLOCK>), dim3(cgrid), dim3(SEARCH_BLOCK), 0, stream, p.row_end, p.rows, p.nnz, tile_items, L.num_tiles, bo);
            else hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, BLOCK * IPT, true>), dim3(cgrid), dim3(SEARCH_BLOCK), 0, stream, p.row_end - 1, p.rows, p.nnz, L.num_tiles, bo, BandDetectArgs{});
            return after_launch(stream, debug_sync, interp ? "coords_interp_kernel" : "coords_scatter_kernel", cgrid, SEARCH_BLOCK);
        }
        prof_mark(stream, slot, 0);
        prof_mark(stream, slot, 1);
        const unsigned long long tag = next_call_tag();
        LookBack lb; lb.rec = reinterpret_cast<unsigned long long *>(base + L.pub_off);
        lb.tag_a = (unsigned) (tag >> 32) | 1u; lb.tag_b = (unsigned) tag; lb.error = reinterpret_cast<int *>(base + L.err_off);
        lb.group_base = L.num_tiles;
        lb.call_tag = lb.tag_a; lb.max_polls = ex.tune.record_polls > 0 ? ex.tune.record_polls : ex.tune.record_polls < 0 ? 0 : REC_MAX_POLLS;
        const unsigned long long stream_bytes = (unsigned long long) p.nnz * (sizeof(V) + 4) + 4ull * p.rows;
        const bool nt = (L.flags & MSPMV_TUNE_FORCE_NT) || (!(L.flags & MSPMV_TUNE_FORCE_TEMPORAL) && stream_bytes > (256ull << 20));
        const unsigned grid = (unsigned) L.num_tiles;
        const size_t xl = (size_t) p.x_lds * sizeof(V);
        const int lean_avg = (L.flags & MSPMV_TUNE_NO_LEAN) ? 0 : lean_avg_default();
        // (every launch below is ONE call into the HIP runtime -- launch_exact: hipLaunchKernel with its status -- and the compact
        //  variant needs nothing else from it: the reference's timing loop is bound by the enqueueing thread for small problems)
        hipError_t launched = hipSuccess;
        // small problems (compact_max_tiles): the same kernel behind its compact front end (kernels: compact_front) -- bit for bit the same y
        bool compact = false;
        if constexpr (BLOCK == COMPACT_BLOCK && IPT == COMPACT_IPT) {
            compact = !nt && ex.tile_map == 0 && L.num_tiles > 1 && L.num_tiles <= (ex.tune.compact_tiles > 0 ? ex.tune.compact_tiles : ex.tune.compact_tiles < 0 ? 0 : compact_max_tiles((int) sizeof(V))) &&
                      (unsigned long long) ex.num_cols * sizeof(V) < (1ull << 32) && (unsigned long long) p.nnz * sizeof(V) < (1ull << 32) &&
                      (unsigned long long) p.rows * 4ull < (1ull << 32) - 8;      // (32-bit byte offsets in the fast lane: every array < 4 GB)
            if (compact) {
                // (a tiny x is gathered from memory here, not from an LDS copy: the copy pays on matrices that stream from HBM, a problem of
                //  this size has x in its caches anyway -- 3.5 -> 2.8 us per call on a 900-row grid -- and the result is the same
                //  bit for bit, tests/test_gpu_parity.py::test_tiny_x_is_gathered_from_lds)
                Params<V> pc = p; pc.x_lds = 0;
                launched = launch_snap_compact<V>(axpby, grid, 0, stream, coords, rstart, L.num_tiles, pc, carries, lb, lean_avg, compact_tile_map());
            }
        }
        if (!compact) {
            static std::atomic<int> snap_cache[64];
            const int chunk_flag = (L.flags >> 24) & 0xf;
            const int wanted = chunk_flag == 0 ? 6 : chunk_flag == 15 ? 0 : chunk_flag;
            // (ex.tile_map: the band-major plan's one contiguous tile range per XCD -- inside a range a lower-numbered tile sits on an
            //  earlier block of the same XCD, so a tile that waits for records waits for blocks dispatched before it -- EXCEPT the first
            //  tiles of XCD k's range, which wait for the LAST tiles of XCD k - 1's range: few waiters, a bounded poll, then the sum
            //  recomputed from the matrix; correct, and slow only for a row longer than HEAD_MAX that crosses a range boundary)
            // (dev library: MSPMV_SNAP_MAP in the environment, read once: 30 = one contiguous tile range per XCD for every one-launch call, 0 .. 8 = that run length)
            static const int env_map = env_int("MSPMV_SNAP_MAP", -1);
            const int chunk_log2 = ex.tile_map ? ex.tile_map : env_map == TILE_MAP_CONTIGUOUS_CODE ? env_map
                                 : safe_chunk_log2(env_map >= 0 && env_map <= 8 ? env_map : wanted, resident_blocks(tile_kernel_snap<V, BLOCK, IPT, true, true>, BLOCK, snap_cache));
#define MSPMV_LAUNCH_SNAP(AX, NTF) launched = launch_exact(tile_kernel_snap<V, BLOCK, IPT, AX, NTF>, dim3(grid), dim3(BLOCK), xl, stream, coords, rstart, L.num_tiles, chunk_log2, p, carries, lb, lean_avg)
            if (axpby) { if (nt) MSPMV_LAUNCH_SNAP(true, true); else MSPMV_LAUNCH_SNAP(true, false); }
            else if (nt) MSPMV_LAUNCH_SNAP(false, true);
            else MSPMV_LAUNCH_SNAP(false, false);
#undef MSPMV_LAUNCH_SNAP
        }
        MSPMV_CHECK(launched);
        if (debug_sync) MSPMV_CHECK(after_launch(stream, debug_sync, compact ? "tile_kernel_snap (compact front end)" : "tile_kernel_snap", grid, BLOCK, lb.error, lb.tag_a));
        prof_mark(stream, slot, 2);
        prof_mark(stream, slot, 3);
        return hipSuccess;
    } else {
    // column-band passes (band_passes_for): 64 sampled windows of column indices decide, on the device, whether the
    // tile kernel (its BAND variant) runs its ordinary body or the passes
    bool band = false, band_sampled = false;
    unsigned band_grid = 0;
    int band_resident_per_cu = 4;
    if constexpr (band_shape(BLOCK, IPT, (int) sizeof(V))) {
        if (vec && ex.band_passes > 1 && phase != PHASE_COORDS_ONLY) {
            // the passes are run by 4 (fp64: 5) blocks per CU, or as many as are resident at once if that is fewer: the gathers
            // of a pass hit L2, so the passes are not short of waves in flight, and with all 8 slots of a CU taken the
            // ~35 000 blocks of the launch that only return would queue up behind the work instead of draining beside it
            // (C2 fp32, blocks per CU 8 / 6 / 4 / 3: 0.857 / 0.842 / 0.832 / 0.843 ms; fp64, 5 / 4.4 / 3.1 / 2.5: 1.297 / 1.296 / 1.36 / 1.50)
            static std::atomic<int> resident{0};
            int per_cu = resident.load(std::memory_order_relaxed);
            if (per_cu == 0) {
                int n = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, tile_kernel_vec<V, BLOCK, IPT, true, false, true, 0, false, true>, BLOCK, 0) != hipSuccess || n < 1) { (void) hipGetLastError(); n = 4; }
                per_cu = std::min(n, 2048 / BLOCK);
                resident.store(per_cu, std::memory_order_relaxed);
            }
            band_resident_per_cu = per_cu;
            const int band_per_cu = std::min(per_cu, sizeof(V) == 4 ? 4 : 5);
            long long want = std::min<long long>(L.num_tiles, (long long) band_per_cu * device_cus());
            if (want >= 8) want &= ~7LL;                                   // (8 interleaved tile sequences, one per XCD)
            band_grid = (unsigned) want;
            band = want >= 8 || want == L.num_tiles;
        }
    }
    int *band_verdict = reinterpret_cast<int *>(base + L.band_off);
    BandDetectArgs da; da.cols = p.cols; da.nnz = p.nnz; da.num_cols = ex.num_cols; da.line_shift = sizeof(V) == 4 ? 5 : 4;
    da.verdict = band_verdict;
    // 1. tile boundary coordinates (also the row start of every boundary: a later prepared call may run tile_kernel_snap on them)
    BoundaryOut bo; bo.coords = coords; bo.rstart = reinterpret_cast<int *>(base + L.rstart_off);
    prof_mark(stream, slot, 0);
    if (phase == PHASE_SKIP_COORDS) {
        // already in d_temp
    } else if (L.flags & MSPMV_TUNE_BINARY_SEARCH) {
        const unsigned grid = (unsigned) ((L.num_tiles + 1 + (SEARCH_BLOCK / WAVE) - 1) / (SEARCH_BLOCK / WAVE));
        hipLaunchKernelGGL((search_kernel<SEARCH_BLOCK>), dim3(grid), dim3(SEARCH_BLOCK), 0, stream, p.row_end, p.rows,
                           p.nnz, tile_items, L.num_tiles, bo);
        MSPMV_CHECK(after_launch(stream, debug_sync, "search_kernel", grid, SEARCH_BLOCK));
    } else if ((L.flags & MSPMV_TUNE_INTERP_COORDS) || (!(L.flags & MSPMV_TUNE_SCATTER_COORDS) && p.rows >= INTERP_MIN_ROWS)) {
        // from 10 M rows up: one thread per boundary, interpolation search (latency-bound: <= 17 us whatever the row
        // count, 2-8 us on regular matrices) instead of reading all of row_offsets (20-23 us at 16.8 M rows)
        const unsigned grid = (unsigned) ((L.num_tiles + 1 + SEARCH_BLOCK - 1) / SEARCH_BLOCK);
        hipLaunchKernelGGL((coords_interp_kernel<SEARCH_BLOCK>), dim3(grid), dim3(SEARCH_BLOCK), 0, stream, p.row_end, p.rows, p.nnz,
                           tile_items, L.num_tiles, bo);
        MSPMV_CHECK(after_launch(stream, debug_sync, "coords_interp_kernel", grid, SEARCH_BLOCK));
    } else {
        const long long threads = ((long long) p.rows + 1 + 3) / 4;       // 4 row indices per thread
        const unsigned grid = (unsigned) ((threads + SEARCH_BLOCK - 1) / SEARCH_BLOCK);
        const int *row_offsets = p.row_end - 1;
        const bool aligned = (reinterpret_cast<uintptr_t>(row_offsets) & 15) == 0;
        if (band) {
            // + BAND_WINDOWS blocks that sample the column windows: no launch of their own
            const unsigned dgrid = grid + BAND_WINDOWS;
            if (aligned) hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, BLOCK * IPT, true, true>), dim3(dgrid), dim3(SEARCH_BLOCK), 0,
                                            stream, row_offsets, p.rows, p.nnz, L.num_tiles, bo, da);
            else hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, BLOCK * IPT, false, true>), dim3(dgrid), dim3(SEARCH_BLOCK), 0,
                                    stream, row_offsets, p.rows, p.nnz, L.num_tiles, bo, da);
            band_sampled = true;
        } else if (aligned)
            hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, BLOCK * IPT, true>), dim3(grid), dim3(SEARCH_BLOCK), 0,
                               stream, row_offsets, p.rows, p.nnz, L.num_tiles, bo, da);
        else
            hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, BLOCK * IPT, false>), dim3(grid), dim3(SEARCH_BLOCK), 0,
                               stream, row_offsets, p.rows, p.nnz, L.num_tiles, bo, da);
        MSPMV_CHECK(after_launch(stream, debug_sync, "coords_scatter_kernel", band ? grid + BAND_WINDOWS : grid, SEARCH_BLOCK));
    }
    if (phase == PHASE_COORDS_ONLY) return hipSuccess;
    if (band && !band_sampled) {             // (prepared calls, the other coordinate passes)
        hipLaunchKernelGGL((band_detect_kernel<SEARCH_BLOCK>), dim3(BAND_WINDOWS), dim3(SEARCH_BLOCK), 0, stream, da);
        MSPMV_CHECK(after_launch(stream, debug_sync, "band_detect_kernel", BAND_WINDOWS, SEARCH_BLOCK));
    }
    // 2. tiles
    prof_mark(stream, slot, 1);
    {
        const unsigned grid = (unsigned) L.num_tiles;
        if (vec) {
            // One tile per block: the hardware's block scheduler balances the load and de-phases the
            // blocks of a CU.  (A persistent, software-prefetching form of the same kernel exists in
            // the -DMSPMV_DEV build; with the flag/segmented-scan reduction it measured 7-10 % slower
            // on streaming matrices)
            // XCD-chunked block -> tile mapping: runs of 2^6 consecutive tiles per XCD.  Measured against
            // plain round-robin: grid2d -19 %, C4 -7 %, dense32 -5 %, band5 -3 %, nothing slower (16 and
            // 256 are within 1-2 % of 64).  Tuning bits 24-27: 0 = default, 15 = off, else log2 of the
            // run length.  Prepared band-major plans ask for one contiguous tile range per XCD instead.
            const int chunk_flag = (L.flags >> 24) & 0xf;
            const int chunk_log2 = ex.tile_map ? ex.tile_map : chunk_flag == 0 ? 6 : chunk_flag == 15 ? 0 : chunk_flag;
            // CSR streams: ordinary loads while the matrix fits the 256 MB Infinity Cache (it then stays there
            // between the SpMVs of a solver: 214 MB dense5 fp64 0.039 vs 0.053 ms, 208 MB grid2d 0.0438 vs
            // 0.0447), non-temporal loads beyond (they keep x in L2: C2 -2 %, band5 -7 %, C4 -14 %, dense32 -7 %;
            // only the 0.7-1.1 GB grids prefer ordinary loads, by 2-5 %; profiles/r02_stream_policy.txt)
            const unsigned long long stream_bytes = (unsigned long long) p.nnz * (sizeof(V) + 4) + 4ull * p.rows;
            const bool nt = (L.flags & MSPMV_TUNE_FORCE_NT) || (!(L.flags & MSPMV_TUNE_FORCE_TEMPORAL) && stream_bytes > (256ull << 20));
            bool launched = false;
#ifdef MSPMV_DEV
            launched = launch_dev_variant<V, BLOCK, IPT>(L, p, axpby, nt, coords, carries, chunk_log2, stream);
#endif
            BandArgs ba; ba.verdict = nullptr; ba.counters = nullptr; ba.next = nullptr; ba.grid = 0; ba.bands = 0; ba.band_cols = 0; ba.force = 0; ba.tdm = TdmArgs{0, 0, 0.f, 0};
            if constexpr (band_shape(BLOCK, IPT, (int) sizeof(V))) {
                if (band && !launched) {
                    // the BAND variant: the same kernel, whose first band_grid blocks run the column-band passes instead
                    // when the verdicts (or mspmv_set_band_passes) say so
                    ba.verdict = band_verdict; ba.counters = band_verdict + BAND_WINDOWS; ba.grid = (int) band_grid;
                    ba.next = reinterpret_cast<int *>(base + L.band_next_off);
                    ba.bands = ex.band_passes; ba.band_cols = ex.band_cols; ba.force = ex.band_force;
                    if (ex.tdm_shift > 0) {
                        // clock-scheduled column bands instead of the passes (mspmv_tdm.hpp): every block of the launch stages its one tile
                        // band by band.  Blocks resident per CU: what the kernel's occupancy says (LDS, registers).
                        const int per_cu = std::max(1, (int) std::min<long long>((L.num_tiles + device_cus() - 1) / device_cus(), band_resident_per_cu));
                        ba.tdm.band_shift = ex.tdm_shift; ba.tdm.bands = ex.tdm_bands;
                        ba.tdm.lookahead = ex.tune.tdm_lookahead > 0 ? ex.tune.tdm_lookahead - 1 : std::max(1, ex.tdm_bands / 8);      // (an eighth of x ahead of the clock: 1 of 12 bands, 3 of 24)
                        // A band stays on air for as long as the resident blocks need for their gathers of it at the L2 gather rate
                        // (1.02 G gathers/s per CU: 262 G/s over 256 CUs, profiles/r02_hw_ceilings.txt) and a sixth more -- or, if
                        // that is longer, for as long as every XCD needs to fetch the band over the fabric (7.8 TB/s for all of them):
                        // C2 fp32 2.13 us (12 bands of 1 MiB, 8 blocks per CU), fp64 1.08 us (24 bands, 5 blocks per CU); both
                        // constants read off sweeps of the slot length (tools/tdm_check.py sweep: the minimum is sharp, +-10 % of the
                        // slot cost 3-5 % -- a block that misses a band's slot waits for the next rotation).  In ticks of the wall clock (10 ns here).
                        // (a tile's BLOCK * IPT path items are nonzeros and row ends: rows of 8 leave 2503 gathers per tile, and the optimum moves with them)
                        const double nz_share = (double) p.nnz / ((double) p.nnz + (double) p.rows);
                        const double gather_us = 1.16 * nz_share * (double) per_cu * BLOCK * IPT / ex.tdm_bands / 1.02e3;
                        const double fabric_us = (double) sizeof(V) * (double) (1u << ex.tdm_shift) * device_caches().xcds / 7.8e6;
                        const double ticks = std::max(8.0, std::max(gather_us, fabric_us) * device_wall_clock_ticks_per_us() * (ex.tune.tdm_slot_permille > 0 ? ex.tune.tdm_slot_permille * 1e-3 : 1.0));
                        ba.tdm.inv_slot = (float) (1.0 / ticks);
                    }
#define MSPMV_LAUNCH_BAND(AX, NTF, TD) hipLaunchKernelGGL((tile_kernel_vec<V, BLOCK, IPT, AX, false, NTF, 0, false, true, TD>), dim3(grid), dim3(BLOCK), (size_t) p.x_lds * sizeof(V), stream, p, coords, carries, L.num_tiles, chunk_log2, ba)
                    if (ex.tdm_shift > 0) {
                        if (axpby) { if (nt) MSPMV_LAUNCH_BAND(true, true, true); else MSPMV_LAUNCH_BAND(true, false, true); }
                        else if (nt) MSPMV_LAUNCH_BAND(false, true, true);
                        else MSPMV_LAUNCH_BAND(false, false, true);
                    } else {
                        if (axpby) { if (nt) MSPMV_LAUNCH_BAND(true, true, false); else MSPMV_LAUNCH_BAND(true, false, false); }
                        else if (nt) MSPMV_LAUNCH_BAND(false, true, false);
                        else MSPMV_LAUNCH_BAND(false, false, false);
                    }
#undef MSPMV_LAUNCH_BAND
                    launched = true;
                }
            }
#define MSPMV_LAUNCH(AX, NTF) hipLaunchKernelGGL((tile_kernel_vec<V, BLOCK, IPT, AX, false, NTF, 0, false>), dim3(grid), dim3(BLOCK), (size_t) p.x_lds * sizeof(V), stream, p, coords, carries, L.num_tiles, chunk_log2, ba)
            if (!launched) {
                if (axpby) { if (nt) MSPMV_LAUNCH(true, true); else MSPMV_LAUNCH(true, false); }
                else if (nt) MSPMV_LAUNCH(false, true);
                else MSPMV_LAUNCH(false, false);
            }
#undef MSPMV_LAUNCH
        } else {
            if (axpby) hipLaunchKernelGGL((tile_kernel<V, BLOCK, IPT, true>), dim3(grid), dim3(BLOCK), 0, stream, p, coords, carries, L.num_tiles);
            else       hipLaunchKernelGGL((tile_kernel<V, BLOCK, IPT, false>), dim3(grid), dim3(BLOCK), 0, stream, p, coords, carries, L.num_tiles);
        }
        MSPMV_CHECK(after_launch(stream, debug_sync, vec ? "tile_kernel_vec" : "tile_kernel", grid, BLOCK));
    }
    }
    // 3. carry fix-up (not needed for a single tile: its carry is the (rows, 0) pair; nor when the self-searching
    //    tiles of a small problem have added the carries themselves)
    prof_mark(stream, slot, 2);
    if (L.num_tiles > 1) {
        if (L.flags & MSPMV_TUNE_ATOMIC_FIX) {
            const unsigned grid = (unsigned) ((L.num_tiles + FIX_BLOCK - 1) / FIX_BLOCK);
            hipLaunchKernelGGL((fixup_atomic_kernel<V, FIX_BLOCK>), dim3(grid), dim3(FIX_BLOCK), 0, stream, carries,
                               L.num_tiles, p.y, p.rows, p.alpha);
            MSPMV_CHECK(after_launch(stream, debug_sync, "fixup_atomic_kernel", grid, FIX_BLOCK));
        } else if (!(L.flags & MSPMV_TUNE_MULTILEVEL_FIX)) {
            const unsigned grid = (unsigned) ((L.num_tiles + FIX_CHUNK - 1) / FIX_CHUNK);
            hipLaunchKernelGGL((fixup_onepass_kernel<V, FIX_BLOCK, FIX_IPT>), dim3(grid), dim3(FIX_BLOCK), 0, stream, carries,
                               L.num_tiles, p.y, p.rows, p.alpha);
            MSPMV_CHECK(after_launch(stream, debug_sync, "fixup_onepass_kernel", grid, FIX_BLOCK));
        } else {
            const Carry<V> *in = carries;
            for (int lvl = 0; lvl < L.fix_levels; ++lvl) {
                const int n = L.fix_n[lvl];
                const unsigned grid = (unsigned) ((n + FIX_CHUNK - 1) / FIX_CHUNK);
                Carry<V> *out = grid > 1 ? reinterpret_cast<Carry<V> *>(base + L.fix_off[lvl]) : nullptr;
                hipLaunchKernelGGL((fixup_kernel<V, FIX_BLOCK, FIX_IPT>), dim3(grid), dim3(FIX_BLOCK), 0, stream, in, n,
                                   out, p.y, p.rows, p.alpha);
                MSPMV_CHECK(after_launch(stream, debug_sync, "fixup_kernel", grid, FIX_BLOCK));
                in = out;
            }
        }
    }
    prof_mark(stream, slot, 3);
    return hipSuccess;
}

template <typename V>
static hipError_t dispatch_shape(const Layout &L, void *d_temp, const Params<V> &p, bool axpby, hipStream_t stream,
                                 int debug_sync, const CallExtra &ex);

#define MSPMV_SHAPE_CASE(V, B, I) \
    if (L.shape.block == B && L.shape.ipt == I) return run_shape<V, B, I>(L, d_temp, p, axpby, stream, debug_sync, ex);

template <>
hipError_t dispatch_shape<float>(const Layout &L, void *d_temp, const Params<float> &p, bool axpby, hipStream_t stream,
                                 int debug_sync, const CallExtra &ex)
{
    MSPMV_SHAPE_CASE(float, 256, 7)
    MSPMV_SHAPE_CASE(float, 256, 11)
#ifdef MSPMV_DEV
    MSPMV_SHAPE_CASE(float, 256, 9)
    MSPMV_SHAPE_CASE(float, 256, 15)
    MSPMV_SHAPE_CASE(float, 256, 5)
    MSPMV_SHAPE_CASE(float, 128, 7)
    MSPMV_SHAPE_CASE(float, 512, 7)
#endif
    return hipErrorInvalidValue;
}

template <>
hipError_t dispatch_shape<double>(const Layout &L, void *d_temp, const Params<double> &p, bool axpby,
                                  hipStream_t stream, int debug_sync, const CallExtra &ex)
{
    MSPMV_SHAPE_CASE(double, 256, 7)
    MSPMV_SHAPE_CASE(double, 256, 11)
#ifdef MSPMV_DEV
    MSPMV_SHAPE_CASE(double, 256, 5)
    MSPMV_SHAPE_CASE(double, 256, 9)
    MSPMV_SHAPE_CASE(double, 256, 3)
    MSPMV_SHAPE_CASE(double, 128, 5)
    MSPMV_SHAPE_CASE(double, 512, 5)
#endif
    return hipErrorInvalidValue;
}

// LARGE fp64 MATRICES OF SHORT ROWS OVER A TINY x (the reference's --dense=<cols> inputs: cpu_spmv.cpp:581-587 is --dense=5, BASELINE config 1's
// matrix) take the SMALL tile shape behind the compact front end at any size that still streams with ordinary loads (<= 256 MB): every
// tile of such a matrix is a closed lean tile, and the fast lane's 234 instructions per wave beat the large shape's general kernel --
// --dense=5 at full size (16.8 M nonzeros, 11 235 tiles instead of 7 150) 35.8 -> 34.1 us per call (rocSPARSE: 35.5), 5-point grids of
// 16 M nonzeros 39.8 -> 39.3 (tools/compact_other_matrices.py, profiles/r05_compact_other_matrices.txt).  Judged on what the host knows:
// 8-byte values, the default tuning, a size that would take the large shape, x within the 4 KB that otherwise go to LDS, at most 8
// nonzeros per row on average.  (Matrices with long rows among the short ones lose in the small shape's general body at these sizes --
// R-MAT, 3400 tiles: +6...16 % -- and the host cannot tell them from grids unless x is tiny: a row over x of <= 512 entries is short.)
// Rows of closed lean tiles are summed left to right whatever the tile shape: their y does not change by a bit; a long row among them is
// associated as the shape's tiles cut it, like under any other choice of shape (tools/fuzz.py compares against the general kernel of the shape run).
static bool skinny_rule(int rows, int cols, int nnz, int value_bytes, const Tune &t, const Layout &L)
{
    return value_bytes == 8 && t.block == 0 && t.flags == 0 && t.compact_tiles == 0 && compact_max_tiles(8) > 0 && L.snap && L.shape.ipt != COMPACT_IPT &&
           cols > 0 && (size_t) cols * 8u <= (size_t) X_LDS_MAX_BYTES && (long long) nnz <= 8LL * rows &&
           (unsigned long long) nnz * 12ull + 4ull * (unsigned long long) rows <= (256ull << 20);      // (the dispatcher's `nt` threshold: ordinary loads)
}

template <typename V>
int csrmv_call(void *d_temp, size_t *temp_bytes, const V *d_values, const int32_t *d_row_offsets, const int32_t *d_cols,
               const V *d_x, V *d_y, int32_t rows, int32_t cols, int32_t nnz, V alpha, V beta, bool axpby,
               hipStream_t stream, int debug_sync, const CallExtra &ex)
{
    if (!temp_bytes || rows < 0 || cols < 0 || nnz < 0) return hipErrorInvalidValue;
    if ((long long) rows + nnz > MAX_ITEMS) return hipErrorInvalidValue;
    Layout L = make_layout(rows, nnz, (int) sizeof(V), ex.tune);
    bool skinny = false;
    if (ex.allow_skinny && skinny_rule(rows, cols, nnz, (int) sizeof(V), ex.tune, L)) {
        // The small-shape layout lives BEHIND the default one in temp storage, so a call that takes it never touches the default
        // layout's coordinates (what mspmv_csrmv_prepare stored, what the classic pipeline of a prepared call trusts): the size query
        // asks for both; a caller that sized its storage without the column count (mspmv_get_launch_info) and brings less runs the
        // default shape.  Taken only when the arrays are 16-byte aligned, i.e. when the ONE-LAUNCH kernel will run (its hints are
        // verified, whatever the region held before).
        const Layout S = make_layout(rows, nnz, (int) sizeof(V), ex.tune, true);
        const uint64_t both = L.total + S.total;
        if (d_temp == nullptr) { *temp_bytes = (size_t) both; return hipSuccess; }
        const bool aligned = nnz >= 4 && rows >= 3 &&
                             ((reinterpret_cast<uintptr_t>(d_values) | reinterpret_cast<uintptr_t>(d_cols) | reinterpret_cast<uintptr_t>(d_row_offsets) |
                               reinterpret_cast<uintptr_t>(d_temp)) & 15) == 0;
        if (*temp_bytes >= both && aligned && S.snap) { d_temp = static_cast<char *>(d_temp) + L.total; L = S; skinny = true; }
    }
    if (d_temp == nullptr) {                      // size query (dispatch_spmv_orig.cuh:651-655)
        *temp_bytes = (size_t) L.total;
        return hipSuccess;
    }
    if (!skinny && *temp_bytes < L.total) return hipErrorInvalidValue;   // util_device.cuh:90-93
    // (the temp storage holds 64-bit records updated atomically and is read with scalar loads: 16-byte alignment, which any
    //  device allocation has, is required rather than silently compensated for)
    if (reinterpret_cast<uintptr_t>(d_temp) & 15) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;             // nothing to write
    if (ex.phase == PHASE_COORDS_ONLY) { if (!d_row_offsets) return hipErrorInvalidValue; }
    else if (!d_row_offsets || !d_y || (nnz > 0 && (!d_values || !d_cols || !d_x))) return hipErrorInvalidValue;
    Params<V> p;
    p.values = d_values; p.row_end = d_row_offsets + 1; p.cols = d_cols; p.x = d_x; p.y = d_y;
    p.rows = rows; p.nnz = nnz; p.alpha = alpha; p.beta = beta;
    // a tiny x is gathered from LDS by the vectorised tile kernels (dynamic shared memory of the launch)
    p.x_lds = (cols > 0 && (size_t) cols * sizeof(V) <= (size_t) X_LDS_MAX_BYTES && !(L.flags & MSPMV_TUNE_NO_XLDS)) ? cols : 0;
    p.band_lo = 0; p.band_len = 0; p.band_pass = 0;
    CallExtra ex2 = ex;
    if (skinny) ex2.tune.compact_tiles = 0x7fffffff;          // (the compact front end whatever the tile count)
    ex2.band_passes = band_passes_for(L, (long long) cols * (long long) sizeof(V), (int) sizeof(V), rows, nnz, ex, &ex2.band_force);
    ex2.band_cols = ex2.band_passes > 1 ? (cols + ex2.band_passes - 1) / ex2.band_passes : 0;
    ex2.num_cols = cols;
    ex2.tdm_shift = tdm_shift_for(cols, (int) sizeof(V), ex2.band_passes, ex);
    ex2.tdm_bands = ex2.tdm_shift > 0 ? (int) (((long long) cols + (1LL << ex2.tdm_shift) - 1) >> ex2.tdm_shift) : 0;
    return (int) dispatch_shape<V>(L, d_temp, p, axpby, stream, debug_sync, ex2);
}
template int csrmv_call<float>(void *, size_t *, const float *, const int32_t *, const int32_t *, const float *, float *, int32_t,
                               int32_t, int32_t, float, float, bool, hipStream_t, int, const CallExtra &);
template int csrmv_call<double>(void *, size_t *, const double *, const int32_t *, const int32_t *, const double *, double *,
                                int32_t, int32_t, int32_t, double, double, bool, hipStream_t, int, const CallExtra &);

uint64_t csrmv_temp_bytes(int32_t rows, int32_t nnz, int32_t value_bytes) { return make_layout(rows, nnz, value_bytes, Tune{}).total; }

template <typename V>
static int csrmv_impl(void *d_temp, size_t *temp_bytes, const V *d_values, const int32_t *d_row_offsets,
                      const int32_t *d_cols, const V *d_x, V *d_y, int32_t rows, int32_t cols, int32_t nnz, V alpha,
                      V beta, bool axpby, mspmv_stream_t stream_, int debug_sync, int phase = PHASE_ALL)
{
    CallExtra ex; ex.phase = phase; ex.tune = thread_tune((int) sizeof(V));
    ex.allow_skinny = phase != PHASE_COORDS_ONLY;       // (stateless and prepared calls pick their shape by the same rule: bitwise the same y)
    return csrmv_call<V>(d_temp, temp_bytes, d_values, d_row_offsets, d_cols, d_x, d_y, rows, cols, nnz, alpha, beta, axpby,
                         reinterpret_cast<hipStream_t>(stream_), debug_sync, ex);
}

// ---------------------------------------------------------------------------
// SpMM: Y[rows x k] = alpha * A * X[cols x k] + beta * Y, X and Y row-major (mspmv_spmm.hpp).
// The right-hand sides are taken in groups of one pack: 64- and 32-byte packs first (a whole row
// of X per gather; small tiles -- 128x3 / 256x3 path items -- because a tile's products must fit
// LDS), then 16, 8, 4 bytes on 256x7 tiles.  One coordinate pass per tile size used by the call;
// all groups of one pack width share one tile-kernel launch and one fix-up launch.
// ---------------------------------------------------------------------------
template <typename T, int K> struct MMShape {
    static constexpr int PACK = K * (int) sizeof(T);
    static constexpr int BLOCK = PACK >= 64 ? 128 : 256;
    static constexpr int IPT = PACK >= 32 ? 3 : 7;
    static constexpr int TILE = BLOCK * IPT;
};
#ifndef MSPMV_MM_LANE_IPT
#define MSPMV_MM_LANE_IPT 11
#endif
constexpr int MM_LANE_IPT = MSPMV_MM_LANE_IPT;
constexpr int MM_TILES[4] = {256 * 7, 256 * 3, 128 * 3, 256 * MM_LANE_IPT};      // tile sizes in use: index 0 narrow packs, 1: 32-byte packs, 2: 64-byte, 3: the slot form
// groups of 8 or 16 right-hand sides of at least 32 bytes run the lane-per-column kernel (spmm_lane_kernel) on the 256 x 7 tiles
// right-hand sides per lane of the slot form: four lanes per slot, at most 16 bytes per lane (32 bytes per lane, two lanes per slot:
// 146-158 registers, 20-70 % slower on every matrix tried)
template <typename T, int K> constexpr int mm_lane_vec() { return K / 4 * (int) sizeof(T) <= 16 ? K / 4 : 16 / (int) sizeof(T); }
// (`lane_ok`: X spans less than 4 GB -- the slot form keeps 32-bit byte offsets of X's rows in LDS; beyond that the packs serve)
static constexpr bool mm_lane(int width, int elem_bytes) { return width >= 8 && width * elem_bytes >= 32; }
static constexpr int mm_ti(int width, int elem_bytes, bool lane_ok)
{
    return lane_ok && mm_lane(width, elem_bytes) ? 3 : width * elem_bytes >= 64 ? 2 : width * elem_bytes >= 32 ? 1 : 0;
}

// generic row-per-thread fallback (arrays not 16-byte aligned, or fewer than 4 nonzeros / 3 rows)
template <typename T>
__global__ void spmm_rowwise_kernel(const T *values, const int *row_offsets, const int *cols, const T *x, T *y, int rows, int k,
                                    int ldx, int ldy, T alpha, T beta)
{
    const long long id = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long) rows * k) return;
    const int r = (int) (id / k), c = (int) (id % k);
    T sum = 0;
    for (int j = row_offsets[r]; j < row_offsets[r + 1]; ++j) sum += values[j] * x[(size_t) cols[j] * ldx + c];
    T *dst = y + (size_t) r * ldy + c;
    *dst = alpha * sum + (beta == (T) 0 ? (T) 0 : beta * *dst);
}

// how a block of k right-hand sides is cut into groups: count[i] groups of width[i] columns
template <typename T> struct MMPlan { int width[5], count[5], n; };
// Wide packs (32 / 64 bytes: a whole row of X per gather, but small tiles and few resident waves) pay
// when the gathers miss -- X larger than a cache -- and cost when they hit anyway: measured, a 2 KB X
// (dense 32-column matrix) runs k = 16 in 1.00 ms with four 16-byte groups and 1.42 ms with one 64-byte
// pack, a 200 MB X (C2) in 7.4 ms vs 1.9 ms.  The X footprint decides -- for the PACK kernel; groups of 8 / 16 columns run the
// lane-per-column kernel whatever the footprint (round 6: the same two cases in 0.3 and 1.9 ms).
template <typename T>
static MMPlan<T> make_mm_plan(int k, bool wide, bool lane_ok)
{
    MMPlan<T> pl; pl.n = 0;
    for (int w = 16; w >= 1; w >>= 1) {
        const bool lane = lane_ok && mm_lane(w, (int) sizeof(T));
        if (!lane && (w * (int) sizeof(T) > 64 || !(wide || w * (int) sizeof(T) <= 16))) continue;
        pl.width[pl.n] = w; pl.count[pl.n] = k / w; k -= pl.count[pl.n] * w; ++pl.n;
    }
    return pl;
}

struct MMLayout { int num_tiles[4]; uint64_t coords_off[4], carries_off, total; };
template <typename T>
static MMLayout make_mm_layout(int rows, int nnz, int k, bool wide, bool lane_ok)
{
    MMLayout L; memset(&L, 0, sizeof(L));
    const long long total = (long long) rows + nnz;
    const MMPlan<T> pl = make_mm_plan<T>(k, wide, lane_ok);
    uint64_t off = 0, carry_bytes = 256;
    bool used[4] = {false, false, false, false};
    for (int i = 0; i < pl.n; ++i) {
        if (pl.count[i] == 0) continue;
        const int pack = pl.width[i] * (int) sizeof(T);
        const int ti = mm_ti(pl.width[i], (int) sizeof(T), lane_ok);
        used[ti] = true;
        const uint64_t tiles = (uint64_t) ((total + MM_TILES[ti] - 1) / MM_TILES[ti]);
        carry_bytes = std::max<uint64_t>(carry_bytes, (tiles > 0 ? tiles : 1) * pl.count[i] * (uint64_t) (sizeof(T) + pack));   // sizeof(CarryMM): key padded to the element size
    }
    for (int ti = 0; ti < 4; ++ti) {
        L.num_tiles[ti] = (int) ((total + MM_TILES[ti] - 1) / MM_TILES[ti]);
        if (used[ti]) { L.coords_off[ti] = off; off = align256(off + uint64_t(L.num_tiles[ti] + 1) * sizeof(Coord)); }
    }
    L.carries_off = off; off = align256(off + carry_bytes);
    L.total = off;
    return L;
}

template <typename T, int K, bool LANE>
static hipError_t run_mm_group(const MMLayout &L, char *base, MMParams<T> p, int groups, bool axpby, bool nt, hipStream_t stream,
                               int debug_sync)
{
    typedef MMShape<T, K> S;
    constexpr int ti = mm_ti(K, (int) sizeof(T), LANE);
    const Coord *coords = reinterpret_cast<const Coord *>(base + L.coords_off[ti]);
    CarryMM<T, K> *carries = reinterpret_cast<CarryMM<T, K> *>(base + L.carries_off);
    static_assert(sizeof(CarryMM<T, K>) == sizeof(T) + K * sizeof(T), "carry layout assumed by make_mm_layout");
    const uintptr_t pack_bytes = LANE ? mm_lane_vec<T, K>() * sizeof(T) : K * sizeof(T);     // (the slot form: four lanes per slot, K / 4 right-hand sides per lane)
    p.x_vec = (reinterpret_cast<uintptr_t>(p.x) % pack_bytes == 0 && ((uintptr_t) p.ldx * sizeof(T)) % pack_bytes == 0) ? 1 : 0;
    p.y_vec = (reinterpret_cast<uintptr_t>(p.y) % pack_bytes == 0 && ((uintptr_t) p.ldy * sizeof(T)) % pack_bytes == 0) ? 1 : 0;
    const int num_tiles = L.num_tiles[ti];
    const unsigned grid = (unsigned) num_tiles;
    constexpr int mm_chunk = MM_CHUNK_LOG2;
    if constexpr (LANE) {
#define MSPMV_MM_LAUNCH(AX, NTF) hipLaunchKernelGGL((spmm_lane_kernel<T, K, mm_lane_vec<T, K>(), MM_LANE_BLOCK, 256 * MM_LANE_IPT / MM_LANE_BLOCK, AX, NTF>), dim3(grid), dim3(MM_LANE_BLOCK), 0, stream, p, coords, carries, num_tiles, groups, mm_chunk)
        if (axpby) { if (nt) MSPMV_MM_LAUNCH(true, true); else MSPMV_MM_LAUNCH(true, false); }
        else { if (nt) MSPMV_MM_LAUNCH(false, true); else MSPMV_MM_LAUNCH(false, false); }
#undef MSPMV_MM_LAUNCH
        MSPMV_CHECK(after_launch(stream, debug_sync, "spmm_lane_kernel", grid, MM_LANE_BLOCK));
    } else {
#define MSPMV_MM_LAUNCH(AX, NTF) hipLaunchKernelGGL((spmm_tile_kernel<T, K, S::BLOCK, S::IPT, AX, NTF>), dim3(grid), dim3(S::BLOCK), 0, stream, p, coords, carries, num_tiles, groups, mm_chunk)
        if (axpby) { if (nt) MSPMV_MM_LAUNCH(true, true); else MSPMV_MM_LAUNCH(true, false); }
        else { if (nt) MSPMV_MM_LAUNCH(false, true); else MSPMV_MM_LAUNCH(false, false); }
#undef MSPMV_MM_LAUNCH
        MSPMV_CHECK(after_launch(stream, debug_sync, "spmm_tile_kernel", grid, S::BLOCK));
    }
    if (num_tiles > 1) {
        const unsigned fgrid = (unsigned) ((num_tiles + FIX_CHUNK - 1) / FIX_CHUNK);
        hipLaunchKernelGGL((spmm_fixup_kernel<T, K, FIX_BLOCK, FIX_IPT>), dim3(fgrid, (unsigned) groups), dim3(FIX_BLOCK), 0, stream, carries, num_tiles,
                           p.y, p.ldy, p.y_vec, p.rows, p.alpha);
        MSPMV_CHECK(after_launch(stream, debug_sync, "spmm_fixup_kernel", fgrid, FIX_BLOCK));
    }
    return hipSuccess;
}

template <typename T, int W>
static hipError_t run_mm_width(int width, bool lane_ok, const MMLayout &L, char *base, const MMParams<T> &p, int groups, bool axpby, bool nt,
                               hipStream_t stream, int debug_sync)
{
    if (width == W) {
        if constexpr (mm_lane(W, (int) sizeof(T))) {
            if (lane_ok) return run_mm_group<T, W, true>(L, base, p, groups, axpby, nt, stream, debug_sync);
        }
        if constexpr (W * sizeof(T) <= 64) return run_mm_group<T, W, false>(L, base, p, groups, axpby, nt, stream, debug_sync);
        else return hipErrorInvalidValue;
    }
    if constexpr (W > 1) return run_mm_width<T, W / 2>(width, lane_ok, L, base, p, groups, axpby, nt, stream, debug_sync);
    return hipErrorInvalidValue;
}

template <typename T>
static int csrmm_impl(void *d_temp, size_t *temp_bytes, const T *d_values, const int32_t *d_row_offsets, const int32_t *d_cols,
                      const T *d_x, int32_t ldx, T *d_y, int32_t ldy, int32_t rows, int32_t cols, int32_t nnz, int32_t k, T alpha,
                      T beta, mspmv_stream_t stream_, int debug_sync)
{
    if (!temp_bytes || rows < 0 || cols < 0 || nnz < 0 || k < 0 || ldx < k || ldy < k) return hipErrorInvalidValue;
    if ((long long) rows + nnz > MAX_ITEMS) return hipErrorInvalidValue;
    const unsigned long long x_bytes = (unsigned long long) cols * (unsigned long long) ldx * sizeof(T);
    const bool wide = x_bytes > (1ull << 20);
    // the slot form (groups of 8 / 16 columns): 32-bit byte offsets into X, and enough tiles to fill the chip (its blocks live long:
    // R-MAT scale 20, 4 M path items, k = 8: 0.152 ms against the packs' 0.08)
    const bool lane_ok = x_bytes < (1ull << 32) && (long long) rows + nnz >= (8ll << 20);
    const MMLayout L = make_mm_layout<T>(rows, nnz, k, wide, lane_ok);
    // one right-hand side stored as a plain vector IS CsrMV: the call goes there (one launch, the column-band passes, the
    // compact front end) -- the temp storage asked for covers both layouts
    const bool as_csrmv = k == 1 && ldx == 1 && ldy == 1;
    size_t need = (size_t) L.total;
    if (as_csrmv) {
        size_t mv = 0;
        const int st = csrmv_impl<T>(nullptr, &mv, nullptr, nullptr, nullptr, nullptr, nullptr, rows, cols, nnz, alpha, beta, true, stream_, 0);
        if (st != 0) return st;
        need = std::max(need, mv);
    }
    if (d_temp == nullptr) { *temp_bytes = need; return hipSuccess; }
    if (*temp_bytes < need || (reinterpret_cast<uintptr_t>(d_temp) & 15)) return hipErrorInvalidValue;
    if (rows == 0 || k == 0) return hipSuccess;
    if (!d_row_offsets || !d_y || (nnz > 0 && (!d_values || !d_cols || !d_x))) return hipErrorInvalidValue;
    if (as_csrmv)
        return csrmv_impl<T>(d_temp, temp_bytes, d_values, d_row_offsets, d_cols, d_x, d_y, rows, cols, nnz, alpha, beta,
                             !(alpha == (T) 1 && beta == (T) 0), stream_, debug_sync);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const bool vec = nnz >= 4 && rows >= 3 &&
                     ((reinterpret_cast<uintptr_t>(d_values) | reinterpret_cast<uintptr_t>(d_cols) |
                       reinterpret_cast<uintptr_t>(d_row_offsets)) & 15) == 0;
    if (!vec) {
        const long long work = (long long) rows * k;
        const unsigned grid = (unsigned) ((work + 255) / 256);
        hipLaunchKernelGGL((spmm_rowwise_kernel<T>), dim3(grid), dim3(256), 0, stream, d_values, d_row_offsets, d_cols, d_x, d_y, rows,
                           k, ldx, ldy, alpha, beta);
        return (int) after_launch(stream, debug_sync, "spmm_rowwise_kernel", grid, 256);
    }
    char *base = static_cast<char *>(d_temp);
    const MMPlan<T> pl = make_mm_plan<T>(k, wide, lane_ok);
    bool have_coords[4] = {false, false, false, false};
    const bool axpby = !(alpha == (T) 1 && beta == (T) 0);
    const unsigned long long stream_bytes = (unsigned long long) nnz * (sizeof(T) + 4) + 4ull * rows;
    const bool nt = stream_bytes > (200ull << 20);
    MMParams<T> p;
    p.values = d_values; p.row_end = d_row_offsets + 1; p.cols = d_cols; p.rows = rows; p.nnz = nnz; p.ldx = ldx; p.ldy = ldy;
    p.alpha = alpha; p.beta = beta; p.x_vec = p.y_vec = 0;
    int c = 0;
    for (int i = 0; i < pl.n; ++i) {
        if (pl.count[i] == 0) continue;
        const int ti = mm_ti(pl.width[i], (int) sizeof(T), lane_ok);
        if (!have_coords[ti]) {                   // tile coordinates for this tile size, once
            Coord *coords = reinterpret_cast<Coord *>(base + L.coords_off[ti]);
            const long long threads = ((long long) rows + 1 + 3) / 4;
            const unsigned grid = (unsigned) ((threads + SEARCH_BLOCK - 1) / SEARCH_BLOCK);
            if (ti == 0) hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, 256 * 7, true>), dim3(grid), dim3(SEARCH_BLOCK), 0, stream, d_row_offsets, rows, nnz, L.num_tiles[0], BoundaryOut{coords, nullptr}, BandDetectArgs{});
            else if (ti == 3) hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, 256 * MM_LANE_IPT, true>), dim3(grid), dim3(SEARCH_BLOCK), 0, stream, d_row_offsets, rows, nnz, L.num_tiles[3], BoundaryOut{coords, nullptr}, BandDetectArgs{});
            else if (ti == 1) hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, 256 * 3, true>), dim3(grid), dim3(SEARCH_BLOCK), 0, stream, d_row_offsets, rows, nnz, L.num_tiles[1], BoundaryOut{coords, nullptr}, BandDetectArgs{});
            else hipLaunchKernelGGL((coords_scatter_kernel<SEARCH_BLOCK, 128 * 3, true>), dim3(grid), dim3(SEARCH_BLOCK), 0, stream, d_row_offsets, rows, nnz, L.num_tiles[2], BoundaryOut{coords, nullptr}, BandDetectArgs{});
            MSPMV_CHECK(after_launch(stream, debug_sync, "coords_scatter_kernel", grid, SEARCH_BLOCK));
            have_coords[ti] = true;
        }
        p.x = d_x + c; p.y = d_y + c;
        const hipError_t e = run_mm_width<T, 16>(pl.width[i], lane_ok, L, base, p, pl.count[i], axpby, nt, stream, debug_sync);
        if (e != hipSuccess) return (int) e;
        c += pl.count[i] * pl.width[i];
    }
    return hipSuccess;
}

}  // namespace mspmv

using namespace mspmv;

extern "C" {

int mspmv_version(void) { return MSPMV_VERSION; }

const char *mspmv_error_string(int status) { return hipGetErrorString((hipError_t) status); }

int mspmv_csrmv_f32(void *d_temp, size_t *temp_bytes, const float *d_values, const int32_t *d_row_offsets,
                    const int32_t *d_column_indices, const float *d_x, float *d_y, int32_t rows, int32_t cols,
                    int32_t nnz, mspmv_stream_t stream, int debug_sync)
{
    return csrmv_impl<float>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, d_y, rows, cols, nnz,
                             1.0f, 0.0f, false, stream, debug_sync);
}

int mspmv_csrmv_f64(void *d_temp, size_t *temp_bytes, const double *d_values, const int32_t *d_row_offsets,
                    const int32_t *d_column_indices, const double *d_x, double *d_y, int32_t rows, int32_t cols,
                    int32_t nnz, mspmv_stream_t stream, int debug_sync)
{
    return csrmv_impl<double>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, d_y, rows, cols, nnz,
                              1.0, 0.0, false, stream, debug_sync);
}

int mspmv_csrmv_axpby_f32(void *d_temp, size_t *temp_bytes, const float *d_values, const int32_t *d_row_offsets,
                          const int32_t *d_column_indices, const float *d_x, float *d_y, int32_t rows, int32_t cols,
                          int32_t nnz, float alpha, float beta, mspmv_stream_t stream, int debug_sync)
{
    return csrmv_impl<float>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, d_y, rows, cols, nnz,
                             alpha, beta, true, stream, debug_sync);
}

int mspmv_csrmv_axpby_f64(void *d_temp, size_t *temp_bytes, const double *d_values, const int32_t *d_row_offsets,
                          const int32_t *d_column_indices, const double *d_x, double *d_y, int32_t rows, int32_t cols,
                          int32_t nnz, double alpha, double beta, mspmv_stream_t stream, int debug_sync)
{
    return csrmv_impl<double>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, d_y, rows, cols, nnz,
                              alpha, beta, true, stream, debug_sync);
}

int mspmv_csrmv_prepare(void *d_temp, size_t *temp_bytes, const int32_t *d_row_offsets, int32_t rows, int32_t nnz,
                        int32_t value_bytes, mspmv_stream_t stream, int debug_sync)
{
    if (value_bytes == 4)
        return csrmv_impl<float>(d_temp, temp_bytes, nullptr, d_row_offsets, nullptr, nullptr, nullptr, rows, 0, nnz, 1.0f, 0.0f,
                                 false, stream, debug_sync, PHASE_COORDS_ONLY);
    if (value_bytes == 8)
        return csrmv_impl<double>(d_temp, temp_bytes, nullptr, d_row_offsets, nullptr, nullptr, nullptr, rows, 0, nnz, 1.0, 0.0,
                                  false, stream, debug_sync, PHASE_COORDS_ONLY);
    return hipErrorInvalidValue;
}

int mspmv_csrmv_prepared_f32(void *d_temp, size_t *temp_bytes, const float *d_values, const int32_t *d_row_offsets,
                             const int32_t *d_column_indices, const float *d_x, float *d_y, int32_t rows, int32_t cols,
                             int32_t nnz, float alpha, float beta, mspmv_stream_t stream, int debug_sync)
{
    if (!d_temp) return hipErrorInvalidValue;
    return csrmv_impl<float>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, d_y, rows, cols, nnz,
                             alpha, beta, !(alpha == 1.0f && beta == 0.0f), stream, debug_sync, PHASE_SKIP_COORDS);
}

int mspmv_csrmv_prepared_f64(void *d_temp, size_t *temp_bytes, const double *d_values, const int32_t *d_row_offsets,
                             const int32_t *d_column_indices, const double *d_x, double *d_y, int32_t rows, int32_t cols,
                             int32_t nnz, double alpha, double beta, mspmv_stream_t stream, int debug_sync)
{
    if (!d_temp) return hipErrorInvalidValue;
    return csrmv_impl<double>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, d_y, rows, cols, nnz,
                              alpha, beta, !(alpha == 1.0 && beta == 0.0), stream, debug_sync, PHASE_SKIP_COORDS);
}

int mspmv_csrmm_f32(void *d_temp, size_t *temp_bytes, const float *d_values, const int32_t *d_row_offsets,
                    const int32_t *d_column_indices, const float *d_x, int32_t ldx, float *d_y, int32_t ldy, int32_t rows,
                    int32_t cols, int32_t nnz, int32_t k, float alpha, float beta, mspmv_stream_t stream, int debug_sync)
{
    return csrmm_impl<float>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, ldx, d_y, ldy, rows, cols, nnz, k,
                             alpha, beta, stream, debug_sync);
}

int mspmv_csrmm_f64(void *d_temp, size_t *temp_bytes, const double *d_values, const int32_t *d_row_offsets,
                    const int32_t *d_column_indices, const double *d_x, int32_t ldx, double *d_y, int32_t ldy, int32_t rows,
                    int32_t cols, int32_t nnz, int32_t k, double alpha, double beta, mspmv_stream_t stream, int debug_sync)
{
    return csrmm_impl<double>(d_temp, temp_bytes, d_values, d_row_offsets, d_column_indices, d_x, ldx, d_y, ldy, rows, cols, nnz, k,
                              alpha, beta, stream, debug_sync);
}

static int launch_info_impl(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, mspmv_launch_info_t *info)
{
    if (!info || (value_bytes != 4 && value_bytes != 8) || rows < 0 || nnz < 0 ||
        (long long) rows + nnz > MAX_ITEMS)
        return hipErrorInvalidValue;
    const Tune &tune = thread_tune(value_bytes);
    Layout L = make_layout(rows, nnz, value_bytes, tune);
    // cols < 0 (mspmv_get_launch_info: the column count is not known): the default layout, with temp_bytes large enough for the small
    // shape a large fp64 matrix of short rows over a tiny x takes (skinny_rule) -- a buffer sized from here serves either;
    // cols >= 0 (mspmv_get_launch_info_cols): the layout a stateless call of exactly these sizes runs
    uint64_t temp_bytes = L.total, base = 0;
    if (cols < 0) {
        if (skinny_rule(rows, 1, nnz, value_bytes, tune, L))      // (the rule holds for SOME column count)
            temp_bytes = L.total + make_layout(rows, nnz, value_bytes, tune, true).total;
    } else if (skinny_rule(rows, cols, nnz, value_bytes, tune, L)) {
        base = L.total;                                           // (the small-shape layout sits behind the default one: csrmv_call)
        L = make_layout(rows, nnz, value_bytes, tune, true);
        temp_bytes = base + L.total;
    }
    memset(info, 0, sizeof(*info));
    info->block_threads = L.shape.block;
    info->items_per_thread = L.shape.ipt;
    info->tile_items = L.shape.block * L.shape.ipt;
    info->num_tiles = L.num_tiles;
    info->fixup_chunk = FIX_CHUNK;
    info->fixup_levels = L.snap ? 0 : L.fix_levels;     // (fix-up launches; 0: the tiles add the carries themselves)
    info->flags = L.flags;
    info->snap_head_max = L.snap ? snap_head_max_host(L.shape.block, L.shape.ipt) : 0;
    info->temp_bytes = temp_bytes;
    info->coords_offset = base + L.coords_off;
    info->carries_offset = base + L.carries_off;
    info->diag_offset = base + L.err_off;
    info->records_offset = base + L.pub_off;
    return hipSuccess;
}
int mspmv_get_launch_info(int32_t rows, int32_t nnz, int32_t value_bytes, mspmv_launch_info_t *info) { return launch_info_impl(rows, -1, nnz, value_bytes, info); }
int mspmv_get_launch_info_cols(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, mspmv_launch_info_t *info)
{
    return cols < 0 ? hipErrorInvalidValue : launch_info_impl(rows, cols, nnz, value_bytes, info);
}

int mspmv_debug_read_tiles(const void *d_temp, int32_t rows, int32_t nnz, int32_t value_bytes, int32_t *h_coords,
                           int32_t *h_carry_keys, void *h_carry_values, mspmv_stream_t stream_)
{
    if (!d_temp || (value_bytes != 4 && value_bytes != 8) || rows < 0 || nnz < 0 || (long long) rows + nnz > MAX_ITEMS) return hipErrorInvalidValue;
    const Layout L = make_layout(rows, nnz, value_bytes, thread_tune(value_bytes));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    MSPMV_CHECK(hipStreamSynchronize(stream));
    const char *base = static_cast<const char *>(d_temp);
    if (h_coords)
        MSPMV_CHECK(hipMemcpy(h_coords, base + L.coords_off, sizeof(Coord) * size_t(L.num_tiles + 1),
                              hipMemcpyDeviceToHost));
    if ((h_carry_keys || h_carry_values) && L.num_tiles > 0) {
        const size_t pair = value_bytes == 8 ? 16 : 8;
        std::vector<char> tmp(pair * size_t(L.num_tiles));
        MSPMV_CHECK(hipMemcpy(tmp.data(), base + L.carries_off, tmp.size(), hipMemcpyDeviceToHost));
        for (int t = 0; t < L.num_tiles; ++t) {
            const char *rec = tmp.data() + pair * size_t(t);
            if (h_carry_keys) memcpy(&h_carry_keys[t], rec, 4);
            if (h_carry_values)
                memcpy(static_cast<char *>(h_carry_values) + size_t(value_bytes) * t, rec + (value_bytes == 8 ? 8 : 4),
                       size_t(value_bytes));
        }
    }
    return hipSuccess;
}

#ifdef MSPMV_TUNING
int mspmv_set_tuning(int32_t value_bytes, int32_t block_threads, int32_t items_per_thread, int32_t flags)
{
    if (value_bytes != 4 && value_bytes != 8) return hipErrorInvalidValue;
    const Shape *tab = value_bytes == 8 ? kShapesF64 : kShapesF32;
    const int count = value_bytes == 8 ? int(sizeof(kShapesF64) / sizeof(Shape)) : int(sizeof(kShapesF32) / sizeof(Shape));
    int allowed = MSPMV_TUNE_TWO_LAUNCH | MSPMV_TUNE_SCATTER_COORDS | MSPMV_TUNE_INTERP_COORDS | MSPMV_TUNE_NO_XLDS | MSPMV_TUNE_ATOMIC_FIX | MSPMV_TUNE_NO_VEC | MSPMV_TUNE_BINARY_SEARCH | MSPMV_TUNE_NO_FUSED |
                  MSPMV_TUNE_FORCE_NT | MSPMV_TUNE_FORCE_TEMPORAL | MSPMV_TUNE_MULTILEVEL_FIX | MSPMV_TUNE_NO_LEAN | 0xf000000;
#ifdef MSPMV_DEV
    allowed |= MSPMV_DEV_FLAG_BITS;        // development kernels (mspmv_dev.hpp): never in the product library
#endif
    if (flags & ~allowed) return hipErrorInvalidValue;
    Tune &t = t_tune[value_bytes == 8];
    if (block_threads == 0 && items_per_thread == 0) { t.block = 0; t.ipt = 0; t.flags = flags; return hipSuccess; }
    for (int i = 0; i < count; ++i)
        if (tab[i].block == block_threads && tab[i].ipt == items_per_thread) {
            t.block = block_threads; t.ipt = items_per_thread; t.flags = flags;
            return hipSuccess;
        }
    return hipErrorInvalidValue;
}

int mspmv_set_band_passes(int32_t value_bytes, int32_t passes)
{
    if ((value_bytes != 4 && value_bytes != 8) || passes == 1 || passes > 64) return hipErrorInvalidValue;
    t_tune[value_bytes == 8].band_passes = passes < 0 ? -1 : passes;
    return hipSuccess;
}

int mspmv_set_tdm(int32_t value_bytes, int32_t policy, int32_t slot_permille, int32_t lookahead_plus_1, int32_t band_shift)
{
    if ((value_bytes != 4 && value_bytes != 8) || slot_permille < 0 || lookahead_plus_1 < 0 || lookahead_plus_1 > 33 || band_shift < 0 || band_shift > TDM_SLOT_SHIFT)
        return hipErrorInvalidValue;
    Tune &t = t_tune[value_bytes == 8];
    t.tdm = policy < 0 ? -1 : policy > 0 ? 1 : 0; t.tdm_slot_permille = slot_permille; t.tdm_lookahead = lookahead_plus_1; t.tdm_band_shift = band_shift;
    return hipSuccess;
}

#endif

int mspmv_get_device_caches(int64_t *l2_bytes_per_xcd, int32_t *xcds, int32_t *cus)
{
    const DeviceCaches dc = device_caches();
    if (l2_bytes_per_xcd) *l2_bytes_per_xcd = dc.l2_bytes;
    if (xcds) *xcds = dc.xcds;
    if (cus) *cus = device_cus();
    return hipSuccess;
}

int mspmv_probe_read_stream(const void *d_buf, size_t bytes, int32_t nontemporal, mspmv_stream_t stream_)
{
    if (!d_buf || (reinterpret_cast<uintptr_t>(d_buf) & 15) != 0) return hipErrorInvalidValue;
    const unsigned long long n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const unsigned long long blocks = (n16 + 256ull * 11 - 1) / (256ull * 11);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (nontemporal) hipLaunchKernelGGL((probe_read_kernel<true>), dim3((unsigned) blocks), dim3(256), 0, stream, static_cast<const int4v *>(d_buf), n16);
    else hipLaunchKernelGGL((probe_read_kernel<false>), dim3((unsigned) blocks), dim3(256), 0, stream, static_cast<const int4v *>(d_buf), n16);
    return (int) hipGetLastError();
}

#ifdef MSPMV_TUNING
int mspmv_set_record_polls(int32_t polls)
{
    t_tune[0].record_polls = t_tune[1].record_polls = polls < 0 ? -1 : polls;
    return hipSuccess;
}
int mspmv_set_compact_tiles(int32_t max_tiles)
{
    t_tune[0].compact_tiles = t_tune[1].compact_tiles = max_tiles < 0 ? -1 : max_tiles;
    return hipSuccess;
}
#endif

int mspmv_get_band_passes(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, int32_t *passes)
{
    if (!passes || rows < 0 || cols < 0 || nnz < 0 || (value_bytes != 4 && value_bytes != 8) || (long long) rows + nnz > MAX_ITEMS)
        return hipErrorInvalidValue;
    const Layout L = make_layout(rows, nnz, value_bytes, thread_tune(value_bytes));
    CallExtra ex; ex.tune = thread_tune(value_bytes); int force = 0;
    *passes = band_passes_for(L, (long long) cols * value_bytes, value_bytes, rows, nnz, ex, &force);
    return hipSuccess;
}

int mspmv_get_clocked_bands(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, int32_t *bands, int32_t *band_cols)
{
    int32_t passes = 0;
    if (!bands || !band_cols) return hipErrorInvalidValue;
    const int st = mspmv_get_band_passes(rows, cols, nnz, value_bytes, &passes);
    if (st != hipSuccess) return st;
    CallExtra ex; ex.tune = thread_tune(value_bytes);
    const int shift = tdm_shift_for(cols, value_bytes, passes, ex);
    *bands = shift > 0 ? (int32_t) (((long long) cols + (1LL << shift) - 1) >> shift) : 0;
    *band_cols = shift > 0 ? (int32_t) (1 << shift) : 0;
    return hipSuccess;
}

int mspmv_debug_band_windows(const void *d_temp, int32_t rows, int32_t nnz, int32_t value_bytes, int32_t *h_verdicts,
                             mspmv_stream_t stream_)
{
    if (!d_temp || !h_verdicts || (value_bytes != 4 && value_bytes != 8) || rows < 0 || nnz < 0 || (long long) rows + nnz > MAX_ITEMS)
        return hipErrorInvalidValue;
    const Layout L = make_layout(rows, nnz, value_bytes, thread_tune(value_bytes));
    MSPMV_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream_)));
    return (int) hipMemcpy(h_verdicts, static_cast<const char *>(d_temp) + L.band_off, sizeof(int32_t) * BAND_WINDOWS, hipMemcpyDeviceToHost);
}

int mspmv_profile_begin(int32_t max_calls)
{
    if (max_calls < 1 || max_calls > (1 << 20)) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> g(g_prof.lock);
    for (hipEvent_t e : g_prof.events) if (e) (void) hipEventDestroy(e);
    g_prof.events.assign(size_t(max_calls) * PROF_EVENTS, nullptr);
    for (auto &e : g_prof.events) MSPMV_CHECK(hipEventCreate(&e));
    g_prof.capacity = max_calls; g_prof.calls = 0; g_prof.active = true;
    return hipSuccess;
}

int mspmv_profile_end(int32_t *calls, float *search_ms, float *tile_ms, float *fixup_ms)
{
    std::lock_guard<std::mutex> g(g_prof.lock);
    g_prof.active = false;
    double acc[3] = {0, 0, 0};
    int n = 0;
    hipError_t first_error = hipSuccess;
    for (int c = 0; c < g_prof.calls; ++c) {
        // a slot whose four marks were not all recorded (a call that failed half way) is skipped
        float ms[3]; bool ok = hipEventSynchronize(g_prof.events[size_t(c) * PROF_EVENTS + 3]) == hipSuccess;
        for (int k = 0; k < 3 && ok; ++k)
            ok = hipEventElapsedTime(&ms[k], g_prof.events[size_t(c) * PROF_EVENTS + k], g_prof.events[size_t(c) * PROF_EVENTS + k + 1]) == hipSuccess;
        if (!ok) { (void) hipGetLastError(); continue; }
        for (int k = 0; k < 3; ++k) acc[k] += ms[k];
        ++n;
    }
    if (calls) *calls = n;
    if (search_ms) *search_ms = n ? float(acc[0] / n) : 0.f;
    if (tile_ms) *tile_ms = n ? float(acc[1] / n) : 0.f;
    if (fixup_ms) *fixup_ms = n ? float(acc[2] / n) : 0.f;
    for (hipEvent_t e : g_prof.events) if (e) (void) hipEventDestroy(e);     // always cleaned up
    g_prof.events.clear(); g_prof.capacity = 0; g_prof.calls = 0;
    return first_error;
}

int mspmv_mg_apply_carries(void *d_y_local, const void *d_carries, const int64_t *row_split, int32_t parts,
                           int32_t part, int32_t value_bytes, mspmv_stream_t stream_)
{
    if (!d_y_local || !d_carries || !row_split || parts < 1 || parts > MSPMV_MG_MAX_PARTS || part < 0 || part >= parts ||
        (value_bytes != 4 && value_bytes != 8))
        return hipErrorInvalidValue;
    // this part owns y for global rows [row_split[part], row_split[part+1]); a
    // carry of part j (key row_split[j+1]) lands here iff the key equals our
    // first row and we own at least one row.
    if (row_split[part + 1] <= row_split[part]) return hipSuccess;
    unsigned long long mask = 0;
    for (int j = 0; j < part; ++j)
        if (row_split[j + 1] == row_split[part]) mask |= 1ull << j;
    if (!mask) return hipSuccess;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (value_bytes == 8)
        hipLaunchKernelGGL((mg_apply_kernel<double>), dim3(1), dim3(64), 0, stream, static_cast<double *>(d_y_local),
                           static_cast<const double *>(d_carries), mask, parts);
    else
        hipLaunchKernelGGL((mg_apply_kernel<float>), dim3(1), dim3(64), 0, stream, static_cast<float *>(d_y_local),
                           static_cast<const float *>(d_carries), mask, parts);
    return (int) hipGetLastError();
}

}  // extern "C"

#ifdef MSPMV_DEV
// development: where tile_kernel_vec<..., ABLATE = 6> writes its cycle stamps (device pointer, 16*8 u64 per block)
extern "C" int mspmv_dev_set_trace(void *d_buf)
{
    unsigned long long *ptr = static_cast<unsigned long long *>(d_buf);
    return (int) hipMemcpyToSymbol(HIP_SYMBOL(mspmv::g_mspmv_trace), &ptr, sizeof(ptr));
}
#endif
