// mspmv_internal.hpp -- what the translation units of libmspmv.so share below the C ABI:
// the dispatcher of mspmv_api.hip as a C++ function (the prepared band-major plan, mspmv_plan.hip,
// and the multi-GPU plan, mspmv_mg_plan.hip, run the ordinary merge-path CsrMV on matrices they own).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#if defined(MSPMV_DEV) && !defined(MSPMV_TUNING)
#define MSPMV_TUNING 1      // the experiment build includes the tuning overrides
#endif
#include "../../include/mspmv_dev.h"      // (for the MSPMV_TUNE_* option bits the dispatcher is written in; the setters it declares are DEFINED in the -DMSPMV_TUNING build only)

namespace mspmv {

// what a call does -- everything (the reference's stateless call), only the tile-coordinate pass
// (mspmv_csrmv_prepare), or everything but it (mspmv_csrmv_prepared_*: the coordinates in d_temp
// depend on row_offsets alone and are reused across the SpMVs of a solver)
enum { PHASE_ALL = 0, PHASE_COORDS_ONLY = 1, PHASE_SKIP_COORDS = 2 };

// Tile shape / option bits / band-pass policy of ONE call (0 everywhere = the library defaults).  The C entry points copy
// the calling host thread's development override (mspmv_set_tuning, mspmv_set_band_passes) into it once, on entry; the
// prepared band-major plan and the multi-GPU plan always pass the defaults, so what they store at build time (tile
// coordinates) can never disagree with what a later apply derives.
struct Tune {
    int block = 0, ipt = 0, flags = 0, band_passes = 0, record_polls = 0, compact_tiles = 0;
    // clock-scheduled column bands (mspmv_tdm.hpp) for the calls the column-band passes are offered to: 0 = the library's rule,
    // < 0 never (the passes as before), > 0 always; slot length in per mille of the computed one, lookahead bands (0 = defaults)
    int tdm = 0, tdm_slot_permille = 0, tdm_lookahead = 0, tdm_band_shift = 0;
};

struct CallExtra {
    int phase = PHASE_ALL;
    Tune tune;
    int tile_map = 0;       // 0: library default (XCD-chunked runs of 64 tiles); else the chunk_log2 code of
                            // xcd_chunked_tile (TILE_MAP_CONTIGUOUS_CODE: one contiguous tile range per XCD)
    // column-band passes (filled in by csrmv_call, see band_passes_for): > 1 = tile_kernel_band may serve the call
    int band_passes = 0, band_cols = 0, band_force = 0, num_cols = 0;
    int tdm_shift = 0, tdm_bands = 0;   // > 0: the clock-scheduled one-pass form serves the call when the windows say "spread" (mspmv_tdm.hpp)
    bool no_bands = false;  // callers that must not take them (the band-major plan: its stacked matrix is banded already)
    bool allow_skinny = false;   // the stateless public calls only: a large fp64 matrix of short rows over a tiny x may take the small tile shape (mspmv_api.hip: skinny_rule)
};
constexpr int TILE_MAP_CONTIGUOUS_CODE = 30;

// The dispatcher behind mspmv_csrmv_* / _axpby_* / _prepare / _prepared_*; same conventions
// (two-phase temp storage, caller-owned buffers, asynchronous on `stream`).
template <typename V>
int csrmv_call(void *d_temp, size_t *temp_bytes, const V *d_values, const int32_t *d_row_offsets, const int32_t *d_cols,
               const V *d_x, V *d_y, int32_t rows, int32_t cols, int32_t nnz, V alpha, V beta, bool axpby,
               hipStream_t stream, int debug_sync, const CallExtra &extra);

// bytes of temp storage csrmv_call needs under the default tuning (what the size query of the plans' inner calls returns)
uint64_t csrmv_temp_bytes(int32_t rows, int32_t nnz, int32_t value_bytes);

// largest rows + nnz one call accepts: 2^31 minus room for the chunk arithmetic of the
// vectorised staging, which indexes up to one tile (+ one chunk row) past the last item in int32
constexpr long long MAX_ITEMS = 0x7fffffffLL - 65536;

static inline uint64_t align256(uint64_t v) { return (v + 255) & ~uint64_t(255); }

}  // namespace mspmv
