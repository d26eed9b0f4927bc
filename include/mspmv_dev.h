/*
 * mspmv_dev.h -- development-only additions of libmspmv_dev.so (built with -DMSPMV_DEV by
 * `make -C merge_spmv_amd dev`; tools/trace_tiles.py and the tuning sweeps load it through
 * MSPMV_LIB=...).  None of this is in the product library: these kernels exist to time parts of
 * the tile kernel, and one of them deliberately computes WRONG results.
 *
 * Extra mspmv_set_tuning bits:
 *   bit 0        one contiguous tile range per XCD in the persistent form
 *   bits  8..15  persistent form of tile_kernel_vec: resident blocks per CU (grid = that x CUs)
 *   bits 20..23  persistent form: tiles per block
 *   bits 16..18  1 = staging only (wrong results, timing ablation), 6 = per-phase cycle stamps written
 *                to the buffer given to mspmv_dev_set_trace, 7 = the reference's per-thread merge-path
 *                walk inside the vectorised tile instead of flags + segmented scan
 */
#ifndef MSPMV_DEV_H_
#define MSPMV_DEV_H_
#include "mspmv.h"
#ifdef __cplusplus
extern "C" {
#endif
/* device buffer (16 x 8 uint64 per block) receiving the clock64() stamps; NULL turns it off */
int mspmv_dev_set_trace(void *d_buf);
#ifdef __cplusplus
}
#endif
#endif
