// mspmv_compact.hip -- the one-launch kernel behind its COMPACT FRONT END (mspmv_kernels.hpp: compact_front), for small problems
// (up to 2304 tiles of the small shape); ref: the reference's own special case for small problems, dispatch_spmv_orig.cuh:674-679,
// agent_spmv_orig.cuh:867-891.
//
// A translation unit of its own because of HOW it has to be compiled: with LLVM's block placement pass OFF
// (-mllvm -disable-block-placement, Makefile), so that the machine code keeps the order of the source -- compact_front is
// written in the order it should run: the fast lane from the hint request to its s_endpgm in one piece of ~2 KB at the head of
// the kernel, what is rarely needed behind it.  A block of a small problem runs alone on its CU and waits for every stretch of
// instructions it jumps to (the instruction cache does not survive a launch); with the placement pass on, the fast lane's
// reduction is merged with the general body's exit and ends up 20 KB away from its barrier.
#include "mspmv_kernels.hpp"

namespace mspmv {

template <typename V>
hipError_t launch_snap_compact(bool axpby, unsigned grid, size_t dyn_lds, hipStream_t stream, Coord *coords, int *rstart, int num_tiles,
                               const Params<V> &p, Carry<V> *carries, const LookBack &lb, int lean_avg, int tile_map)
{
    constexpr int B = COMPACT_BLOCK, I = COMPACT_IPT;
    if (axpby) return launch_exact(tile_kernel_snap<V, B, I, true, false, true>, dim3(grid), dim3(B), dyn_lds, stream, coords, rstart, num_tiles, tile_map, p, carries, lb, lean_avg);
    return launch_exact(tile_kernel_snap<V, B, I, false, false, true>, dim3(grid), dim3(B), dyn_lds, stream, coords, rstart, num_tiles, tile_map, p, carries, lb, lean_avg);
}
template hipError_t launch_snap_compact<float>(bool, unsigned, size_t, hipStream_t, Coord *, int *, int, const Params<float> &, Carry<float> *, const LookBack &, int, int);
template hipError_t launch_snap_compact<double>(bool, unsigned, size_t, hipStream_t, Coord *, int *, int, const Params<double> &, Carry<double> *, const LookBack &, int, int);

}  // namespace mspmv
