// Pieces shared by the graph kernel sets (energy_graph.hip: curveFitting, ARAP; graph_engine.h: the functor-driven ones).
#pragma once
#include "common.h"
#include "energy.h"

namespace optamd {

// val summed over each contiguous run of equal `key` inside the wave; then one atomic per run.
template <class T>
__device__ __forceinline__ void segmentedAtomicAdd(T* __restrict__ base, long key, T val, bool active) {
    const int lane = threadIdx.x & (kWave - 1);
    const long k = active ? key : -1 - lane;              // inactive lanes get unique keys: never merged
    const long prev = __shfl_up(k, 1, kWave);
    const bool head = (lane == 0) || (prev != k);
    const unsigned long long heads = __ballot(head);
    const unsigned long long above = (lane == kWave - 1) ? 0ull : (heads >> (lane + 1));
    const int runEnd = above ? lane + 1 + __builtin_ctzll(above) : kWave;   // first lane of the next run
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const T other = __shfl_down(val, off, kWave);
        if (lane + off < runEnd) val += other;             // only lanes of my own contiguous run are folded in
    }
    if (active && head) unsafeAtomicAdd(base + key, val);
}
template <class T> __device__ __forceinline__ void plainAtomicAdd(T* addr, T val) { unsafeAtomicAdd(addr, val); }


// workgroups of an edge pass: its partial sums take the upper half of a Reduction, the vertex pass the lower half
inline int edgeGrid(long nE, int cus) { return (int)std::max<long>(1, std::min<long>((nE + kBlock - 1) / kBlock, std::min<long>(kMaxPartials / 2, (long)cus * 8))); }

// volumetric_mesh_deformation is arap_mesh_deformation on the 6-neighbour lattice graph (the same fit and regularisation residuals, volumetric_mesh_deformation.t against
// arap_mesh_deformation.t): this factory (energy_graph.hip) runs it on ARAP's kernel set over a half-edge list generated from the lattice dimensions.
template <class T> EnergyOps<T>* makeVolumetricOnArap(const unsigned* dims);

}  // namespace optamd
