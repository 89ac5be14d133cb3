// Grid-wide hand-off primitives of the persistent ("on-chip") solver kernels: iw_onchip.h (image_warping), sfs_onchip.h, stencil_onchip.h.
//
// Everything that crosses workgroups inside such a kernel travels as naturally aligned 8-byte words {payload, tag}: ONE relaxed agent-scope store each (global_store sc1:
// written through, no fence, no cache write-back) and relaxed agent-scope loads on the polling side (MI355X_MICROARCH.md "handoff-1to1" / granule "R2"; measured in
// tools/microbench_gridsync.hip).  The tag is the iteration (phase) number, so a word says by itself whether it is the one the reader waits for; tags never repeat
// over the life of a buffer.  Every wait is bounded by the device's 100 MHz wall clock and gives up as soon as another waiter has (the `bad` word).
#pragma once
#include "common.h"

namespace optamd {
namespace {

typedef unsigned long long oc_u64;

// Bounds of the waits inside a persistent kernel, in ticks of the device's 100 MHz wall clock.  `first`: the waits of the first phase -- every workgroup posts its words
// before it waits, so passing them proves the whole grid resident; a grid that is NOT co-resident (a foreign tenant holds CUs, another plan's persistent kernel) gives up
// there after 10 ms, before anything has been written, and the solver redoes the step on the streaming kernels.  `later`: once resident, a word is microseconds away; the
// bound only ends a hang (a peer that faulted) and scales with the solve: 100 ms + 100 us per PCG iteration.  Kernels whose first sum waits for OTHER PROCESSES' launches
// (row slabs: the rank hop) keep 2 s for both.  OPT_AMD_ONCHIP_TIMEOUT_MS (tests of the time-out path) overrides `later` and caps `first`.
struct OcTimeouts { long long first, later; };
inline OcTimeouts ocTimeouts(long long overrideTicks, int lIterations, bool waitsForOtherProcesses) {
    OcTimeouts t;
    t.later = waitsForOtherProcesses ? 2000LL * 100000 : (100LL + lIterations / 10) * 100000;
    t.first = waitsForOtherProcesses ? t.later : 10LL * 100000;
    if (overrideTicks > 0) { t.later = overrideTicks; t.first = std::min(t.first, overrideTicks); }
    return t;
}

// SYS: words that cross GPUs (the peer window: uncached memory, system scope); else agent scope
template <bool SYS = false> __device__ __forceinline__ oc_u64 ocLoad(const oc_u64* p) {
    return SYS ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SYS = false> __device__ __forceinline__ void ocStore(oc_u64* p, unsigned tag, unsigned half) {
    if (SYS) __hip_atomic_store(p, ((oc_u64)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(p, ((oc_u64)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Waits until *src carries `tag`; returns the payload.  Bounded: after timeoutTicks of the 100 MHz wall clock -- or as soon as another waiter has given up --
// the wait falls through with whatever is there (the caller's loop ends at its next sum).
template <bool SYS = false> __device__ __forceinline__ unsigned ocAwait(const oc_u64* src, unsigned tag, int* bad, long long timeoutTicks) {
    oc_u64 v = ocLoad<SYS>(src);
    if ((unsigned)(v >> 32) != tag) {
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            v = ocLoad<SYS>(src);
            if ((unsigned)(v >> 32) == tag) break;
            if ((++spins & 31u) == 0) {
                if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (wall_clock64() - t0 > timeoutTicks) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    return (unsigned)v;
}
__device__ __forceinline__ double ocJoin(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((oc_u64)hi << 32) | lo)); }

// Sum over the wave, valid in lane 63: prefix sums inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then row 0 -> 1 and 2 -> 3 (row_bcast:15), then rows 0-1 -> 2-3
// (row_bcast:31).  13 DPP moves + 6 adds per double on the VALU, against six ds_bpermute round trips for the __shfl_down tree (1.2 us per iteration for four sums).
template <int CTRL, int ROWMASK> __device__ __forceinline__ double ocDppAdd(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ocWaveSum63(double v) {
    v = ocDppAdd<0x111, 0xf>(v); v = ocDppAdd<0x112, 0xf>(v); v = ocDppAdd<0x114, 0xf>(v); v = ocDppAdd<0x118, 0xf>(v);      // lane 15 of every row: the row's sum
    v = ocDppAdd<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = ocDppAdd<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

}  // namespace
}  // namespace optamd
