// What a persistent (whole-PCG-solve) slab kernel would pay per iteration for its two synchronisations:
//   sum    -- every workgroup contributes K = 4 double partial sums and needs the grid-wide totals (alpha / beta of the iteration): flag-in-data words
//             (payload half + tag in one 8-byte relaxed agent-scope store; no release / acquire, no L2 write-back), every workgroup reads every slot and
//             adds in slot order (the same bits everywhere); "tree" = groups of 16 workgroups first, then the 16 group totals
//   halo   -- every workgroup of a 16 x 16 tile grid hands its edge rows (256 px x 3 floats up and down) and edge columns (32 px x 3 floats left and right)
//             to its neighbours' inboxes as tagged words and waits for its own inbox
//   both   -- one of each per round, the shape of an on-chip iteration
// Every wait is bounded (wall clock): a protocol error ends the run with a message instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gridsync tools/microbench_gridsync.hip && /tmp/gridsync [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int K = 4, NW = 2 * K, BLOCK = 256;
constexpr long long kTimeoutTicks = 100LL * 1000 * 1000;     // 1 s at 100 MHz

struct Sync {
    u64* slots;        // [2][G][NW]
    u64* groupSlots;   // [2][G/16][NW]
    u64* inbox;        // [2][G][4 sides][768 words]
    int* bad;
    int G, tilesX, tilesY;
};

__device__ __forceinline__ u64 word(unsigned tag, unsigned half) { return ((u64)tag << 32) | half; }

// waits for `src` to carry `tag`; returns the payload.  After a time-out every later wait of the run falls through at once.
__device__ __forceinline__ unsigned awaitWord(const u64* src, unsigned tag, int* bad) {
    u64 v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(v >> 32) == tag) return (unsigned)v;
    const long long t0 = wall_clock64();
    while ((unsigned)((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (wall_clock64() - t0 > kTimeoutTicks) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    return (unsigned)v;
}

__device__ __forceinline__ double joinHalves(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((u64)hi << 32) | lo)); }

// flat: G x NW words read by every workgroup
__device__ void sumFlat(const Sync& S, unsigned tag, const double (&mine)[K], double (&total)[K], unsigned* lds) {
    const int tid = threadIdx.x, g = blockIdx.x;
    u64* buf = S.slots + (size_t)(tag & 1) * S.G * NW;
    if (tid < NW) {
        const u64 bits = (u64)__double_as_longlong(mine[tid >> 1]);
        __hip_atomic_store(buf + (size_t)g * NW + tid, word(tag, (unsigned)(bits >> ((tid & 1) * 32))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < S.G * NW; i += BLOCK) lds[i] = awaitWord(buf + i, tag, S.bad);
    __syncthreads();
    if (tid < K) {
        double t = 0;
        for (int r = 0; r < S.G; ++r) t += joinHalves(lds[r * NW + 2 * tid], lds[r * NW + 2 * tid + 1]);
        reinterpret_cast<double*>(lds + S.G * NW)[tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) total[k] = reinterpret_cast<double*>(lds + S.G * NW)[k];
    __syncthreads();
}

// tree: groups of 16 workgroups, the first of each group posts the group total
__device__ void sumTree(const Sync& S, unsigned tag, const double (&mine)[K], double (&total)[K], unsigned* lds) {
    const int tid = threadIdx.x, g = blockIdx.x, grp = g >> 4, nGroups = S.G >> 4;
    u64* buf = S.slots + (size_t)(tag & 1) * S.G * NW;
    u64* top = S.groupSlots + (size_t)(tag & 1) * nGroups * NW;
    double* sums = reinterpret_cast<double*>(lds + 16 * NW + nGroups * NW);
    if (tid < NW) {
        const u64 bits = (u64)__double_as_longlong(mine[tid >> 1]);
        __hip_atomic_store(buf + (size_t)g * NW + tid, word(tag, (unsigned)(bits >> ((tid & 1) * 32))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((g & 15) == 0) {                                  // group leader: 16 x NW words -> group total -> top slot
        if (tid < 16 * NW) lds[tid] = awaitWord(buf + (size_t)grp * 16 * NW + tid, tag, S.bad);
        __syncthreads();
        if (tid < K) {
            double t = 0;
            for (int r = 0; r < 16; ++r) t += joinHalves(lds[r * NW + 2 * tid], lds[r * NW + 2 * tid + 1]);
            sums[tid] = t;
        }
        __syncthreads();
        if (tid < NW) {
            const u64 bits = (u64)__double_as_longlong(sums[tid >> 1]);
            __hip_atomic_store(top + (size_t)grp * NW + tid, word(tag, (unsigned)(bits >> ((tid & 1) * 32))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < nGroups * NW) lds[16 * NW + tid] = awaitWord(top + tid, tag, S.bad);
    __syncthreads();
    if (tid < K) {
        double t = 0;
        for (int r = 0; r < nGroups; ++r) t += joinHalves(lds[16 * NW + r * NW + 2 * tid], lds[16 * NW + r * NW + 2 * tid + 1]);
        sums[tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) total[k] = sums[k];
    __syncthreads();
}

constexpr int kSideWords = 768;      // 256 px x 3 floats (rows); columns use the first 96
__device__ float haloRound(const Sync& S, unsigned tag, float seed) {
    const int tid = threadIdx.x, g = blockIdx.x, tx = g % S.tilesX, ty = g / S.tilesX;
    u64* box = S.inbox + (size_t)(tag & 1) * S.G * 4 * kSideWords;
    // side 0: from above, 1: from below, 2: from the left, 3: from the right (named by where the data comes from, seen from the receiver)
    const int nb[4] = {ty + 1 < S.tilesY ? g + S.tilesX : -1, ty > 0 ? g - S.tilesX : -1, tx + 1 < S.tilesX ? g + 1 : -1, tx > 0 ? g - 1 : -1};
    const int count[4] = {kSideWords, kSideWords, 96, 96};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (nb[s] < 0) continue;
        u64* dst = box + ((size_t)nb[s] * 4 + s) * kSideWords;
        for (int i = tid; i < count[s]; i += BLOCK) __hip_atomic_store(dst + i, word(tag, __float_as_uint(seed + i)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int from[4] = {ty > 0 ? 1 : 0, ty + 1 < S.tilesY ? 1 : 0, tx > 0 ? 1 : 0, tx + 1 < S.tilesX ? 1 : 0};
    float acc = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (!from[s]) continue;
        const u64* src = box + ((size_t)g * 4 + s) * kSideWords;
        for (int i = tid; i < count[s]; i += BLOCK) acc += __uint_as_float(awaitWord(src + i, tag, S.bad));
    }
    return acc;
}

template <int MODE>      // 0 flat sum, 1 tree sum, 2 halo, 3 tree sum + halo, 4 flat sum + halo
__global__ __launch_bounds__(BLOCK) void k_rounds(Sync S, int rounds, unsigned tag0, double* out) {
    extern __shared__ unsigned lds[];
    double mine[K], total[K] = {0, 0, 0, 0};
    float h = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned tag = tag0 + r;
#pragma unroll
        for (int k = 0; k < K; ++k) mine[k] = (double)(blockIdx.x + 1) * (k + 1) + (total[k] > 1e300 ? 1.0 : 0.0);     // depends on the previous round's result
        if (MODE == 0 || MODE == 4) sumFlat(S, tag, mine, total, lds);
        if (MODE == 1 || MODE == 3) sumTree(S, tag, mine, total, lds);
        if (MODE >= 2) h += haloRound(S, tag, (float)total[0]);
    }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = total[K - 1]; out[blockIdx.x * 2 + 1] = h; }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int G = 256, tilesX = 16, tilesY = 16;
    printf("%s, %d CUs; %d workgroups of %d threads, %d rounds per launch\n", prop.name, prop.multiProcessorCount, G, BLOCK, rounds);
    Sync S{};
    S.G = G; S.tilesX = tilesX; S.tilesY = tilesY;
    CHECK(hipMalloc(&S.slots, sizeof(u64) * 2 * G * NW));
    CHECK(hipMalloc(&S.groupSlots, sizeof(u64) * 2 * (G / 16) * NW));
    CHECK(hipMalloc(&S.inbox, sizeof(u64) * 2 * G * 4 * kSideWords));
    CHECK(hipMalloc(&S.bad, sizeof(int)));
    CHECK(hipMemset(S.slots, 0, sizeof(u64) * 2 * G * NW));
    CHECK(hipMemset(S.groupSlots, 0, sizeof(u64) * 2 * (G / 16) * NW));
    CHECK(hipMemset(S.inbox, 0, sizeof(u64) * 2 * G * 4 * kSideWords));
    CHECK(hipMemset(S.bad, 0, sizeof(int)));
    double* out;
    CHECK(hipMalloc(&out, sizeof(double) * 2 * G));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t ldsBytes = sizeof(unsigned) * (G * NW) + 64;
    const char* names[5] = {"sum, flat (every workgroup reads 256 x 8 words)", "sum, tree (16 groups of 16)", "halo (16 x 16 tiles, 6 KB rows + 0.75 KB columns)", "tree sum + halo", "flat sum + halo"};
    unsigned tag = 1;
    const double expect = 4.0 * G * (G + 1) / 2;     // total of k = 3: sum over g of (g + 1) * 4
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            CHECK(hipEventRecord(e0));
            switch (mode) {
                case 0: k_rounds<0><<<G, BLOCK, ldsBytes>>>(S, rounds, tag, out); break;
                case 1: k_rounds<1><<<G, BLOCK, ldsBytes>>>(S, rounds, tag, out); break;
                case 2: k_rounds<2><<<G, BLOCK, ldsBytes>>>(S, rounds, tag, out); break;
                case 3: k_rounds<3><<<G, BLOCK, ldsBytes>>>(S, rounds, tag, out); break;
                default: k_rounds<4><<<G, BLOCK, ldsBytes>>>(S, rounds, tag, out); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            tag += rounds + (rounds & 1);           // keep the parity of the first tag: both buffers hold older tags only
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            int bad = 0;
            std::vector<double> h(2 * G);
            CHECK(hipMemcpy(&bad, S.bad, sizeof(int), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(h.data(), out, sizeof(double) * 2 * G, hipMemcpyDeviceToHost));
            bool same = true;
            for (int g = 0; g < G; ++g) same = same && (mode == 2 || h[2 * g] == expect);
            printf("rep %d  %-62s %7.2f us per round%s%s\n", rep, names[mode], 1e3 * ms / rounds, bad ? "  TIMED OUT" : "", same ? "" : "  WRONG TOTAL");
            if (bad) return 2;
        }
    return 0;
}
