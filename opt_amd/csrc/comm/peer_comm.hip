// "peer" implementation of OptAmd_SlabComm: peer-mapped mailboxes over xGMI instead of RCCL collectives.
//
// Why: the PCG loop's inter-GPU traffic is one sum of four doubles per iteration plus a few hundred KiB of edge rows every
// 7th iteration.  At 8 slabs of 4096^2 the streaming iteration kernel takes ~29 us (and the on-chip solve ~15 us per iteration
// with no launch at all between iterations: peerOnChipPlan), so the loop is bound by the latency of whatever sits between two launches; an ncclAllReduce of 32 bytes costs a kernel launch plus a multi-hop
// protocol.  Here every rank maps every other rank's window (hipIpc handles, one process per GPU) and
//   * all-reduce = ONE small kernel: sum this rank's per-workgroup partials, store the totals into every peer's mailbox slot
//     (direct peer stores over the pair's own xGMI link) as 8-byte words that each carry 4 bytes of payload and the all-reduce's
//     sequence number -- no separate flag, no fence --, poll the local mailbox until every rank's words carry that number, add
//     the contributions in rank order (deterministic, same bits on every rank);
//   * halo exchange = two kernels: push my edge rows into the neighbours' staging buffers + publish, then wait for the
//     neighbours' rows in my own staging buffers and copy them into the ghost rows + acknowledge (double-buffered staging).
// No host involvement inside the loop.  Every spin carries a wall-clock timeout that raises an error flag in pinned host
// memory instead of hanging the GPU; the host side checks the flag at every call.
//
// The reference has no multi-GPU path; this is north_star work.
#include "../../../include/OptAmd.h"
#include "../common.h"      // pollMailSums: the consumer side of a posted all-reduce, as the iteration kernels run it (self-test only)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

// Set-up and self-test: a HIP error there is reported and makes the entry point fail (the launcher falls back to RCCL); nothing in this library calls exit().
#define CK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "OptComm(peer): HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 0; } } while (0)
// Inside the communicator callbacks (which return nothing): record the error on the context; every later callback is then a no-op and the owner of the communicator
// finds the code with OptComm_PeerError (bench.py prints a JSON error line and exits non-zero; opt_amd.slab.SlabJob.close raises).
#define CK_HIP_CB(x, X) do { hipError_t e_ = (X); if (e_ != hipSuccess) { fprintf(stderr, "OptComm(peer) rank %d: HIP error %s at %s:%d\n", (x)->rank, hipGetErrorString(e_), __FILE__, __LINE__); fail((x), 7); return; } } while (0)

namespace {

constexpr int kMaxWorld = 16;
constexpr int kSlots = 4;          // mailbox slots (2 would do: a rank can be at most one all-reduce ahead of any peer)
constexpr int kMaxVals = 8;        // doubles per all-reduce
constexpr int kStageDepth = 2;
typedef unsigned long long u64;

// One per rank, in device memory of that rank, mapped by every peer.  Only 8-byte words are used for signalling.
struct Window {
    // All-reduce mailbox, "flag-in-data" (the idea of RCCL's LL protocol): every 8-byte word carries 4 bytes of payload and the low 32 bits of the
    // all-reduce's sequence number, and 8-byte stores are atomic -- so a contribution needs no separate flag, no fence between payload and flag, and
    // the receiver polls the payload words themselves.  A double travels as two words.  [slot][source rank][2 * value + half]
    u64 ll[kSlots][kMaxWorld][2 * kMaxVals];
    u64 haloSeq[2];                            // [0]: last exchange pushed by the rank above, [1]: by the rank below
    u64 haloAck[2];                            // [0]: last exchange of MINE the rank above has consumed, [1]: the rank below
    u64 pad[4];
    // followed by staging[2 sides][kStageDepth][stageBytes], then the edge boxes of the on-chip linear solve: [2 sides: 0 = written by the rank above, 1 = by the rank
    // below][2 parities][kEdgeWords] tagged words
};
constexpr size_t kEdgeWords = 1 << 17;      // per (side, parity): 1 MiB -- 170 tiles of 768 words (float), 85 of 1536 (double)
constexpr size_t kEdgeBytes = 2 * 2 * kEdgeWords * sizeof(u64);

struct PeerCtx {
    int rank, world;
    int said = 0;                  // the error-state message has been printed for THIS communicator
    Window* win[kMaxWorld];        // win[rank] = my own window, others IPC-mapped
    char* stage[kMaxWorld];        // staging area behind each window
    u64* edge[kMaxWorld];          // edge boxes behind each staging area
    size_t stageBytes;             // per side and depth
    void* base;                    // my allocation
    hipIpcMemHandle_t handle;
    u64 arSeq, haloSeq;
    unsigned int* dCounter;        // last-block counters of the copy kernels
    volatile int* hostErr;         // pinned, device-visible
    long long timeoutTicks;        // wall_clock64 ticks (100 MHz)
    int memKind;                   // 3 uncached, 1 fine-grained, 0 plain hipMalloc
    bool sharedDevice;             // some other rank's window lives on this rank's GPU (tests, bench.py --share-gpu)
    char busId[32];                // PCI bus id of this rank's GPU, exchanged with the IPC handle
    // optional hipEvent timing of the communicator's own kernels (OptComm_PeerSetTiming): {all-reduce, halo} x {count, pairs}
    bool timing; std::vector<std::pair<hipEvent_t, hipEvent_t>> evAr, evHalo; double msAr, msHalo; long nAr, nHalo;
    OptAmd_SlabComm api;
    OptAmd_SlabCommExt ext;
};

__device__ __forceinline__ u64 ldSys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void stSys(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double ldSysD(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void stSysD(double* p, double v) { __hip_atomic_store((u64*)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// spin until *p >= want or the wall clock runs out; returns false on timeout
__device__ __forceinline__ bool waitAtLeast(const u64* p, u64 want, long long timeoutTicks) {
    if (ldSys(p) >= want) return true;
    const long long t0 = wall_clock64();
    while (ldSys(p) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeoutTicks) return false;
    }
    return true;
}

struct Peers { Window* win[kMaxWorld]; };

struct PartialsIn { const double* p[kMaxVals]; int n[kMaxVals]; };

// ---- all-reduce: one workgroup of 256 threads --------------------------------------------------------------------------------------
// `in` != nullptr semantics: if parts.p[i] is set, value i = sum of parts.n[i] partials (fixed order); else value i = buf[i].
__global__ __launch_bounds__(256) void k_mailAllReduce(double* __restrict__ buf, PartialsIn parts, int usePartials, int n, Peers P, int rank, int world, u64 seq,
                                                       long long timeoutTicks, volatile int* hostErr) {
    __shared__ double vals[kMaxVals];
    __shared__ double wsum[kMaxVals][4];
    __shared__ int timedOut;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) timedOut = 0;      // (ordered before its use by the barriers of the summation phase below)
    if (usePartials) {      // all n arrays in one pass: the loads (L2 misses on what the previous kernel just wrote) are in flight together, one barrier
        double t[kMaxVals];
#pragma unroll
        for (int i = 0; i < kMaxVals; ++i) { t[i] = 0; if (i < n) for (int k = tid; k < parts.n[i]; k += 256) t[i] += parts.p[i][k]; }
#pragma unroll
        for (int i = 0; i < kMaxVals; ++i) {
            for (int off = 32; off > 0; off >>= 1) t[i] += __shfl_down(t[i], off, 64);
            if (lane == 0) wsum[i][wave] = t[i];
        }
        __syncthreads();
        if (tid < n) vals[tid] = ((wsum[tid][0] + wsum[tid][1]) + wsum[tid][2]) + wsum[tid][3];
        __syncthreads();
    } else {
        if (tid < n) vals[tid] = buf[tid];
        __syncthreads();
    }
    const int slot = (int)(seq % kSlots);
    const unsigned tag = (unsigned)seq;
    // post: thread (peer t, word w) stores one tagged word into peer t's window -- fire and forget, no fence
    const int nw = 2 * n;
    for (int j = tid; j < world * nw; j += 256) {
        const int t = j / nw, w = j % nw;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(vals[w >> 1]);
        const unsigned half = (w & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(&P.win[t]->ll[slot][rank][w], ((u64)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // receive: thread (source rank r, word w) polls its word in this rank's own window until it carries this all-reduce's tag
    __shared__ unsigned halves[kMaxWorld][2 * kMaxVals];
    for (int j = tid; j < world * nw; j += 256) {
        const int r = j / nw, w = j % nw;
        const u64* src = &P.win[rank]->ll[slot][r][w];
        u64 v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(v >> 32) != tag) {
            const long long t0 = wall_clock64();
            while ((unsigned)((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) != tag) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > timeoutTicks) { *hostErr = 1; timedOut = 1; break; }
            }
        }
        halves[r][w] = (unsigned)v;
    }
    __syncthreads();
    if (tid < n) {
        double t = 0;
        for (int r = 0; r < world; ++r)      // rank order: identical bits on every rank
            t += __longlong_as_double((long long)(((unsigned long long)halves[r][2 * tid + 1] << 32) | halves[r][2 * tid]));
        // a peer never answered: the sum is poisoned rather than left looking like a result (the host sees the flag at its next call, SlabJob.close() at the latest)
        buf[tid] = timedOut ? __longlong_as_double(0x7ff8000000000000ll) : t;
    }
}

// ---- post-only all-reduce (OptAmd_SlabCommExt.allReducePost): the first half of k_mailAllReduce -- sum this rank's partials, store the tagged words into
// every rank's mailbox (this rank's included) -- and nothing else.  The consumer (the next PCG iteration kernel's prologue) polls the words in its own
// window, so its launch and first loads overlap the flight of the contributions instead of following a kernel that waited for them.
__global__ __launch_bounds__(256) void k_mailPost(PartialsIn parts, int n, Peers P, int rank, int world, u64 seq) {
    __shared__ double vals[kMaxVals];
    __shared__ double wsum[kMaxVals][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double t[kMaxVals];
#pragma unroll
    for (int i = 0; i < kMaxVals; ++i) { t[i] = 0; if (i < n) for (int k = tid; k < parts.n[i]; k += 256) t[i] += parts.p[i][k]; }
#pragma unroll
    for (int i = 0; i < kMaxVals; ++i) {
        for (int off = 32; off > 0; off >>= 1) t[i] += __shfl_down(t[i], off, 64);
        if (lane == 0) wsum[i][wave] = t[i];
    }
    __syncthreads();
    if (tid < n) vals[tid] = ((wsum[tid][0] + wsum[tid][1]) + wsum[tid][2]) + wsum[tid][3];      // the same order as k_mailAllReduce: the same bits
    __syncthreads();
    const int slot = (int)(seq % kSlots);
    const unsigned tag = (unsigned)seq;
    const int nw = 2 * n;
    for (int j = tid; j < world * nw; j += 256) {
        const int tr = j / nw, w = j % nw;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(vals[w >> 1]);
        const unsigned half = (w & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(&P.win[tr]->ll[slot][rank][w], ((u64)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// self-test of the posted path: what an iteration kernel's prologue does with a MailRef
__global__ __launch_bounds__(256) void k_mailPollTest(optamd::MailRefDev M, double* __restrict__ out) {
    __shared__ double scr[4 + 1 + 64];
    double o4[4];
    optamd::pollMailSums<4>(M, scr, o4);
    if (threadIdx.x < 4) out[threadIdx.x] = o4[threadIdx.x];
}

// ---- halo exchange -------------------------------------------------------------------------------------------------------------------
struct HaloArgs {
    int nb;
    const char* sendUp[8]; const char* sendDown[8]; char* recvUp[8]; char* recvDown[8];
    long bytes[8], offset[8];      // offset of buffer k inside a staging block
};
__device__ __forceinline__ void copyBytes(char* dst, const char* src, long bytes, int tid, int nthreads) {   // all pointers and sizes are multiples of 4 B; 16 B when aligned
    if ((((size_t)dst | (size_t)src | (size_t)bytes) & 15) == 0) {
        const uint4* s = (const uint4*)src; uint4* d = (uint4*)dst;
        for (long i = tid; i < bytes / 16; i += nthreads) d[i] = s[i];
    } else {
        const unsigned* s = (const unsigned*)src; unsigned* d = (unsigned*)dst;
        for (long i = tid; i < bytes / 4; i += nthreads) d[i] = s[i];
    }
}
// push: my first owned rows -> the staging block "from below" of the rank above; my last owned rows -> "from above" of the rank below
__global__ __launch_bounds__(256) void k_haloPush(HaloArgs H, Peers P, char* stageUp, char* stageDown, int rank, int world, u64 seq, unsigned int* counter,
                                                  long long timeoutTicks, volatile int* hostErr) {
    const bool up = rank > 0, down = rank < world - 1;
    // the staging block of exchange `seq` was last used by exchange seq - kStageDepth: the neighbour must have consumed that one
    if (threadIdx.x == 0 && seq > kStageDepth) {
        if (up && !waitAtLeast(&P.win[rank]->haloAck[0], seq - kStageDepth, timeoutTicks)) *hostErr = 2;
        if (down && !waitAtLeast(&P.win[rank]->haloAck[1], seq - kStageDepth, timeoutTicks)) *hostErr = 2;
    }
    __syncthreads();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int k = 0; k < H.nb; ++k) {
        if (up) copyBytes(stageUp + H.offset[k], H.sendUp[k], H.bytes[k], tid, nt);
        if (down) copyBytes(stageDown + H.offset[k], H.sendDown[k], H.bytes[k], tid, nt);
    }
    __threadfence_system();
    __syncthreads();
    __shared__ bool last;
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        *counter = 0;
        __threadfence_system();
        if (up) stSys(&P.win[rank - 1]->haloSeq[1], seq);        // I am the rank below my upper neighbour
        if (down) stSys(&P.win[rank + 1]->haloSeq[0], seq);
    }
}
// pull: wait for the neighbours' rows in my staging blocks, copy them into the ghost rows, acknowledge
__global__ __launch_bounds__(256) void k_haloPull(HaloArgs H, Peers P, const char* fromUp, const char* fromDown, int rank, int world, u64 seq, unsigned int* counter,
                                                  long long timeoutTicks, volatile int* hostErr) {
    const bool up = rank > 0, down = rank < world - 1;
    if (threadIdx.x == 0) {
        if (up && !waitAtLeast(&P.win[rank]->haloSeq[0], seq, timeoutTicks)) *hostErr = 3;
        if (down && !waitAtLeast(&P.win[rank]->haloSeq[1], seq, timeoutTicks)) *hostErr = 3;
    }
    __syncthreads();
    __threadfence_system();      // every thread acquires what thread 0 waited for
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int k = 0; k < H.nb; ++k) {
        if (up) copyBytes(H.recvUp[k], fromUp + H.offset[k], H.bytes[k], tid, nt);
        if (down) copyBytes(H.recvDown[k], fromDown + H.offset[k], H.bytes[k], tid, nt);
    }
    __threadfence();
    __syncthreads();
    __shared__ bool last;
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        *counter = 0;
        if (up) stSys(&P.win[rank - 1]->haloAck[1], seq);
        if (down) stSys(&P.win[rank + 1]->haloAck[0], seq);
    }
}

// Error codes (OptComm_PeerError): 1 all-reduce timed out, 2 halo acknowledgement, 3 halo rows, 4 a posted all-reduce polled by an iteration kernel,
// 5 an exchange larger than the staging area, 6 more values / buffers than a call supports, 7 a HIP error inside a callback.  Sticky: once set, every
// callback returns at once (the sums of that run are garbage either way) and the owner reports it -- the library never ends the process.
void fail(PeerCtx* x, int code) { if (!*x->hostErr) *x->hostErr = code; }
bool failed(PeerCtx* x, const char* where) {
    const int e = *x->hostErr;
    if (!e) return false;
    if (!x->said++) fprintf(stderr, "OptComm(peer) rank %d: communicator in error state at %s (code %d: 1 = all-reduce timed out, 2 = halo ack, 3 = halo rows, 4 = posted all-reduce polled by the "
                                  "iteration kernel, 5 = oversize exchange, 6 = too many values, 7 = HIP error) -- a rank died or fell out of step; further collectives are skipped\n", x->rank, where, e);
    return true;
}
struct ScopedEv {      // brackets a callback's launches with an event pair when timing is on
    PeerCtx* x; std::vector<std::pair<hipEvent_t, hipEvent_t>>* v; hipStream_t s; hipEvent_t b = nullptr;
    ScopedEv(PeerCtx* x_, std::vector<std::pair<hipEvent_t, hipEvent_t>>* v_, hipStream_t s_) : x(x_), v(v_), s(s_) {
        if (!x->timing) return;
        // bounded: with timing left on for a long run the oldest pairs (long since completed) are folded into the totals instead of accumulating
        if (v->size() >= 4096) {
            double& ms = (v == &x->evAr) ? x->msAr : x->msHalo; long& n = (v == &x->evAr) ? x->nAr : x->nHalo;
            for (size_t i = 0; i < 2048; ++i) {
                auto& p = (*v)[i]; float t = 0;
                if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) { ms += t; ++n; }
                (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
            }
            v->erase(v->begin(), v->begin() + 2048);
        }
        hipEvent_t a; if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { b = nullptr; return; }
        (void)hipEventRecord(a, s); v->push_back({a, b});
    }
    ~ScopedEv() { if (b) (void)hipEventRecord(b, s); }
};
Peers peersOf(PeerCtx* x) { Peers P; for (int r = 0; r < kMaxWorld; ++r) P.win[r] = x->win[r]; return P; }

void peerAllReduceImpl(PeerCtx* x, double* buf, const PartialsIn* parts, int n, hipStream_t s) {
    if (failed(x, "allReduce")) return;
    if (n > kMaxVals) { fprintf(stderr, "OptComm(peer): all-reduce of %d > %d doubles\n", n, kMaxVals); fail(x, 6); return; }
    PartialsIn pin{}; if (parts) pin = *parts;
    ++x->arSeq;
    ScopedEv ev(x, &x->evAr, s);
    k_mailAllReduce<<<1, 256, 0, s>>>(buf, pin, parts ? 1 : 0, n, peersOf(x), x->rank, x->world, x->arSeq, x->timeoutTicks, x->hostErr);
    CK_HIP_CB(x, hipGetLastError());
}
void peerAllReduce(void* c, double* buf, int n, void* stream) { peerAllReduceImpl((PeerCtx*)c, buf, nullptr, n, (hipStream_t)stream); }
void peerAllReducePartials(void* c, const double* const* parts, const int* counts, int n, double* out, void* stream) {
    PartialsIn pin{};
    for (int i = 0; i < n && i < kMaxVals; ++i) { pin.p[i] = parts[i]; pin.n[i] = counts[i]; }
    peerAllReduceImpl((PeerCtx*)c, out, &pin, n, (hipStream_t)stream);
}
int peerAllReducePost(void* c, const double* const* parts, const int* counts, int n, OptAmd_MailRef* ref, void* stream) {
    auto* x = (PeerCtx*)c;
    if (failed(x, "allReducePost")) return 0;
    if (n > kMaxVals || !ref) return 0;
    PartialsIn pin{};
    for (int i = 0; i < n; ++i) { pin.p[i] = parts[i]; pin.n[i] = counts[i]; }
    const u64 seq = ++x->arSeq;
    {
        ScopedEv ev(x, &x->evAr, (hipStream_t)stream);
        k_mailPost<<<1, 256, 0, (hipStream_t)stream>>>(pin, n, peersOf(x), x->rank, x->world, seq);
    }
    if (hipGetLastError() != hipSuccess) { fail(x, 7); return 0; }
    ref->words = &x->win[x->rank]->ll[seq % kSlots][0][0];
    ref->world = x->world; ref->stride = 2 * kMaxVals; ref->tag = (unsigned)seq; ref->timeoutTicks = x->timeoutTicks; ref->errFlag = (int*)x->hostErr;
    return 1;
}
int peerAllReducePlan(void* c, int n, OptAmd_MailPost* post, OptAmd_MailRef* ref) {
    auto* x = (PeerCtx*)c;
    if (failed(x, "allReducePlan")) return 0;
    if (n > kMaxVals || !post || !ref) return 0;
    const u64 seq = ++x->arSeq;
    const int slot = (int)(seq % kSlots);
    for (int t = 0; t < kMaxWorld; ++t) post->dst[t] = t < x->world ? &x->win[t]->ll[slot][x->rank][0] : nullptr;
    post->world = x->world; post->tag = (unsigned)seq; post->ticket = x->dCounter + 2;
    ref->words = &x->win[x->rank]->ll[slot][0][0];
    ref->world = x->world; ref->stride = 2 * kMaxVals; ref->tag = (unsigned)seq; ref->timeoutTicks = x->timeoutTicks; ref->errFlag = (int*)x->hostErr;
    return 1;
}
// The on-chip linear solve across ranks: `count` all-reduces reserved at once (sequence numbers seq0 .. seq0 + count - 1; the kernel's workgroup 0 posts, every
// workgroup polls -- the mailbox words of k_mailPost / pollMailSums) and the edge boxes behind the staging areas.
int peerOnChipPlan(void* c, int n, int count, int tilesX, long wordsPerTile, OptAmd_OnChipLinks* L) {
    auto* x = (PeerCtx*)c;
    if (failed(x, "onChipPlan")) return 0;
    if (n > kMaxVals || (size_t)tilesX * (size_t)wordsPerTile > kEdgeWords) return 0;
    if (count == 0) return 1;      // dry query: could a plan be made? (part of the ranks' vote on running on chip; nothing is reserved)
    if (count < 1 || !L) return 0;
    const u64 seq0 = x->arSeq + 1;
    x->arSeq += (u64)count;
    for (int t = 0; t < 16; ++t) L->mailDst[t] = t < x->world ? &x->win[t]->ll[0][x->rank][0] : nullptr;
    L->mailMine = &x->win[x->rank]->ll[0][0][0];
    L->world = x->world; L->rank = x->rank; L->slots = kSlots; L->slotStride = kMaxWorld * 2 * kMaxVals; L->rankStride = 2 * kMaxVals;
    L->seq0 = (unsigned)seq0;
    // my top tiles write the LOWER box (side 1: "written by the rank below") of the rank above and read my own upper box (side 0); mirrored for the bottom tiles
    L->edgeSendUp = x->rank > 0 ? x->edge[x->rank - 1] + 1 * 2 * kEdgeWords : nullptr;
    L->edgeRecvUp = x->rank > 0 ? x->edge[x->rank] + 0 * 2 * kEdgeWords : nullptr;
    L->edgeSendDown = x->rank < x->world - 1 ? x->edge[x->rank + 1] + 0 * 2 * kEdgeWords : nullptr;
    L->edgeRecvDown = x->rank < x->world - 1 ? x->edge[x->rank] + 1 * 2 * kEdgeWords : nullptr;
    L->edgeParityStride = (long)kEdgeWords;
    L->timeoutTicks = x->timeoutTicks; L->errFlag = (int*)x->hostErr;
    return 1;
}
void peerHalo(void* c, int nb, const void* const* su, const void* const* sd, void* const* ru, void* const* rd, const long* bytes, void* stream) {
    auto* x = (PeerCtx*)c; hipStream_t s = (hipStream_t)stream;
    if (failed(x, "haloExchange")) return;
    if (x->world == 1) return;
    if (nb > 8) { fprintf(stderr, "OptComm(peer): %d > 8 buffers in one exchange\n", nb); fail(x, 6); return; }
    HaloArgs H{}; H.nb = nb;
    long off = 0;
    for (int k = 0; k < nb; ++k) {
        H.sendUp[k] = (const char*)su[k]; H.sendDown[k] = (const char*)sd[k]; H.recvUp[k] = (char*)ru[k]; H.recvDown[k] = (char*)rd[k];
        H.bytes[k] = bytes[k]; H.offset[k] = off; off += (bytes[k] + 15) / 16 * 16;
    }
    if ((size_t)off > x->stageBytes) { fprintf(stderr, "OptComm(peer): exchange of %ld bytes per side exceeds the staging capacity %zu (OptComm_PeerCreate)\n", off, x->stageBytes); fail(x, 5); return; }
    const u64 seq = ++x->haloSeq;
    const size_t blk = (size_t)(seq % kStageDepth) * x->stageBytes;
    // staging layout behind every window: [side 0 = rows coming from the rank above][side 1 = from the rank below], each kStageDepth blocks
    char* stageUp = x->rank > 0 ? x->stage[x->rank - 1] + (size_t)kStageDepth * x->stageBytes + blk : nullptr;      // I am "below" for the rank above
    char* stageDown = x->rank < x->world - 1 ? x->stage[x->rank + 1] + blk : nullptr;                                // I am "above" for the rank below
    const char* fromUp = x->stage[x->rank] + blk;
    const char* fromDown = x->stage[x->rank] + (size_t)kStageDepth * x->stageBytes + blk;
    const int grid = (int)std::max<long>(1, std::min<long>(64, off / (256 * 16) + 1));
    ScopedEv ev(x, &x->evHalo, s);
    k_haloPush<<<grid, 256, 0, s>>>(H, peersOf(x), stageUp, stageDown, x->rank, x->world, seq, x->dCounter, x->timeoutTicks, x->hostErr);
    k_haloPull<<<grid, 256, 0, s>>>(H, peersOf(x), fromUp, fromDown, x->rank, x->world, seq, x->dCounter + 1, x->timeoutTicks, x->hostErr);
    CK_HIP_CB(x, hipGetLastError());
}

}  // namespace

extern "C" {

// What the ranks all-gather: the IPC handle of the window followed by the PCI bus id of the GPU it lives on (so that a rank can tell that a peer SHARES its GPU).
int OptComm_PeerHandleBytes(void) { return (int)sizeof(hipIpcMemHandle_t) + 32; }
int OptComm_PeerMaxWorld(void) { return kMaxWorld; }

// Phase 1 (every rank, its own GPU current): allocate the window + staging (stageBytes per side and depth), export its IPC handle.
void* OptComm_PeerCreate(int rank, int world, long stageBytes, double timeoutSeconds) {
    if (world > kMaxWorld || rank < 0 || rank >= world) { fprintf(stderr, "OptComm(peer): world %d > %d\n", world, kMaxWorld); return nullptr; }
    auto* x = new PeerCtx();
    x->rank = rank; x->world = world; x->stageBytes = (size_t)((stageBytes + 255) / 256 * 256);
    const size_t total = (sizeof(Window) + 255) / 256 * 256 + 2 * (size_t)kStageDepth * x->stageBytes + kEdgeBytes;
    // uncached device memory: peers' stores and our polling loads bypass the local caches (what RCCL uses for its own flags);
    // fall back to fine-grained, then plain device memory (loads/stores above are system-scope atomics either way)
    x->memKind = -1;
    if (const char* e = getenv("OPT_AMD_PEER_MEM")) x->memKind = atoi(e);
    // (plain hipMalloc memory -- OPT_AMD_PEER_MEM=0, experiments only -- may be cached in a reader's L2 and is not tried by default)
    const int kinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, 0};
    for (int k : kinds) {
        if (x->memKind >= 0 ? k != x->memKind : k == 0) continue;
        hipError_t e = k ? hipExtMallocWithFlags(&x->base, total, k) : hipMalloc(&x->base, total);
        if (e == hipSuccess) {
            if (hipIpcGetMemHandle(&x->handle, x->base) == hipSuccess) { x->memKind = k; break; }
            (void)hipFree(x->base);
        }
        (void)hipGetLastError();
        x->base = nullptr;
    }
    if (!x->base) { fprintf(stderr, "OptComm(peer): could not allocate an IPC-exportable window (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)\n"); delete x; return nullptr; }
    {
        int dev = 0; memset(x->busId, 0, sizeof x->busId);
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(x->busId, (int)sizeof x->busId - 1, dev) != hipSuccess) snprintf(x->busId, sizeof x->busId, "device-%d", dev);
    }
    x->sharedDevice = false; x->timing = false; x->msAr = x->msHalo = 0; x->nAr = x->nHalo = 0;
    CK_HIP(hipMemset(x->base, 0, total));
    CK_HIP(hipDeviceSynchronize());
    x->win[rank] = (Window*)x->base;
    x->stage[rank] = (char*)x->base + (sizeof(Window) + 255) / 256 * 256;
    x->edge[rank] = (u64*)(x->stage[rank] + 2 * (size_t)kStageDepth * x->stageBytes);
    CK_HIP(hipMalloc((void**)&x->dCounter, 4 * sizeof(unsigned int)));      // [0], [1]: last-block counters of the halo kernels; [2]: ticket of in-kernel posts
    CK_HIP(hipMemset(x->dCounter, 0, 4 * sizeof(unsigned int)));
    CK_HIP(hipHostMalloc((void**)&x->hostErr, sizeof(int), hipHostMallocMapped));
    *x->hostErr = 0;
    x->timeoutTicks = (long long)((timeoutSeconds > 0 ? timeoutSeconds : 20.0) * 1e8);       // wall_clock64 runs at 100 MHz
    x->api = OptAmd_SlabComm{x, rank, world, peerHalo, peerAllReduce};
    x->ext = OptAmd_SlabCommExt{};
    x->ext.size = sizeof(OptAmd_SlabCommExt); x->ext.allReducePartials = peerAllReducePartials;
    // The posted all-reduce is polled by EVERY workgroup of the next iteration kernel, which is only safe while all ranks' kernels are co-resident: on one GPU per
    // rank they are; ranks that share a GPU would wait for a post kernel queued behind a peer's full-chip launch.  OptComm_PeerConnect therefore takes it away when it
    // finds a peer on this rank's GPU -- unless OPT_AMD_PEER_POST=1 insists (tests and bench.py --share-gpu, which cap the grids with OPT_AMD_ITER_MAXWG); =0: never.
    if (const char* e = getenv("OPT_AMD_PEER_POST")) { if (atoi(e) != 0) x->ext.allReducePost = peerAllReducePost; }
    else x->ext.allReducePost = peerAllReducePost;
    // allReducePlan (the iteration kernel's last workgroup posts; no kernel of ours between two iterations) is not offered: measured on one GPU
    // (tools/slab_overhead.py, profiles/r03_slab_overhead_posted_allreduce.txt) the device-scope release / acquire around the ticket costs more than the
    // one-workgroup post kernel it removes -- 4096 x 512 slab: 36.6 us per iteration against 34.3 (k_mailPost) and 36.1 (round 2's waiting all-reduce); plain: 30.4.
    // (Development builds, -DOPT_AMD_DEV_SWITCHES: OPT_AMD_PEER_PLAN=1 offers it.)
#ifdef OPT_AMD_DEV_SWITCHES
    if (const char* e = getenv("OPT_AMD_PEER_PLAN")) { if (atoi(e) != 0) x->ext.allReducePlan = peerAllReducePlan; }
#endif
    x->ext.onChipPlan = peerOnChipPlan;      // (taken away like allReducePost when ranks share a device, unless OPT_AMD_PEER_POST=1 says the grids are capped)
    return x;
}
void OptComm_PeerHandle(void* c, char* out) { auto* x = (PeerCtx*)c; memcpy(out, &x->handle, sizeof(hipIpcMemHandle_t)); memcpy(out + sizeof(hipIpcMemHandle_t), x->busId, 32); }
int OptComm_PeerSharesDevice(void* c) { return ((PeerCtx*)c)->sharedDevice ? 1 : 0; }
int OptComm_PeerPosts(void* c) { return ((PeerCtx*)c)->ext.allReducePost ? 1 : 0; }
// hipEvent timing of the communicator's own kernels from now on (totals restart): what an iteration spends in the all-reduce / post kernel and in the halo copies.
void OptComm_PeerSetTiming(void* c, int on) {
    auto* x = (PeerCtx*)c;
    for (auto* v : {&x->evAr, &x->evHalo}) { for (auto& p : *v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); } v->clear(); }
    x->msAr = x->msHalo = 0; x->nAr = x->nHalo = 0; x->timing = on != 0;
}
// out[4] = {all-reduce launches, their total ms, halo exchanges, their total ms} since OptComm_PeerSetTiming; synchronises the recorded events.
void OptComm_PeerTimings(void* c, double* out) {
    auto* x = (PeerCtx*)c;
    auto fold = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double& ms, long& n) {
        for (auto& p : v) { float t = 0; if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) { ms += t; ++n; } (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        v.clear();
    };
    fold(x->evAr, x->msAr, x->nAr); fold(x->evHalo, x->msHalo, x->nHalo);
    out[0] = (double)x->nAr; out[1] = x->msAr; out[2] = (double)x->nHalo; out[3] = x->msHalo;
}
int OptComm_PeerMemKind(void* c) { return ((PeerCtx*)c)->memKind; }
// Phase 2 (after the handles of all ranks were gathered, rank-major): map every peer's window.
int OptComm_PeerConnect(void* c, const char* allHandles) {
    auto* x = (PeerCtx*)c;
    const size_t rec = sizeof(hipIpcMemHandle_t) + 32;
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) continue;
        if (strncmp(allHandles + (size_t)r * rec + sizeof(hipIpcMemHandle_t), x->busId, 32) == 0) x->sharedDevice = true;
        hipIpcMemHandle_t h; memcpy(&h, allHandles + (size_t)r * rec, sizeof(h));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { fprintf(stderr, "OptComm(peer) rank %d: hipIpcOpenMemHandle(rank %d) failed: %s\n", x->rank, r, hipGetErrorString(e)); return 0; }
        x->win[r] = (Window*)p;
        x->stage[r] = (char*)p + (sizeof(Window) + 255) / 256 * 256;
        x->edge[r] = (u64*)(x->stage[r] + 2 * (size_t)kStageDepth * x->stageBytes);
    }
    if (x->sharedDevice) {      // see OptComm_PeerCreate: co-residency of all ranks' kernels is not given on a shared GPU
        const char* e = getenv("OPT_AMD_PEER_POST");
        if (!(e && atoi(e) != 0)) { x->ext.allReducePost = nullptr; x->ext.allReducePlan = nullptr; x->ext.onChipPlan = nullptr; }
    }
    return 1;
}
const OptAmd_SlabComm* OptComm_PeerSlabComm(void* c) { return &((PeerCtx*)c)->api; }
const OptAmd_SlabCommExt* OptComm_PeerSlabCommExt(void* c) { return &((PeerCtx*)c)->ext; }
// One all-reduce and one halo exchange with known answers, run once after OptComm_PeerConnect by every rank (collective).  Returns 1 if this rank
// saw the right values, 0 on a wrong value or a timeout -- it never exits, so the launcher can fall back to RCCL when the peer path does not work
// on a machine (IPC mapping across devices, coherence of the window memory kind, ...).  Uses a short timeout of its own.
int OptComm_PeerSelfTest(void* c, double timeoutSeconds) {
    auto* x = (PeerCtx*)c;
    const long long keep = x->timeoutTicks;
    x->timeoutTicks = (long long)((timeoutSeconds > 0 ? timeoutSeconds : 5.0) * 1e8);
    int ok = 1;
    double* d = nullptr; float* rows = nullptr;
    CK_HIP(hipMalloc((void**)&d, 4 * sizeof(double)));
    const int nf = 4096;                                           // floats per test row block
    CK_HIP(hipMalloc((void**)&rows, 4 * nf * sizeof(float)));      // [recvUp | sendUp | sendDown | recvDown]
    double h[4] = {1.0 + x->rank, 10.0 * (1 + x->rank), -1.0, 0.5};
    CK_HIP(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    PartialsIn pin{};
    ++x->arSeq;
    k_mailAllReduce<<<1, 256, 0, 0>>>(d, pin, 0, 4, peersOf(x), x->rank, x->world, x->arSeq, x->timeoutTicks, x->hostErr);
    CK_HIP(hipDeviceSynchronize());
    CK_HIP(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    const double w = x->world, tri = w * (w + 1) / 2;
    if (*x->hostErr || h[0] != tri || h[1] != 10.0 * tri || h[2] != -w || h[3] != 0.5 * w) ok = 0;
    if (ok && x->world > 1 && (size_t)nf * sizeof(float) <= x->stageBytes) {
        float* hostRows = (float*)malloc(4 * nf * sizeof(float));
        for (int i = 0; i < 4 * nf; ++i) hostRows[i] = (i / nf == 1 || i / nf == 2) ? (float)(100 * x->rank + i / nf) : -7.f;
        CK_HIP(hipMemcpy(rows, hostRows, 4 * nf * sizeof(float), hipMemcpyHostToDevice));
        const void* su[1] = {rows + nf}; const void* sd[1] = {rows + 2 * nf}; void* ru[1] = {rows}; void* rd[1] = {rows + 3 * nf}; const long bytes[1] = {(long)(nf * sizeof(float))};
        // peerHalo would exit on an error flag: run its two kernels by hand
        HaloArgs H{}; H.nb = 1; H.sendUp[0] = (const char*)su[0]; H.sendDown[0] = (const char*)sd[0]; H.recvUp[0] = (char*)ru[0]; H.recvDown[0] = (char*)rd[0]; H.bytes[0] = bytes[0]; H.offset[0] = 0;
        const u64 seq = ++x->haloSeq;
        const size_t blk = (size_t)(seq % kStageDepth) * x->stageBytes;
        char* stageUp = x->rank > 0 ? x->stage[x->rank - 1] + (size_t)kStageDepth * x->stageBytes + blk : nullptr;
        char* stageDown = x->rank < x->world - 1 ? x->stage[x->rank + 1] + blk : nullptr;
        k_haloPush<<<2, 256, 0, 0>>>(H, peersOf(x), stageUp, stageDown, x->rank, x->world, seq, x->dCounter, x->timeoutTicks, x->hostErr);
        k_haloPull<<<2, 256, 0, 0>>>(H, peersOf(x), x->stage[x->rank] + blk, x->stage[x->rank] + (size_t)kStageDepth * x->stageBytes + blk, x->rank, x->world, seq, x->dCounter + 1, x->timeoutTicks, x->hostErr);
        CK_HIP(hipDeviceSynchronize());
        CK_HIP(hipMemcpy(hostRows, rows, 4 * nf * sizeof(float), hipMemcpyDeviceToHost));
        if (*x->hostErr) ok = 0;
        for (int i = 0; i < nf && ok; ++i) {
            if (x->rank > 0 && hostRows[i] != (float)(100 * (x->rank - 1) + 2)) ok = 0;                       // the rank above sent its "down" rows
            if (x->rank < x->world - 1 && hostRows[3 * nf + i] != (float)(100 * (x->rank + 1) + 1)) ok = 0;   // the rank below sent its "up" rows
        }
        free(hostRows);
    }
    // The posted all-reduce (k_mailPost + a consumer that polls this rank's mailbox like an iteration kernel's prologue): a machine on which it does not deliver the
    // right sums keeps the waiting all-reduce (the communicator stays usable), it does not fall back to RCCL.
    if (ok && x->ext.allReducePost) {
        double hp[12], *dp = nullptr, *dout = nullptr;
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) hp[3 * i + k] = (i + 1) * (1.0 + x->rank) + 0.5 * k;
        CK_HIP(hipMalloc((void**)&dp, sizeof hp)); CK_HIP(hipMalloc((void**)&dout, 4 * sizeof(double)));
        CK_HIP(hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice));
        const double* parts[4] = {dp, dp + 3, dp + 6, dp + 9}; const int counts[4] = {3, 3, 3, 3};
        OptAmd_MailRef ref{};
        bool good = peerAllReducePost(x, parts, counts, 4, &ref, nullptr) != 0;
        if (good) {
            k_mailPollTest<<<1, 256, 0, 0>>>(optamd::MailRefDev{ref.words, ref.world, ref.stride, ref.tag, ref.timeoutTicks, ref.errFlag}, dout);
            CK_HIP(hipDeviceSynchronize());
            double got[4]; CK_HIP(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
            const double tri2 = x->world * (x->world + 1.0) / 2.0;
            for (int i = 0; i < 4; ++i) if (got[i] != 3.0 * (i + 1) * tri2 + 1.5 * x->world) good = false;
            if (*x->hostErr) good = false;
        }
        if (!good) {
            fprintf(stderr, "OptComm(peer) rank %d: the posted all-reduce failed its self-test; using the waiting all-reduce\n", x->rank);
            x->ext.allReducePost = nullptr; x->ext.allReducePlan = nullptr;
        }
        (void)hipFree(dp); (void)hipFree(dout);
    }
    *x->hostErr = 0;                 // a failed self-test is reported through the return value, not through the run-time abort
    x->timeoutTicks = keep;
    (void)hipFree(d); (void)hipFree(rows);
    return ok;
}
int OptComm_PeerError(void* c) { return *((PeerCtx*)c)->hostErr; }
// Callers must make sure (barrier) that no peer still uses this rank's window.
void OptComm_PeerDestroy(void* c) {
    auto* x = (PeerCtx*)c;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r) if (r != x->rank && x->win[r]) (void)hipIpcCloseMemHandle(x->win[r]);
    (void)hipFree(x->base); (void)hipFree(x->dCounter); (void)hipHostFree((void*)x->hostErr);
    delete x;
}

}  // extern "C"
