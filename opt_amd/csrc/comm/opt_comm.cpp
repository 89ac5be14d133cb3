// libOptComm.so -- two implementations of OptAmd_SlabComm (include/OptAmd.h) for tiling an image problem over
// several GPUs of one node, one rank per GPU.  The reference has no multi-GPU path at all;
// this is new work sized for MI355X's xGMI fabric: every GPU pair has its own point-to-point link, and the only
// data the PCG loop moves between slabs is one image row per neighbour per vector (tens of KiB) plus two
// 8-byte sums per iteration -- latency-bound, so everything is enqueued on the solver's own stream and never
// touches the host.
//   * "rccl"   : ncclSend/ncclRecv pairs (grouped) for the halo rows, ncclAllReduce for the sums.  One process
//                per GPU; the communicator is created from a unique id that the launcher broadcasts.
//   * "peer"   : peer-mapped mailboxes written by small kernels (peer_comm.hip) -- the default of bench.py --gpus N.
//   * "threads": all ranks are threads of ONE process sharing ONE device (test harness for single-GPU boxes):
//                rows are copied device-to-device, sums go through the host, std::barrier-style rendezvous.
#include "../../../include/OptAmd.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#define CK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "OptComm: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
// RCCL errors inside a callback are recorded on the context (sticky: later callbacks are skipped) and reported through OptComm_RcclError -- a library does not exit().
#define CK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "OptComm(rccl) rank %d: %s at %s:%d\n", ctx_->rank, ncclGetErrorString(r_), __FILE__, __LINE__); ctx_->err = (int)r_; return; } } while (0)

namespace {

// ---- RCCL ------------------------------------------------------------------------------------------------------------
struct RcclCtx {
    ncclComm_t comm;
    int rank, world;
    int err = 0;
    OptAmd_SlabComm api;
};
void rcclHalo(void* c, int nb, const void* const* su, const void* const* sd, void* const* ru, void* const* rd, const long* bytes, void* stream) {
    auto* x = (RcclCtx*)c; hipStream_t s = (hipStream_t)stream;
    RcclCtx* const ctx_ = x;
    if (x->err) return;
    CK_NCCL(ncclGroupStart());
    for (int k = 0; k < nb; ++k) {
        if (x->rank > 0) { CK_NCCL(ncclSend(su[k], bytes[k], ncclChar, x->rank - 1, x->comm, s)); CK_NCCL(ncclRecv(ru[k], bytes[k], ncclChar, x->rank - 1, x->comm, s)); }
        if (x->rank < x->world - 1) { CK_NCCL(ncclSend(sd[k], bytes[k], ncclChar, x->rank + 1, x->comm, s)); CK_NCCL(ncclRecv(rd[k], bytes[k], ncclChar, x->rank + 1, x->comm, s)); }
    }
    CK_NCCL(ncclGroupEnd());
}
void rcclAllReduce(void* c, double* buf, int n, void* stream) {
    auto* x = (RcclCtx*)c;
    RcclCtx* const ctx_ = x;
    if (x->err) return;
    CK_NCCL(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, x->comm, (hipStream_t)stream));
}

// ---- threads (single process, single device) -------------------------------------------------------------------
struct Barrier {
    std::mutex m; std::condition_variable cv; int count = 0, gen = 0, n;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> l(m);
        const int g = gen;
        if (++count == n) { count = 0; ++gen; cv.notify_all(); }
        else cv.wait(l, [&] { return gen != g; });
    }
};
struct ThreadWorld {
    int world;
    Barrier bar;
    std::vector<const void*> sendUp, sendDown;   // published pointers, [rank * 8 + k]
    std::vector<double> sums;                    // [rank * 8 + i]
    explicit ThreadWorld(int w) : world(w), bar(w), sendUp(w * 8), sendDown(w * 8), sums(w * 8) {}
};
struct ThreadCtx {
    ThreadWorld* W; int rank;
    OptAmd_SlabComm api;
};
void thrHalo(void* c, int nb, const void* const* su, const void* const* sd, void* const* ru, void* const* rd, const long* bytes, void* stream) {
    auto* x = (ThreadCtx*)c; hipStream_t s = (hipStream_t)stream;
    CK_HIP(hipStreamSynchronize(s));                      // my rows are final
    for (int k = 0; k < nb; ++k) { x->W->sendUp[x->rank * 8 + k] = su[k]; x->W->sendDown[x->rank * 8 + k] = sd[k]; }
    x->W->bar.wait();
    for (int k = 0; k < nb; ++k) {
        if (x->rank > 0) CK_HIP(hipMemcpyAsync(ru[k], x->W->sendDown[(x->rank - 1) * 8 + k], bytes[k], hipMemcpyDeviceToDevice, s));
        if (x->rank < x->W->world - 1) CK_HIP(hipMemcpyAsync(rd[k], x->W->sendUp[(x->rank + 1) * 8 + k], bytes[k], hipMemcpyDeviceToDevice, s));
    }
    CK_HIP(hipStreamSynchronize(s));
    x->W->bar.wait();                                     // nobody overwrites a row a neighbour is still reading
}
void thrAllReduce(void* c, double* buf, int n, void* stream) {
    auto* x = (ThreadCtx*)c; hipStream_t s = (hipStream_t)stream;
    double h[8];
    CK_HIP(hipMemcpyAsync(h, buf, n * sizeof(double), hipMemcpyDeviceToHost, s));
    CK_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) x->W->sums[x->rank * 8 + i] = h[i];
    x->W->bar.wait();
    for (int i = 0; i < n; ++i) { double t = 0; for (int r = 0; r < x->W->world; ++r) t += x->W->sums[r * 8 + i]; h[i] = t; }   // fixed rank order
    x->W->bar.wait();
    CK_HIP(hipMemcpyAsync(buf, h, n * sizeof(double), hipMemcpyHostToDevice, s));
    CK_HIP(hipStreamSynchronize(s));
}

}  // namespace

extern "C" {

int OptComm_UniqueIdBytes(void) { return (int)sizeof(ncclUniqueId); }
int OptComm_GetUniqueId(char* out) { ncclUniqueId id; if (ncclGetUniqueId(&id) != ncclSuccess) return 0; memcpy(out, &id, sizeof(id)); return 1; }
void* OptComm_CreateRccl(const char* uniqueId, int rank, int world) {
    auto* x = new RcclCtx; x->rank = rank; x->world = world;
    ncclUniqueId id; memcpy(&id, uniqueId, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&x->comm, world, id, rank);
    if (r != ncclSuccess) { fprintf(stderr, "OptComm(rccl) rank %d: ncclCommInitRank failed: %s\n", rank, ncclGetErrorString(r)); delete x; return nullptr; }
    x->api = OptAmd_SlabComm{x, rank, world, rcclHalo, rcclAllReduce};
    return x;
}
const OptAmd_SlabComm* OptComm_RcclSlabComm(void* c) { return &((RcclCtx*)c)->api; }
int OptComm_RcclCount(void* c) { int n = 0; if (ncclCommCount(((RcclCtx*)c)->comm, &n) != ncclSuccess) return -1; return n; }   // ranks RCCL itself sees
int OptComm_RcclError(void* c) { return ((RcclCtx*)c)->err; }
void OptComm_DestroyRccl(void* c) { auto* x = (RcclCtx*)c; ncclCommDestroy(x->comm); delete x; }

void* OptComm_CreateThreadWorld(int world) { return new ThreadWorld(world); }
void OptComm_DestroyThreadWorld(void* w) { delete (ThreadWorld*)w; }
void* OptComm_CreateThreadRank(void* world, int rank) {
    auto* x = new ThreadCtx; x->W = (ThreadWorld*)world; x->rank = rank;
    x->api = OptAmd_SlabComm{x, rank, x->W->world, thrHalo, thrAllReduce};
    return x;
}
const OptAmd_SlabComm* OptComm_ThreadSlabComm(void* c) { return &((ThreadCtx*)c)->api; }
void OptComm_DestroyThreadRank(void* c) { delete (ThreadCtx*)c; }

}  // extern "C"
