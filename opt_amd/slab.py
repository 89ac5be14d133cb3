"""Row-slab tiling of an image problem over the GPUs of one node (one rank per GPU).

The reference is single-GPU; this is the MI355X-native extension the north star asks for:
rank g owns rows [row0, row0+rows) of the image and passes libOpt.so arrays that hold its slab plus `ghost` ghost
rows above and below (OptAmd_PlanSetSlab, include/OptAmd.h; at least 2 lets image_warping run its PCG iteration without
the A*p vector in memory).  Per PCG iteration the ranks all-reduce four scalars; every `ghost` - 1 iterations they exchange
`ghost` image rows of r and p per neighbour (in between, the iteration kernel keeps the ghost rows it still needs current by
itself: SlabJob uses 8 ghost rows, one exchange per 7 iterations).  Nothing else crosses GPUs.

Host-side pieces (numpy only, unit-tested on CPU with gloo): SlabLayout, split_problem, merge_unknowns.
Device-side drivers: SlabJob (RCCL, one process per GPU, used by bench.py) and run_threads (all ranks as
threads of one process on one GPU -- the single-GPU test harness).
"""
import ctypes
import os
import threading

import numpy as np

from . import api, workloads as wl

_HERE = os.path.dirname(os.path.abspath(__file__))
COMM_LIB_PATH = os.path.join(_HERE, "lib", "libOptComm.so")
_comm = None


class SlabLayout:
    """Even split of H rows over `world` ranks; earlier ranks take the remainder."""

    def __init__(self, W, H, rank, world, ghost=2):
        if world * ghost > H:
            raise ValueError("more ranks than image rows allow (every slab must own at least `ghost` rows)")
        self.W, self.H, self.rank, self.world, self.ghost = W, H, rank, world, ghost
        base, rem = divmod(H, world)
        self.rows = base + (1 if rank < rem else 0)
        self.row0 = rank * base + min(rank, rem)
        self.local_H = self.rows + 2 * ghost              # + ghost rows above and below

    @property
    def owned(self):
        return slice(self.row0, self.row0 + self.rows)

    def has_up(self):
        return self.rank > 0

    def has_down(self):
        return self.rank < self.world - 1


def _is_image(a, H, W):
    return isinstance(a, np.ndarray) and a.ndim >= 2 and a.shape[0] == H and a.shape[1] == W


def split_problem(problem, layout):
    """Local copy of a global (host) workloads.Problem for one rank: every image keeps rows row0-ghost .. row0+rows+ghost-1
    (ghost rows outside the global image are zero-filled; for image_warping their Mask is set non-zero so they
    are also excluded by the energy itself).  Scalars are shared."""
    W, H = layout.W, layout.H
    out = []
    for idx, p in enumerate(problem.params):
        a = np.asarray(p)
        if not _is_image(a, H, W):
            out.append(np.array(a, copy=True))
            continue
        loc = np.zeros((layout.local_H,) + a.shape[1:], dtype=a.dtype)
        g = layout.ghost
        lo, hi = layout.row0 - g, layout.row0 + layout.rows + g
        glo, ghi = max(lo, 0), min(hi, H)
        loc[glo - lo: glo - lo + (ghi - glo)] = a[glo:ghi]
        if problem.energy == "image_warping" and idx == 4:          # Mask
            if lo < 0:
                loc[:-lo] = 255
            if hi > H:
                loc[layout.local_H - (hi - H):] = 255
        out.append(loc)
    return wl.Problem(problem.energy, (W, layout.local_H), out, problem.unknown_slots, problem.double, dict(problem.meta))


def merge_unknowns(problem, layouts, local_unknowns):
    """Write every rank's owned rows of the unknown images back into the global problem (in place)."""
    for lay, unk in zip(layouts, local_unknowns):
        for slot, arr in zip(problem.unknown_slots, unk):
            problem.params[slot][lay.owned] = np.asarray(arr)[lay.ghost:lay.ghost + lay.rows]


def comm_lib():
    global _comm
    if _comm is None:
        if not os.path.exists(COMM_LIB_PATH):
            raise RuntimeError(f"{COMM_LIB_PATH} not found: build it with `python -m opt_amd.build`")
        L = ctypes.CDLL(COMM_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.OptComm_UniqueIdBytes.restype = ci
        L.OptComm_GetUniqueId.restype = ci; L.OptComm_GetUniqueId.argtypes = [ctypes.c_char_p]
        L.OptComm_CreateRccl.restype = vp; L.OptComm_CreateRccl.argtypes = [ctypes.c_char_p, ci, ci]
        L.OptComm_RcclSlabComm.restype = ctypes.POINTER(api.OptAmd_SlabComm); L.OptComm_RcclSlabComm.argtypes = [vp]
        L.OptComm_DestroyRccl.argtypes = [vp]
        L.OptComm_RcclCount.restype = ci; L.OptComm_RcclCount.argtypes = [vp]
        L.OptComm_RcclError.restype = ci; L.OptComm_RcclError.argtypes = [vp]
        L.OptComm_PeerSharesDevice.restype = ci; L.OptComm_PeerSharesDevice.argtypes = [vp]
        L.OptComm_PeerPosts.restype = ci; L.OptComm_PeerPosts.argtypes = [vp]
        L.OptComm_PeerSetTiming.argtypes = [vp, ci]
        L.OptComm_PeerTimings.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.OptComm_PeerHandleBytes.restype = ci
        L.OptComm_PeerMaxWorld.restype = ci
        L.OptComm_PeerCreate.restype = vp; L.OptComm_PeerCreate.argtypes = [ci, ci, ctypes.c_long, ctypes.c_double]
        L.OptComm_PeerHandle.argtypes = [vp, ctypes.c_char_p]
        L.OptComm_PeerMemKind.restype = ci; L.OptComm_PeerMemKind.argtypes = [vp]
        L.OptComm_PeerConnect.restype = ci; L.OptComm_PeerConnect.argtypes = [vp, ctypes.c_char_p]
        L.OptComm_PeerSlabComm.restype = ctypes.POINTER(api.OptAmd_SlabComm); L.OptComm_PeerSlabComm.argtypes = [vp]
        L.OptComm_PeerSlabCommExt.restype = ctypes.POINTER(api.OptAmd_SlabCommExt); L.OptComm_PeerSlabCommExt.argtypes = [vp]
        L.OptComm_PeerError.restype = ci; L.OptComm_PeerError.argtypes = [vp]
        L.OptComm_PeerDestroy.argtypes = [vp]
        L.OptComm_PeerSelfTest.restype = ci; L.OptComm_PeerSelfTest.argtypes = [vp, ctypes.c_double]
        L.OptComm_CreateThreadWorld.restype = vp; L.OptComm_CreateThreadWorld.argtypes = [ci]
        L.OptComm_DestroyThreadWorld.argtypes = [vp]
        L.OptComm_CreateThreadRank.restype = vp; L.OptComm_CreateThreadRank.argtypes = [vp, ci]
        L.OptComm_ThreadSlabComm.restype = ctypes.POINTER(api.OptAmd_SlabComm); L.OptComm_ThreadSlabComm.argtypes = [vp]
        L.OptComm_DestroyThreadRank.argtypes = [vp]
        _comm = L
    return _comm


def attach_slab(solver, layout, slab_comm_ptr, ext_ptr=None):
    ok = api.lib().OptAmd_PlanSetSlab(solver.plan, layout.row0, layout.rows, layout.H, slab_comm_ptr)
    if not ok:
        raise RuntimeError("this energy's kernel set does not support slab tiling")
    if ext_ptr is not None:                      # the communicator's optional fast paths (OptAmd_SlabCommExt)
        api.lib().OptAmd_PlanSetSlabExt(solver.plan, ext_ptr)


class PeerComm:
    """The peer-mailbox communicator (csrc/comm/peer_comm.hip): every rank allocates a window on its GPU, the hipIpc handles are
    all-gathered through torch.distributed (any backend -- the data path never touches it again) and every rank maps every window."""

    def __init__(self, rank, world, stage_bytes, timeout_s=None):
        import torch.distributed as dist
        L = comm_lib()
        if world > L.OptComm_PeerMaxWorld():
            raise ValueError(f"peer communicator supports at most {L.OptComm_PeerMaxWorld()} ranks")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if timeout_s is None:
            timeout_s = float(os.environ.get("OPT_AMD_PEER_TIMEOUT", "60"))    # ranks may reach their first all-reduce seconds apart (page-in, module load)
        self.rank, self.world = rank, world
        self.preflight = {"ipc_window": False, "ipc_open": False, "self_test": False, "self_test_ms": None}      # what happened on THIS rank, for bench.py's pre-flight block

        def agree(local_ok, what):             # failures are made collective: either every rank goes on or every rank raises (no rank left in a gather)
            oks = [None] * world
            if world > 1:
                dist.all_gather_object(oks, bool(local_ok))
            else:
                oks[0] = bool(local_ok)
            if not all(oks):
                if self._ctx:
                    L.OptComm_PeerDestroy(self._ctx)
                    self._ctx = None
                raise RuntimeError(f"{what} failed on rank(s) {[r for r, o in enumerate(oks) if not o]}")

        self._ctx = L.OptComm_PeerCreate(rank, world, int(stage_bytes), float(timeout_s))
        self.preflight["ipc_window"] = bool(self._ctx)
        agree(self._ctx, "OptComm_PeerCreate (IPC-exportable device memory)")
        n = L.OptComm_PeerHandleBytes()
        buf = ctypes.create_string_buffer(n)
        L.OptComm_PeerHandle(self._ctx, buf)
        handles = [None] * world
        if world > 1:
            dist.all_gather_object(handles, bytes(buf.raw))
        else:
            handles[0] = bytes(buf.raw)
        opened = bool(L.OptComm_PeerConnect(self._ctx, b"".join(handles)))
        self.preflight["ipc_open"] = opened
        agree(opened, "OptComm_PeerConnect (hipIpcOpenMemHandle)")
        if world > 1:
            dist.barrier()                     # every window is mapped everywhere before the first peer store
        # one all-reduce and one halo exchange with known answers; every rank must pass (collective verdict), else the caller falls back to RCCL
        import time
        t0 = time.perf_counter()
        ok = bool(L.OptComm_PeerSelfTest(self._ctx, float(os.environ.get("OPT_AMD_PEER_SELFTEST_TIMEOUT", "5"))))
        self.preflight.update(self_test=ok, self_test_ms=1e3 * (time.perf_counter() - t0), shares_device=bool(L.OptComm_PeerSharesDevice(self._ctx)),
                              posted_allreduce=bool(L.OptComm_PeerPosts(self._ctx)))
        if world > 1:
            verdicts = [None] * world
            dist.all_gather_object(verdicts, ok)
            ok = all(verdicts)
        self.self_test_ok = ok
        self.mem_kind = {3: "uncached", 1: "fine-grained", 0: "plain"}.get(L.OptComm_PeerMemKind(self._ctx), "?")

    def slab_comm(self):
        return comm_lib().OptComm_PeerSlabComm(self._ctx)

    def slab_comm_ext(self):
        return comm_lib().OptComm_PeerSlabCommExt(self._ctx)

    def error(self):
        return comm_lib().OptComm_PeerError(self._ctx)

    def set_timing(self, on):
        """hipEvent timing of the communicator's own kernels (all-reduce / post, halo push + pull) from now on."""
        comm_lib().OptComm_PeerSetTiming(self._ctx, 1 if on else 0)

    def timings(self):
        """{all-reduce: (launches, total ms), halo: (exchanges, total ms)} since set_timing(True)."""
        out = (ctypes.c_double * 4)()
        comm_lib().OptComm_PeerTimings(self._ctx, out)
        return {"allreduce": (int(out[0]), out[1]), "halo": (int(out[2]), out[3])}

    def close(self):
        import torch.distributed as dist
        if self._ctx:
            if self.world > 1 and dist.is_initialized():
                dist.barrier()                 # nobody unmaps a window a peer may still be writing to
            comm_lib().OptComm_PeerDestroy(self._ctx)
            self._ctx = None


class SlabJob:
    """One rank of a multi-process solve (bench.py --gpus N).  torch.distributed is already initialised (nccl = RCCL on the GPU
    boxes; gloo works too: it only carries the set-up).  comm = "peer": peer-mapped mailboxes (PeerComm, the default);
    comm = "rccl": ncclAllReduce / grouped send-recv on a communicator created here from a broadcast id."""

    def __init__(self, energy, W, H, rank, world, kind="gaussNewtonGPU", double=False, problem=None, ghost=None, comm="peer"):
        import torch
        import torch.distributed as dist
        if ghost is None:                      # deep ghost zones: 16 extra rows per slab buy 6 of 7 halo exchanges
            ghost = max(2, min(8, H // world))
        self.layout = SlabLayout(W, H, rank, world, ghost)
        if problem is None and energy == "image_warping":      # the benchmark workload: every rank builds only its own rows (8192^2 would be 2 GB per rank otherwise)
            lay = self.layout
            self.local = wl.image_warping_rows(W, H, lay.row0 - ghost, lay.row0 + lay.rows + ghost, double=double)
        else:
            glob = problem if problem is not None else getattr(wl, energy)(W, H, double=double)
            self.local = split_problem(glob, self.layout)
            del glob
        self.params = api.to_device(self.local)
        self.comm_kind, self.world = comm, world
        L = comm_lib()
        self._peer = None
        dev = torch.cuda.current_device()
        self.preflight = {"rank": rank, "device": dev, "requested_comm": comm,
                          "can_access_peer": [bool(torch.cuda.can_device_access_peer(dev, j)) if j != dev else True for j in range(torch.cuda.device_count())]}
        why = "requested"
        if comm == "peer":
            # the largest exchange moves `ghost` rows of two solver vectors (r and p) per side; rows are W * (unknown scalars per pixel) wide
            scalars = sum(int(np.prod(np.asarray(self.local.params[i]).shape[2:])) or 1 for i in self.local.unknown_slots)
            stage = 2 * ghost * W * scalars * (8 if double else 4) + 4096
            try:
                self._peer = PeerComm(rank, world, stage)
                usable = self._peer.self_test_ok
            except RuntimeError as e:                      # no IPC-exportable window / mapping failed: same on every rank (collective calls inside)
                self._peer, usable = None, False
                why = f"peer communicator unavailable: {e}"
                if rank == 0:
                    print(f"opt_amd.slab: peer communicator unavailable ({e}); falling back to RCCL", flush=True)
            if self._peer is not None:
                self.preflight["peer"] = dict(self._peer.preflight, mem_kind=self._peer.mem_kind)
            if not usable:
                if self._peer is not None:
                    why = "peer communicator failed its self-test on some rank"
                    if rank == 0:
                        print("opt_amd.slab: peer communicator failed its self-test; falling back to RCCL", flush=True)
                    self._peer.close()
                    self._peer = None
                comm = self.comm_kind = "rccl"
        slab_ext = None
        if comm == "peer":
            slab_comm, slab_ext = self._peer.slab_comm(), self._peer.slab_comm_ext()
        elif comm == "rccl":
            n = L.OptComm_UniqueIdBytes()
            buf = ctypes.create_string_buffer(n)
            if rank == 0:
                assert L.OptComm_GetUniqueId(buf)
            ids = [bytes(buf.raw)]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            self._id = ids[0]
            self._ctx = L.OptComm_CreateRccl(self._id, rank, world)
            if not self._ctx:
                raise RuntimeError(f"rank {rank}: ncclCommInitRank failed (ranks that share one GPU cannot form an RCCL communicator)")
            slab_comm = L.OptComm_RcclSlabComm(self._ctx)
        else:
            raise ValueError(f"unknown comm {comm!r}")
        self.preflight.update(comm=self.comm_kind, why=why)
        self.solver = api.Solver(api.energy_file(energy), kind, (W, self.layout.local_H), double=double)
        attach_slab(self.solver, self.layout, slab_comm, slab_ext)

    def comm_ranks(self):
        """How many ranks the communicator itself sees (ncclCommCount for RCCL; the mapped windows for the peer communicator)."""
        if self.comm_kind == "rccl":
            return comm_lib().OptComm_RcclCount(self._ctx)
        return self._peer.world

    def comm_error(self):
        """Non-zero once the communicator is in its error state (peer: a wait timed out, an exchange was oversize, a HIP error -- codes in
        csrc/comm/peer_comm.hip; RCCL: the ncclResult_t of the failed call).  The collectives after that point were skipped: the sums are garbage."""
        return self._peer.error() if self.comm_kind == "peer" else comm_lib().OptComm_RcclError(self._ctx)

    def set_comm_timing(self, on):
        if self.comm_kind == "peer":
            self._peer.set_timing(on)

    def comm_timings(self):
        return self._peer.timings() if self.comm_kind == "peer" else None

    def owned_unknowns(self):
        g = self.layout.ghost
        return [self.params[i][g:g + self.layout.rows].cpu().numpy() for i in self.local.unknown_slots]

    def close(self):
        import torch
        torch.cuda.synchronize()
        err = self.comm_error()                # a time-out in the LAST collective of a run is seen by no later call: look once more
        self.solver.close()
        if self.comm_kind == "rccl":
            comm_lib().OptComm_DestroyRccl(self._ctx)
        else:
            self._peer.close()
        if err:
            raise RuntimeError(f"peer communicator of rank {self.layout.rank} timed out waiting for a peer (code {err}); results of this job are invalid")


def run_threads(problem, world, kind="gaussNewtonGPU", solver_params=None, steps=None, ghost=2):
    """Solve `problem` (host arrays, modified in place) with `world` slabs as threads of this process on the
    current GPU.  Returns the per-step cost list of rank 0.  Test harness for single-GPU boxes."""
    import torch
    L = comm_lib()
    W, H = problem.dims
    tw = L.OptComm_CreateThreadWorld(world)
    layouts = [SlabLayout(W, H, r, world, ghost) for r in range(world)]
    costs = [[] for _ in range(world)]
    results = [None] * world
    errors = []
    dev_index = torch.cuda.current_device()

    def work(r):
        try:
            torch.cuda.set_device(dev_index)
            lay = layouts[r]
            loc = split_problem(problem, lay)
            dev = api.to_device(loc)
            s = api.Solver(api.energy_file(problem.energy), kind, (W, lay.local_H), double=problem.double)
            for k, v in (solver_params or {}).items():
                s.set_parameter(k, v)
            ctx = L.OptComm_CreateThreadRank(tw, r)
            attach_slab(s, lay, L.OptComm_ThreadSlabComm(ctx))
            s.init(dev)
            costs[r].append(s.cost())
            n = 0
            while s.step(dev):
                costs[r].append(s.cost())
                n += 1
                if steps is not None and n >= steps:
                    break
            torch.cuda.synchronize()
            results[r] = [dev[i].cpu().numpy() for i in problem.unknown_slots]
            s.close()
            L.OptComm_DestroyThreadRank(ctx)
        except Exception as e:  # noqa
            errors.append((r, e))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    L.OptComm_DestroyThreadWorld(tw)
    if errors:
        raise errors[0][1]
    merge_unknowns(problem, layouts, results)
    return costs[0]
