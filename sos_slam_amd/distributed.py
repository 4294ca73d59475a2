"""Multi-GPU glue (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI).

The path has exactly one exchange step per Gauss-Newton iteration: the packed fp32
accumulator blocks [top_A | top_L | accD | accE | accEB | Hcc | bc | counts] (0.55 MB at 12 KF, 1.26 MB at
16 KF) are summed over ranks with ONE all-reduce; every rank then runs the identical fp64 stitch + solve.
The only other cross-rank value is the order statistic behind frameEnergyTH
(FS/FullSystemOptimize.cpp:101-116), obtained with an all-gather of the <= P energies of the newest frame.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

AR_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)
NTH_FN = C.CFUNCTYPE(C.c_float, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_float)
AR64_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)


class _DevView:
    """Zero-copy view of a raw device pointer for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def global_nth(dist, local: np.ndarray, frac: float) -> float:
    """Element at index int(frac * N) of the globally sorted concatenation of all ranks' `local` arrays
    (what std::nth_element + [nthIdx] yields in setNewFrameEnergyTH); -1 when the global list is empty."""
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, np.ascontiguousarray(local, dtype=np.float32))
    allv = np.concatenate(parts) if parts else np.zeros(0, np.float32)
    if allv.size == 0:
        return -1.0
    # the index as the C code forms it (FS/FullSystemOptimize.cpp:104: a FLOAT setting times the count, truncated): in double
    # 0.7f * 170 = 118.999998 -> 118, in float it rounds to 119.0 -> 119; one rank and N ranks must pick the same element
    k = int(np.float32(frac) * np.float32(allv.size))
    k = min(k, allv.size - 1)
    return float(np.partition(allv, k)[k])


def make_hooks(dist, torch, device_pointers=None):
    """The three exchange callbacks of a multi-rank run as ctypes function objects (AR_FN, NTH_FN, AR64_FN):
    all-reduce of the packed fp32 accumulator (a DEVICE pointer with the nccl backend; with gloo -- CPU tests -- the pointer is
    host memory), the global order statistic of setNewFrameEnergyTH, and the all-reduce of the keyframe-rate fp64 sums."""
    on_gpu = dist.get_backend() == "nccl"
    # gloo with the GPU facade (several ranks sharing ONE GPU, bench.py's SOS_BENCH_SINGLE_GPU=1 rehearsal of the N > 1 flow): the
    # accumulator pointer is device memory, the collective runs on a host copy
    staged = (not on_gpu) and bool(device_pointers)

    def _ar(user, ptr, n):
        if staged:
            t = torch.as_tensor(_DevView(ptr, n), device="cuda")
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
            torch.cuda.synchronize()
        elif on_gpu:
            t = torch.as_tensor(_DevView(ptr, n), device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()
        else:
            buf = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
            t = torch.from_numpy(buf)  # shares the memory: summed in place
            dist.all_reduce(t)

    def _nth(user, ptr, count, frac):
        local = np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count > 0 else np.zeros(0, np.float32)
        return global_nth(dist, local, frac)

    def _ar64(user, ptr, n):
        # keyframe-rate fp64 sums (marginalisation prior update, mean |idepth| of the termination test): host buffer
        buf = np.ctypeslib.as_array(ptr, shape=(n,))
        t = torch.from_numpy(buf.copy())
        if on_gpu:
            t = t.cuda()
        dist.all_reduce(t)
        buf[:] = t.cpu().numpy()

    return AR_FN(_ar), NTH_FN(_nth), AR64_FN(_ar64)


def attach(sysm, dist, torch, device_pointers=None):
    """Install the all-reduce / order-statistic hooks of a host.System for a multi-rank run."""
    from . import host

    sysm._ar_cb, sysm._nth_cb, sysm._ar64_cb = make_hooks(dist, torch, device_pointers)  # the system keeps the callbacks alive
    L = host.load()
    L.sosf_set_hooks.argtypes = [C.c_void_p, AR_FN, NTH_FN, C.c_void_p]
    rc = L.sosf_set_hooks(sysm.h_, sysm._ar_cb, sysm._nth_cb, None)
    if rc != 0:
        raise RuntimeError(f"sosf_set_hooks failed: {rc}")
    L.sosf_set_allreduce_f64_hook.argtypes = [C.c_void_p, AR64_FN]
    rc = L.sosf_set_allreduce_f64_hook(sysm.h_, sysm._ar64_cb)
    if rc != 0:
        raise RuntimeError(f"sosf_set_allreduce_f64_hook failed: {rc}")


class NativeComm:
    """RCCL communicator owned by the library (sos_comm): the per-iteration all-reduce / all-gather run on the
    library's own stream inside the fused calls, with no Python or host round trip (include/sos_slam.h, "multi-GPU
    exchange").  torch.distributed is only used once, to hand rank 0's 128-byte RCCL id to the other ranks."""

    def __init__(self, dist, torch, device: int):
        import os
        from . import lib as _lib
        self.L = _lib.load()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")  # the copy torch already loaded
        rc = self.L.sos_rccl_load(path.encode() if os.path.exists(path) else None)
        rank, world = dist.get_rank(), dist.get_world_size()
        idbuf = (C.c_ubyte * 128)()
        if rank == 0 and rc == 0:
            rc = self.L.sos_rccl_unique_id(idbuf)
        # the ranks must leave this constructor the same way: agree on the local status before anybody raises
        st = torch.tensor([rc != 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        if int(st.item()) != 0:
            raise RuntimeError(f"RCCL not usable by the library on some rank (local status {rc})")
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        idbuf = (C.c_ubyte * 128).from_buffer_copy(raw)
        self.h = C.c_void_p()
        rc = self.L.sos_comm_create(idbuf, world, rank, device, C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"sos_comm_create failed: {rc}")
        self._systems = []

    def attach(self, sysm):
        from . import host
        rc = host.load().sosf_set_comm(sysm.h_, self.h)
        if rc != 0:
            raise RuntimeError(f"sosf_set_comm failed: {rc}")
        self._systems.append(sysm)

    def close(self):
        from . import host
        for s in self._systems:
            if getattr(s, "h_", None):
                host.load().sosf_set_comm(s.h_, None)
        self._systems = []
        if self.h:
            self.L.sos_comm_destroy(self.h)
            self.h = C.c_void_p()
