"""torch.distributed-backed implementation of the `ffcomm` callbacks (host/ffcomm.h).

One process per GPU; the process group is created by the launcher (bench.py / run_dlrm.py):
backend "nccl" (= RCCL over xGMI on ROCm) for device buffers, "gloo" for the CPU tests.
The C++ model hands raw device (or host) pointers and the HIP stream its work is ordered on;
the callback wraps them as tensors without copying and issues the collective on that stream.
PyTorch is plumbing here: it owns no data and runs no arithmetic of the model.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from .ffmodel import ALLGATHER_FN, ALLREDUCE_FN, ALLTOALL_FN, BARRIER_FN, REDUCE_SCATTER_FN, FFComm


class _CudaView:
    """Zero-copy view of device memory through __cuda_array_interface__."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}


def _as_tensor(ptr: int, count: int, on_gpu: bool) -> torch.Tensor:
    if count == 0:
        return torch.empty(0, dtype=torch.float32, device="cuda" if on_gpu else "cpu")
    if on_gpu:
        return torch.as_tensor(_CudaView(ptr, count), device="cuda")
    buf = (C.c_float * count).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.float32))


class TorchComm:
    """Builds the FFComm struct; keep the object alive as long as the model uses it."""

    def __init__(self, on_gpu: bool, group=None):
        assert dist.is_initialized()
        self.on_gpu = on_gpu
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.calls = {"alltoall": 0, "allreduce": 0, "reduce_scatter": 0, "allgather": 0}
        self._tensors = {}     # (ptr, count) -> zero-copy tensor view; the model's buffers are fixed after compile()
        self._streams = {}     # hipStream_t -> torch.cuda.ExternalStream
        self._splits = {}
        self._a2a = ALLTOALL_FN(self._alltoall)
        self._ar = ALLREDUCE_FN(self._allreduce)
        self._bar = BARRIER_FN(self._barrier)
        self._rs = REDUCE_SCATTER_FN(self._reduce_scatter)
        self._ag = ALLGATHER_FN(self._allgather)
        self.struct = FFComm(self.rank, self.world, None, self._a2a, self._ar, self._bar, 0, self._rs, self._ag)

    def _stream_ctx(self, stream):
        if self.on_gpu and stream:
            ext = self._streams.get(stream)
            if ext is None:
                ext = self._streams[stream] = torch.cuda.ExternalStream(stream)
            return torch.cuda.stream(ext)
        import contextlib
        return contextlib.nullcontext()

    def _view(self, ptr, count):
        key = (ptr, count)
        t = self._tensors.get(key)
        if t is None:
            t = self._tensors[key] = _as_tensor(ptr, count, self.on_gpu)
        return t

    def _alltoall(self, user, send, send_counts, recv, recv_counts, stream):
        try:
            key = (C.addressof(send_counts.contents), C.addressof(recv_counts.contents))
            sp = self._splits.get(key)
            if sp is None:
                sp = self._splits[key] = ([int(send_counts[i]) for i in range(self.world)], [int(recv_counts[i]) for i in range(self.world)])
            sc, rc = sp
            with self._stream_ctx(stream):
                dist.all_to_all_single(self._view(recv, sum(rc)), self._view(send, sum(sc)), output_split_sizes=rc, input_split_sizes=sc,
                                       group=self.group)
            self.calls["alltoall"] += 1
            return 0
        except Exception as e:  # noqa: BLE001  (must not unwind into C++)
            print("ffcomm alltoall failed:", repr(e), flush=True)
            return 1

    def _allreduce(self, user, buf, count, stream):
        try:
            with self._stream_ctx(stream):
                dist.all_reduce(self._view(buf, int(count)), op=dist.ReduceOp.SUM, group=self.group)
            self.calls["allreduce"] += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm allreduce failed:", repr(e), flush=True)
            return 1

    def _reduce_scatter(self, user, send, recv, recv_count, stream):
        """Row-wise sharded table, forward: partial bag sums of the global batch -> this rank's samples, summed."""
        try:
            n = int(recv_count)
            with self._stream_ctx(stream):
                src = self._view(send, n * self.world)
                if self.on_gpu:
                    dist.reduce_scatter_tensor(self._view(recv, n), src, op=dist.ReduceOp.SUM, group=self.group)
                else:            # gloo has no reduce-scatter: all-reduce a copy, keep this rank's block
                    tmp = src.clone()
                    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
                    self._view(recv, n).copy_(tmp[self.rank * n:(self.rank + 1) * n])
            self.calls["reduce_scatter"] += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm reduce_scatter failed:", repr(e), flush=True)
            return 1

    def _allgather(self, user, send, recv, send_count, stream):
        try:
            n = int(send_count)
            with self._stream_ctx(stream):
                dist.all_gather_into_tensor(self._view(recv, n * self.world), self._view(send, n), group=self.group)
            self.calls["allgather"] += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm allgather failed:", repr(e), flush=True)
            return 1

    def _barrier(self, user):
        try:
            if self.on_gpu:
                torch.cuda.synchronize()
            dist.barrier(group=self.group)
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm barrier failed:", repr(e), flush=True)
            return 1


class RcclComm:
    """The same FFComm callbacks served by RCCL directly from the C++ host layer (host/rccl_comm.cc): the collectives
    are enqueued on the model's HIP streams without a round trip through Python (measured with a 1-rank group on the
    Kaggle shape: the Python callbacks cost the step ~50 us, mostly the gaps around each RCCL kernel).
    `boot` is a TorchComm over an initialised NCCL process group: it broadcasts the unique id and keeps the barrier.
    Raises RuntimeError on every rank if any rank cannot set it up (use `boot` then)."""

    def __init__(self, boot: TorchComm, own_bucket_channel: bool = True):
        """own_bucket_channel (default): a second communicator (ncclCommSplit) for the MLP-gradient buckets, so that RCCL does not order them
        against the all-to-alls; taken only when EVERY rank can make it (`--allreduce-shared-channel` of bench.py / run_dlrm.py turns it off)."""
        from . import ffmodel
        assert boot.on_gpu
        self.boot = boot
        self.rank, self.world = boot.rank, boot.world
        L = ffmodel.lib()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = path.encode() if os.path.exists(path) else None
        flag = torch.tensor([1 if L.flexflow_rccl_available(path) == 0 else 0], dtype=torch.int32, device="cuda")
        idb = (C.c_ubyte * 128)()
        if self.rank == 0 and int(flag.item()) and L.flexflow_rccl_get_unique_id(idb, path) != 0:
            flag.zero_()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=boot.group)          # everyone or no one enters ncclCommInitRank
        if int(flag.item()) == 0:
            raise RuntimeError("direct RCCL unavailable: " + (L.flexflow_rccl_last_error() or b"").decode())
        t = torch.tensor(list(idb), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, src=dist.get_global_rank(boot.group, 0) if boot.group is not None else 0, group=boot.group)
        idb = (C.c_ubyte * 128)(*t.cpu().tolist())
        self.struct = ffmodel.FFComm()
        ok = L.flexflow_rccl_comm_create(idb, self.rank, self.world, path, C.byref(self.struct)) == 0
        flag.fill_(1 if ok else 0)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=boot.group)
        if int(flag.item()) == 0:
            if ok:
                L.flexflow_rccl_comm_destroy(C.byref(self.struct))
            raise RuntimeError("ncclCommInitRank failed on some rank: " + (L.flexflow_rccl_last_error() or b"").decode())
        self.struct.barrier = boot._bar
        self._L = L
        # known-answer check of the collectives on the new communicator (uneven all-to-all with a zero-length block,
        # all-reduce, reduce-scatter, all-gather); every rank must pass or all fall back to the torch callbacks together
        self._base = {"alltoall": 0, "allreduce": 0, "reduce_scatter": 0, "allgather": 0}
        ok = self._self_test()
        flag.fill_(1 if ok else 0)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=boot.group)
        if int(flag.item()) == 0:
            self.close()
            raise RuntimeError("direct RCCL communicator failed its self-test on some rank")
        # The buckets' own channel.  ncclCommSplit is COLLECTIVE: the ranks agree (MIN) that every one of them has the symbol BEFORE any of them
        # makes the call, and agree again on its outcome -- a rank that failed alone would leave its peers inside the call, or send its buckets on
        # a communicator they do not use (round-5 advisor).  Only after the self-test's own agreement above: every rank is still here.
        self.own_bucket_channel = False
        if own_bucket_channel:
            L.flexflow_rccl_comm_enable_bucket_channel.restype = C.c_int
            L.flexflow_rccl_has_comm_split.restype = C.c_int
            L.flexflow_rccl_has_comm_split.argtypes = [C.c_char_p]
            flag.fill_(1 if L.flexflow_rccl_has_comm_split(path) == 0 else 0)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=boot.group)
            if int(flag.item()) == 1:
                mine = L.flexflow_rccl_comm_enable_bucket_channel(C.byref(self.struct)) == 0
                flag.fill_(1 if mine else 0)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=boot.group)
                if int(flag.item()) == 1:
                    self.own_bucket_channel = True
                else:
                    L.flexflow_rccl_comm_disable_bucket_channel(C.byref(self.struct))        # every rank: back to the shared channel together
        self._base = self.calls                    # `calls` counts the model's collectives only

    def _self_test(self) -> bool:
        W, r = self.world, self.rank
        # rank r sends (p + r) % 3 floats to peer p; element value = 1000 r + 10 p + j
        sc = [(p + r) % 3 for p in range(W)]
        rc = [(r + p) % 3 for p in range(W)]
        send = torch.tensor([1000.0 * r + 10.0 * p + j for p in range(W) for j in range(sc[p])] or [0.0], dtype=torch.float32, device="cuda")
        recv = torch.full((max(sum(rc), 1),), -1.0, dtype=torch.float32, device="cuda")
        ar = torch.full((5,), float(r + 1), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        scA, rcA = (C.c_int64 * W)(*sc), (C.c_int64 * W)(*rc)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.struct.alltoall_f32(self.struct.user, send.data_ptr(), scA, recv.data_ptr(), rcA, stream) != 0:
            return False
        if self.struct.allreduce_sum_f32(self.struct.user, ar.data_ptr(), 5, stream) != 0:
            return False
        # reduce-scatter: rank r contributes (r + 1) * (block index + 1) in every element; all-gather: 3 floats of 100 r + j
        rs_in = torch.tensor([(r + 1.0) * (p + 1.0) for p in range(W) for _ in range(3)], dtype=torch.float32, device="cuda")
        rs_out = torch.full((3,), -1.0, dtype=torch.float32, device="cuda")
        ag_in = torch.tensor([100.0 * r + j for j in range(3)], dtype=torch.float32, device="cuda")
        ag_out = torch.full((3 * W,), -1.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if self.struct.reduce_scatter_sum_f32(self.struct.user, rs_in.data_ptr(), rs_out.data_ptr(), 3, stream) != 0:
            return False
        if self.struct.allgather_f32(self.struct.user, ag_in.data_ptr(), ag_out.data_ptr(), 3, stream) != 0:
            return False
        torch.cuda.synchronize()
        exp = [1000.0 * p + 10.0 * r + j for p in range(W) for j in range(rc[p])]
        got = recv[:len(exp)].cpu().tolist()
        return (got == exp and ar.cpu().tolist() == [W * (W + 1) / 2.0] * 5
                and rs_out.cpu().tolist() == [(r + 1.0) * W * (W + 1) / 2.0] * 3
                and ag_out.cpu().tolist() == [100.0 * p + j for p in range(W) for j in range(3)])

    @property
    def calls(self):
        a, r, s, g = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._L.flexflow_rccl_comm_calls(C.byref(self.struct), C.byref(a), C.byref(r))
        self._L.flexflow_rccl_comm_calls2(C.byref(self.struct), C.byref(s), C.byref(g))
        self._L.flexflow_rccl_comm_bucket_calls.restype = C.c_int64
        nb = int(self._L.flexflow_rccl_comm_bucket_calls(C.byref(self.struct), None))       # the MLP-gradient buckets (issued from inside backward())
        return {"alltoall": a.value - self._base["alltoall"], "allreduce": r.value - self._base["allreduce"], "allreduce_buckets": nb,
                "reduce_scatter": s.value - self._base["reduce_scatter"], "allgather": g.value - self._base["allgather"]}

    def close(self):
        if self._L is not None:
            self._L.flexflow_rccl_comm_destroy(C.byref(self.struct))
            self._L = None
