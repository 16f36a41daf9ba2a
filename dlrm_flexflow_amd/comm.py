"""torch.distributed-backed implementation of the `ffcomm` callbacks (host/ffcomm.h).

One process per GPU; the process group is created by the launcher (bench.py / run_dlrm.py):
backend "nccl" (= RCCL over xGMI on ROCm) for device buffers, "gloo" for the CPU tests.
The C++ model hands raw device (or host) pointers and the HIP stream its work is ordered on;
the callback wraps them as tensors without copying and issues the collective on that stream.
PyTorch is plumbing here: it owns no data and runs no arithmetic of the model.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from .ffmodel import ALLREDUCE_FN, ALLTOALL_FN, BARRIER_FN, FFComm


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
        self.calls = {"alltoall": 0, "allreduce": 0}
        self._tensors = {}     # (ptr, count) -> zero-copy tensor view; the model's buffers are fixed after compile()
        self._streams = {}     # hipStream_t -> torch.cuda.ExternalStream
        self._splits = {}
        self._a2a = ALLTOALL_FN(self._alltoall)
        self._ar = ALLREDUCE_FN(self._allreduce)
        self._bar = BARRIER_FN(self._barrier)
        self.struct = FFComm(self.rank, self.world, None, self._a2a, self._ar, self._bar)

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

    def _barrier(self, user):
        try:
            if self.on_gpu:
                torch.cuda.synchronize()
            dist.barrier(group=self.group)
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm barrier failed:", repr(e), flush=True)
            return 1
