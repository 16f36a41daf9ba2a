"""TEST transport (moved out of the product package in round 6): several ranks on ONE GPU.  RCCL refuses two ranks on one device, so the
device buffers are staged through host memory and exchanged over a CPU process group (gloo) -- the model's multi-rank code paths (table-wise
shards, exchange, buckets) then run on the real HIP kernels of a single MI355X.  Used by tests/_dist_worker_gpu.py only."""
import ctypes as C  # noqa: F401

import numpy as np  # noqa: F401
import torch
import torch.distributed as dist

from dlrm_flexflow_amd.comm import TorchComm


class HostStagedComm(TorchComm):
    """TEST transport: device buffers are staged through host memory and exchanged over a CPU process group (gloo).
    Slow and fully synchronous, but it lets several ranks share ONE GPU (RCCL refuses that), so the multi-rank path --
    table-wise shards, exchange buffers, global-batch update, gradient all-reduce -- can run on the HIP kernels of a
    single-GPU box.  Not used by bench.py."""

    def __init__(self, group=None):
        super().__init__(on_gpu=True, group=group)

    def _alltoall(self, user, send, send_counts, recv, recv_counts, stream):
        try:
            sc = [int(send_counts[i]) for i in range(self.world)]
            rc = [int(recv_counts[i]) for i in range(self.world)]
            torch.cuda.synchronize()
            src = self._view(send, sum(sc)).cpu() if sum(sc) else torch.empty(0)
            dst = torch.empty(sum(rc), dtype=torch.float32)
            dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
            if sum(rc):
                self._view(recv, sum(rc)).copy_(dst)
            torch.cuda.synchronize()
            self.calls["alltoall"] += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm alltoall (host-staged) failed:", repr(e), flush=True)
            return 1

    def _allreduce(self, user, buf, count, stream):
        try:
            torch.cuda.synchronize()
            t = self._view(buf, int(count))
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            torch.cuda.synchronize()
            self.calls["allreduce"] += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm allreduce (host-staged) failed:", repr(e), flush=True)
            return 1


    def _reduce_scatter(self, user, send, recv, recv_count, stream):
        try:
            n = int(recv_count)
            torch.cuda.synchronize()
            h = self._view(send, n * self.world).cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            self._view(recv, n).copy_(h[self.rank * n:(self.rank + 1) * n])
            torch.cuda.synchronize()
            self.calls["reduce_scatter"] += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm reduce_scatter (host-staged) failed:", repr(e), flush=True)
            return 1

    def _allgather(self, user, send, recv, send_count, stream):
        try:
            n = int(send_count)
            torch.cuda.synchronize()
            dst = torch.empty(n * self.world, dtype=torch.float32)
            dist.all_gather_into_tensor(dst, self._view(send, n).cpu(), group=self.group)
            self._view(recv, n * self.world).copy_(dst)
            torch.cuda.synchronize()
            self.calls["allgather"] += 1
            return 0
        except Exception as e:  # noqa: BLE001
            print("ffcomm allgather (host-staged) failed:", repr(e), flush=True)
            return 1
