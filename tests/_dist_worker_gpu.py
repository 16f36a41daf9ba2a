"""One RCCL rank on the GPU box (launched by tests/test_gpu_model.py): product kernels + product
TorchComm over the nccl backend, with --force-exchange so a single rank walks the whole
all-to-all / all-reduce path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dlrm_flexflow_amd import capi  # noqa: E402
from dlrm_flexflow_amd.comm import RcclComm, TorchComm  # noqa: E402
import dlrm_helpers as H  # noqa: E402


def main():
    outdir = sys.argv[1]
    direct = len(sys.argv) > 2 and sys.argv[2] == "direct"     # RCCL called from the C++ host layer (host/rccl_comm.cc)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", init_method=f"tcp://{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}",
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    comm = TorchComm(on_gpu=True)
    if direct:
        comm = RcclComm(comm)
    m, h = H.build_golden_dlrm(capi.HIP_LIB_PATH, comm=comm.struct, overlap=True, force_exchange=True)
    recs = H.run_steps(m, h, 2)
    out = {}
    for step, rec in enumerate(recs):
        for k, v in rec.items():
            out[f"s{step}/{k}"] = v
    out["alltoall_calls"] = np.array(comm.calls["alltoall"])
    out["allreduce_calls"] = np.array(comm.calls["allreduce"])
    np.savez(os.path.join(outdir, f"rank{dist.get_rank()}.npz"), **out)
    m.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
