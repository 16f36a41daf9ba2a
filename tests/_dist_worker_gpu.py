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
from host_staged_comm import HostStagedComm  # noqa: E402
import dlrm_helpers as H  # noqa: E402


def main():
    outdir = sys.argv[1]
    direct = len(sys.argv) > 2 and sys.argv[2] == "direct"     # RCCL called from the C++ host layer (host/rccl_comm.cc)
    staged = len(sys.argv) > 2 and sys.argv[2] == "staged"     # several ranks on ONE GPU: host-staged test transport over gloo
    torch.cuda.set_device(0 if staged else int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("gloo" if staged else "nccl", init_method=f"tcp://{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}",
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    comm = HostStagedComm() if staged else TorchComm(on_gpu=True)
    if direct:
        comm = RcclComm(comm)
    out = {}
    if len(sys.argv) > 3 and sys.argv[3].startswith("kaggle"):
        # the Criteo-Kaggle shape at 2048 samples per rank through the DLRM application object (driver flags)
        # kaggle-repl[-noov]: tables of <= 8192 rows data-parallel; six steady-state steps on the resident batch (the inputs are
        # never rewritten, so no new-batch event orders the side-stream gather behind the slab optimizer of the step before)
        from dlrm_flexflow_amd import ffmodel
        world = dist.get_world_size()
        extra, steps = list(sys.argv[4:]), 3            # further driver flags (math modes ...)
        if sys.argv[3].startswith("kaggle-repl"):
            extra, steps = extra + ["--replicate-embedding-rows", "8192"], 6
            if sys.argv[3].endswith("noov"):
                extra.append("--no-overlap")
        graph = sys.argv[3] == "kaggle-graph"          # --capture-exchange: the exchange step captured and replayed as a hipGraph (RcclComm only)
        app = ffmodel.DLRM(H.KAGGLE_ARGS(2048 * world) + ["--device", "0", "--force-exchange"] + extra, comm=comm.struct)
        app.warmup()
        app.train_steps(steps, trace=graph)
        out["uses_graph"] = np.array(int(bool(app.model.uses_graph)))
        app.model.sync()
        m = app.model
        out["pred"] = m.layer_output(m.num_layers - 1).get()
        for l in range(m.num_layers):
            if m.layer_num_weights(l) and m.parameter(l, 0).is_local:
                w = m.parameter(l, 0).get_weights()
                out[f"p{l}"] = w if w.size <= 1 << 16 else np.array([w.astype(np.float64).sum(), np.abs(w).astype(np.float64).sum(),
                                                                     float(w[:64].astype(np.float64).sum())])
        out["alltoall_calls"] = np.array(comm.calls["alltoall"])
        out["allreduce_calls"] = np.array(comm.calls["allreduce"])
        out["allreduce_bucket_calls"] = np.array(comm.calls.get("allreduce_buckets", 0))
        np.savez(os.path.join(outdir, f"rank{dist.get_rank()}.npz"), **out)
        app.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if len(sys.argv) > 3 and sys.argv[3].startswith("opt:"):
        # any optimizer x any placement on the HIP kernels: "opt:<adam|mom>:<dense|sparse>:<table|column|row|mixed>" (tests/test_gpu_round4.py)
        _, okind, path, place = sys.argv[3].split(":")
        kw = dict(adam=H.ADAM_HP) if okind == "adam" else dict(sgd=H.MOM_HP)
        m, h = H.build_golden_dlrm(capi.HIP_LIB_PATH, comm=comm.struct, overlap=True, force_exchange=True,
                                   column_shard_rows=40 if place == "column" else 0, row_shard_rows=40 if place == "row" else 0,
                                   replicate_rows=39 if place == "mixed" else 0,
                                   extra_argv=["--device", "0"] + (["--sparse-embedding-optimizer"] if path == "sparse" else []), **kw)
        recs = H.run_steps(m, h, 3)
        for step, rec in enumerate(recs):
            for k, v in rec.items():
                out[f"s{step}/{k}"] = v
        out["allreduce_calls"] = np.array(comm.calls["allreduce"])
        out["allreduce_bucket_calls"] = np.array(comm.calls.get("allreduce_buckets", 0))
        np.savez(os.path.join(outdir, f"rank{dist.get_rank()}.npz"), **out)
        m.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    column = len(sys.argv) > 3 and sys.argv[3] == "column"     # the 50-row table of the golden model split column-wise over the ranks
    row = len(sys.argv) > 3 and sys.argv[3] == "row"           # ... row-wise: partial sums + reduce-scatter, all-gather backward
    strategy = len(sys.argv) > 3 and sys.argv[3] == "strategy" # table owners from <outdir>/strategy.txt (reference text format)
    replicated = len(sys.argv) > 3 and sys.argv[3] == "replicated"   # tables of <= 39 rows data-parallel (every rank holds a copy)
    sx = ["--import", os.path.join(outdir, "strategy.txt"), "--export", os.path.join(outdir, "export.txt")] if strategy else []
    m, h = H.build_golden_dlrm(capi.HIP_LIB_PATH, comm=comm.struct, overlap=True, force_exchange=True, extra_argv=["--device", "0"] + sx,
                               column_shard_rows=40 if column else 0, row_shard_rows=40 if row else 0, replicate_rows=39 if replicated else 0)
    recs = H.run_steps(m, h, 2)
    for step, rec in enumerate(recs):
        for k, v in rec.items():
            out[f"s{step}/{k}"] = v
    out["alltoall_calls"] = np.array(comm.calls["alltoall"])
    out["allreduce_calls"] = np.array(comm.calls["allreduce"])
    out["allreduce_bucket_calls"] = np.array(comm.calls.get("allreduce_buckets", 0))
    out["reduce_scatter_calls"] = np.array(comm.calls["reduce_scatter"])
    out["allgather_calls"] = np.array(comm.calls["allgather"])
    np.savez(os.path.join(outdir, f"rank{dist.get_rank()}.npz"), **out)
    m.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
