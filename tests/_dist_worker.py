"""One rank of the world_size-2 gloo tests (launched by tests/test_ffmodel_host.py).
Kernels come from the CPU oracle (test infrastructure); the collectives are the product's
TorchComm callbacks over the gloo backend."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dlrm_flexflow_amd import ffmodel  # noqa: E402
from dlrm_flexflow_amd.comm import TorchComm  # noqa: E402
import dlrm_helpers as H  # noqa: E402


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo", init_method=f"tcp://{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}",
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank = dist.get_rank()
    comm = TorchComm(on_gpu=False)
    out = {}
    if mode.startswith("opt:"):
        # any optimizer x any placement (round 4): "opt:<adam|mom>:<dense|sparse>:<table|column|row|mixed>"
        _, okind, path, place = mode.split(":")
        kw = dict(adam=H.ADAM_HP) if okind == "adam" else dict(sgd=H.MOM_HP)
        m, h = H.build_golden_dlrm(H.oracle_backend(), comm=comm.struct, overlap=True, force_exchange=True,
                                   column_shard_rows=40 if place == "column" else 0, row_shard_rows=40 if place == "row" else 0,
                                   replicate_rows=39 if place == "mixed" else 0,
                                   extra_argv=["--sparse-embedding-optimizer"] if path == "sparse" else [], **kw)
        recs = H.run_steps(m, h, 3)
        for step, rec in enumerate(recs):
            for k, v in rec.items():
                out[f"s{step}/{k}"] = v
        m.close()
    elif mode.startswith("buckets"):
        # round 5: the MLP gradients' all-reduce in buckets issued from inside backward() (forced: TorchComm's calls block the host, so
        # the default keeps one bucket), tiny buckets, the biggest layer's weight gradient in two row blocks; "buckets-mixed": with
        # data-parallel tables in the dense slab (the part no bucket covers)
        extra = ["--bucket-allreduce", "--allreduce-bucket-floats", "64", "--big-dw-chunks", "2", "--big-dw-min-weights", "1"]
        if mode.endswith("-direct"):        # round 6: every bucket (and what no bucket covers) through the direct all-reduce
            extra.append("--direct-allreduce")
        m, h = H.build_golden_dlrm(H.oracle_backend(), comm=comm.struct, overlap=True, force_exchange=True, extra_argv=extra,
                                   replicate_rows=39 if mode.startswith("buckets-mixed") else 0)
        recs = H.run_steps(m, h, 2)
        for step, rec in enumerate(recs):
            for k, v in rec.items():
                out[f"s{step}/{k}"] = v
        out["bucket_calls"] = np.array(m.counter("allreduce_bucket_calls"))
        out["direct_allreduces"] = np.array(m.counter("direct_allreduces"))
        m.close()
    elif mode in ("golden", "column", "strategy", "row", "replicated", "replicated_all", "replicated_adam"):
        extra = ["--import", os.path.join(outdir, "strategy.txt"), "--export", os.path.join(outdir, "export.txt")] if mode == "strategy" else []
        m, h = H.build_golden_dlrm(H.oracle_backend(), comm=comm.struct, overlap=True, force_exchange=True,
                                   column_shard_rows=40 if mode == "column" else 0, row_shard_rows=40 if mode == "row" else 0, extra_argv=extra,
                                   replicate_rows={"replicated": 39, "replicated_all": 1000, "replicated_adam": 1000}.get(mode, 0),
                                   adam=dict(alpha=0.001) if mode == "replicated_adam" else None)
        recs = H.run_steps(m, h, 2)
        for step, rec in enumerate(recs):
            for k, v in rec.items():
                out[f"s{step}/{k}"] = v
        m.close()
    elif mode in ("hdf5", "hdf5_replicated"):
        rep = ["--replicate-embedding-rows", "60"] if mode == "hdf5_replicated" else []      # tables of 50 and 7 rows data-parallel, the 300-row one owned
        app = ffmodel.DLRM(["--backend", H.oracle_backend()] + H.HDF5_ARGS + rep + ["--dataset", os.path.join(outdir, "day.h5")], comm=comm.struct)
        app.warmup()
        app.train_steps(7, trace=False)          # 1 + 7 steps over 6 batches: wraps around once
        app.model.sync()
        out["dense"] = app.dense_input().get()
        out["label"] = app.model.label_tensor.get()
        out["num_samples"] = np.array(app.num_samples)
        for t in range(3):
            if app.sparse_input(t).is_local:
                out[f"sparse{t}"] = app.sparse_input(t).get(np.int64)
                out[f"emb{t}"] = app.model.parameter(2 + t, 0).get_weights()
        out["w_bot"] = app.model.parameter(0, 0).get_weights()
        out["w_top"] = app.model.parameter(app.model.num_layers - 1, 0).get_weights()
        app.close()
    elif mode == "dot":
        # the fused pairwise-dot interaction behind the exchange: embedding outputs arrive by all-to-all, the Concat in
        # front of the interaction unpacks them, its backward packs the gradients for the way back
        app = ffmodel.DLRM(["--backend", H.oracle_backend()] + H.DOT_ARGS + ["--arch-interaction-op", "dot-tril"], comm=comm.struct)
        app.warmup()
        app.train_steps(3, trace=False)
        app.model.sync()
        m = app.model
        out["pred"] = m.layer_output(m.num_layers - 1).get()
        for l in range(m.num_layers):
            if m.layer_num_weights(l) and m.parameter(l, 0).is_local:
                out[f"p{l}"] = m.parameter(l, 0).get_weights()
        app.close()
    else:
        args = ["--backend", H.oracle_backend(), "-b", "64", "--arch-sparse-feature-size", "8", "--arch-embedding-size",
                "50-7-300-3-1000-20-11", "--arch-mlp-bot", "13-32-8", "--arch-mlp-top", "64-32-1", "--data-size", "128", "--epochs", "6"]
        app = ffmodel.DLRM(args, comm=comm.struct)
        app.warmup()
        pm0 = app.model.perf_metrics()
        out["mse_first"] = np.array(pm0.mse_loss / max(pm0.train_all, 1))
        app.run_epochs()
        pm1 = app.model.perf_metrics()
        out["mse_last"] = np.array(pm1.mse_loss / max(pm1.train_all, 1))
        out["w_bot"] = app.model.parameter(0, 0).get_weights()
        out["w_top"] = app.model.parameter(app.model.num_layers - 1, 0).get_weights()
        app.close()
    out["alltoall_calls"] = np.array(comm.calls["alltoall"])
    out["allreduce_calls"] = np.array(comm.calls["allreduce"])
    out["reduce_scatter_calls"] = np.array(comm.calls["reduce_scatter"])
    out["allgather_calls"] = np.array(comm.calls["allgather"])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
