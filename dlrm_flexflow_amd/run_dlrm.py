#!/usr/bin/env python3
"""run_dlrm.py -- the examples/cpp/DLRM driver, launched from Python with the reference's own flags.

  python dlrm_flexflow_amd/run_dlrm.py -ll:gpu 8 -b 32768 --arch-sparse-feature-size 128 --arch-embedding-size ... \
         --arch-mlp-bot 13-512-256-128 --arch-mlp-top 3456-1024-1024-512-256-1 --epochs 2
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 dlrm_flexflow_amd/run_dlrm.py <flags>

[ref: examples/cpp/DLRM/run_random.sh:3 -- one command, `-ll:gpu N`; src/runtime/cpp_driver.cc:22-44]

One process per GPU.  With `-ll:gpu N` (N > 1) and no WORLD_SIZE in the environment this process starts N ranks of
itself BEFORE anything initialises a GPU, waits for them and exits non-zero if any fails; under torchrun it is one of
the ranks.  A rank bootstraps torch.distributed ("nccl" = RCCL), hands the C++ host layer an RCCL communicator
(comm.RcclComm; `--torch-collectives` keeps the torch.distributed callbacks) and runs the C++ driver
(`DLRMApp::run_epochs`: warm-up iteration, timed epochs, the reference's THROUGHPUT line from rank 0).
The `dlrm` binary does the same without Python (host/launcher.cc).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def gpus_requested(argv) -> int:
    n = 0
    for i, a in enumerate(argv[:-1]):
        if a == "-ll:gpu":
            n = int(argv[i + 1])
    return n


def spawn(n: int, argv) -> int:
    """Parent: never imports torch, never touches a GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        if any(rc not in (None, 0) for rc in rcs):
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()                      # the ranks started above, by handle
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    bad = [i for i, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"run_dlrm.py: rank(s) {bad} failed (exit codes {rcs})\n")
        return 1
    return 0


def rank_main(argv) -> int:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu = "--backend" in argv                       # an explicit kernel library (the tests' CPU oracle): gloo, host buffers
    torch_coll = "--torch-collectives" in argv
    argv = [a for a in argv if a != "--torch-collectives"]
    import torch
    import torch.distributed as dist
    from dlrm_flexflow_amd import ffmodel
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        from dlrm_flexflow_amd.comm import RcclComm, TorchComm
        if cpu:
            dist.init_process_group("gloo")
            comm = TorchComm(on_gpu=False)
        else:
            if not torch.cuda.is_available():
                raise SystemExit("run_dlrm.py: no GPU (the product path has no CPU fallback)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            comm = TorchComm(on_gpu=True)
            if not torch_coll:
                try:
                    comm = RcclComm(comm, own_bucket_channel="--allreduce-shared-channel" not in sys.argv)
                except Exception as e:  # noqa: BLE001  every rank raises together: all keep the torch callbacks
                    if rank == 0:
                        print("run_dlrm: direct RCCL not used:", e, file=sys.stderr, flush=True)
    flags = list(argv) + ([] if cpu else ["--device", str(local_rank)])
    app = ffmodel.DLRM(flags, comm=comm.struct if comm is not None else None)
    app.run_epochs()                                # prints the reference's THROUGHPUT line on rank 0
    app.close()
    sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main() -> int:
    argv = sys.argv[1:]
    n = gpus_requested(argv)
    if n > 1 and "WORLD_SIZE" not in os.environ:
        return spawn(n, argv)
    return rank_main(argv)


if __name__ == "__main__":
    sys.exit(main())
