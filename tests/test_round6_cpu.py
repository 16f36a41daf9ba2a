"""Round 6, host logic on the CPU (oracle kernels, gloo): the order in which a step issues its collectives, by the kind of channel the
transport gives the MLP-gradient buckets [ref: the reference issues one ncclAllReduce per parameter from that parameter's own update task,
ordered by region dependences only: src/runtime/optimizer.cc:93-189, src/runtime/optimizer_kernel.cu:114-179].

One RCCL communicator runs its collectives in ISSUE order whatever streams they are on.  With the buckets on the all-to-alls' communicator
(ffcomm.bucket_channel_own == 0) a bucket issued before the step's backward all-to-all would put the exchange of the embedding gradients --
and the table update and the next gather behind it -- behind a weight-gradient GEMM and its all-reduce; the model therefore holds the
buckets until that all-to-all has been enqueued.  With a channel of their own (the launchers' default: ncclCommSplit, agreed by all ranks)
every bucket goes out as soon as its layers have issued their backward.
"""
import ctypes as C
import os

import numpy as np
import pytest

import dlrm_helpers as H


def _recording_comm(nonblocking, own):
    import torch.distributed as dist
    from dlrm_flexflow_amd.comm import TorchComm
    from dlrm_flexflow_amd.ffmodel import ALLREDUCE_FN
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{32300 + os.getpid() % 1000}", rank=0, world_size=1)

    class Recording(TorchComm):
        def __init__(self):
            super().__init__(on_gpu=False)
            self.log = []
            self._bk = ALLREDUCE_FN(self._bucket)
            self.struct.allreduce_bucket_sum_f32 = self._bk
            self.struct.nonblocking = nonblocking
            self.struct.bucket_channel_own = own

        def _alltoall(self, *a):
            self.log.append("alltoall")
            return super()._alltoall(*a)

        def _allreduce(self, *a):
            self.log.append("allreduce")
            return super()._allreduce(*a)

        def _bucket(self, user, buf, count, stream):
            self.log.append("bucket")
            return TorchComm._allreduce(self, user, buf, count, stream)

    return Recording()


@pytest.mark.parametrize("nonblocking", [0, 1])
def test_buckets_wait_for_the_backward_alltoall_on_a_shared_channel_and_not_on_their_own(nonblocking):
    """One gloo rank, exchange path forced, buckets forced on (tiny buckets, the biggest layer's weight gradient in two row blocks), a transport
    that records its calls.  Per step [forward all-to-all ... backward all-to-all]: on a SHARED channel no bucket lies between the two and every
    bucket follows the second; on a channel of their OWN at least one bucket (the top MLP's) precedes the backward all-to-all.  Same bits
    either way and as without buckets."""
    res = {}
    for own in (0, 1):
        comm = _recording_comm(nonblocking, own)
        extra = ["--bucket-allreduce", "--allreduce-bucket-floats", "64", "--big-dw-chunks", "2", "--big-dw-min-weights", "1"]
        m, h = H.build_golden_dlrm(H.oracle_backend(), comm=comm.struct, overlap=True, force_exchange=True, extra_argv=extra)
        nb = m.counter("allreduce_buckets")
        assert nb >= 3 and m.counter("allreduce_bucket_channel_own") == own
        comm.log.clear()
        recs = H.run_steps(m, h, 3)
        log = list(comm.log)
        m.close()
        res[own] = recs
        # cut the log into steps at every forward all-to-all (the 1st, 3rd, 5th "alltoall")
        a2a = [i for i, c in enumerate(log) if c == "alltoall"]
        assert len(a2a) == 6, log
        for s in range(3):
            fwd, bwd = a2a[2 * s], a2a[2 * s + 1]
            end = a2a[2 * s + 2] if s < 2 else len(log)
            between, after = log[fwd + 1:bwd], log[bwd + 1:end]
            assert between.count("bucket") + after.count("bucket") == nb, (own, s, log)
            if own == 0:
                assert "bucket" not in between, (s, log)                 # held until the exchange of the embedding gradients is enqueued
            else:
                assert "bucket" in between, (s, log)                     # the top MLP's buckets go out while its backward still runs
    for sa, sb in zip(res[0], res[1]):
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), k


def test_ffcomm_struct_layout_matches_the_header():
    """ffmodel.FFComm (ctypes) against host/ffcomm.h: same fields in the same order (the field appended in round 6 included)."""
    import re
    from dlrm_flexflow_amd import ffmodel
    src = open(os.path.join(H.ROOT if hasattr(H, "ROOT") else os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dlrm_flexflow_amd", "host", "ffcomm.h")).read()
    body = src[src.index("typedef struct ffcomm {"):src.index("} ffcomm;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\(\*(\w+)\)\(|(?:int|void\*)\s+(\w+);", body)
    header = [a or b for a, b in names]
    assert header == [f[0] for f in ffmodel.FFComm._fields_], (header, [f[0] for f in ffmodel.FFComm._fields_])


# ---------------------------------------------------------------------------------------------------------------------------
# the direct all-reduce of the gradient buckets (--direct-allreduce): all-to-all of 1 / N slices, local sum in rank order, all-gather
import subprocess
import sys

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_worker.py")


def _run_ranks(world, outdir, mode, port_salt):
    os.makedirs(outdir, exist_ok=True)
    port = 33500 + (os.getpid() % 2000) + port_salt
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(outdir)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


@pytest.mark.parametrize("world,mode", [(2, "buckets"), (4, "buckets"), (2, "buckets-mixed")])
def test_direct_allreduce_equals_the_ring_and_is_identical_on_every_rank(tmp_path, world, mode):
    """2 / 4 gloo ranks on the oracle kernels, two steps, buckets forced on: --direct-allreduce (all-to-all of slices, ffh_sum_slices_f32 in
    rank order, all-gather; the slices are multiples of 4 floats, so the last rank's is shorter -- or empty -- for most bucket sizes) against the transport's all-reduce -- the same sums to
    1e-6 (the order of the adds differs: rank order here, gloo's ring order there), the direct path really ran, and every rank holds the SAME
    BITS in every data-parallel parameter after each step (the sum of a slice is computed once, by its owner, and gathered)."""
    a_dir, b_dir = os.path.join(tmp_path, "ring"), os.path.join(tmp_path, "direct")
    _run_ranks(world, a_dir, mode, 3)
    _run_ranks(world, b_dir, mode + "-direct", 11)
    recs = [np.load(os.path.join(b_dir, f"rank{r}.npz")) for r in range(world)]
    for r in range(world):
        za, zb = np.load(os.path.join(a_dir, f"rank{r}.npz")), recs[r]
        assert int(za["direct_allreduces"]) == 0 and int(zb["direct_allreduces"]) >= 2 * 3, (int(za["direct_allreduces"]), int(zb["direct_allreduces"]))
        keys = [k for k in za.files if k.startswith("s")]
        assert keys and set(keys) == {k for k in zb.files if k.startswith("s")}
        for k in keys:
            np.testing.assert_allclose(za[k], zb[k], rtol=1e-6, atol=1e-7, err_msg=k)
    for k in recs[0].files:
        if k.startswith("s") and ("top" in k or "bot" in k) and all(k in z.files for z in recs):      # the MLPs' parameters: data-parallel, one copy per rank
            for r in range(1, world):
                assert recs[0][k].tobytes() == recs[r][k].tobytes(), f"{k}: ranks 0 and {r} differ"


# ---------------------------------------------------------------------------------------------------------------------------
# tensor-op mode: the backward of a small layer with a live activation derivative in exact mode (FFModel::allocate step 7, bwd_exact)
def test_tensor_op_mode_runs_the_small_live_relu_backward_in_exact_mode():
    """Host logic on the oracle kernels: bottom MLP 13-256-128 under a Concat, top 384-256-1.  With --allow-tensor-op-math-conversion the
    bottom MLP's last layer (256 -> 128: wide enough for the bf16 pipe, its dy arrives from the Concat with relu' still to apply) is the one
    layer whose backward the shim switches to exact mode; --no-bf16-exact-small-backward keeps it on the bf16 pipe.  Both are the same
    mathematics to bf16 accuracy and not the same bits; the first is closer to the all-exact run in that layer's weight gradient."""
    from dlrm_flexflow_amd import ffmodel

    def run(extra):
        args = ["--backend", H.oracle_backend(), "-b", "64", "--arch-sparse-feature-size", "128", "--arch-embedding-size", "300-70",
                "--arch-mlp-bot", "13-256-128", "--arch-mlp-top", "384-256-1", "--data-size", "64"] + extra
        app = ffmodel.DLRM(args)
        w0 = {l: app.model.parameter(l, 0).get_weights() for l in range(app.model.num_layers) if app.model.layer_num_weights(l)}
        app.warmup()
        app.train_steps(1)
        m = app.model
        m.sync()
        res = {l: m.parameter(l, 0).get_weights() - w0[l] for l in w0}
        n = m.counter("tensor_op_exact_backward_layers")
        app.close()
        return res, n

    exact, n0 = run([])
    on, n1 = run(["--allow-tensor-op-math-conversion"])
    off, n2 = run(["--allow-tensor-op-math-conversion", "--no-bf16-exact-small-backward"])
    assert (n0, n1, n2) == (0, 1, 0), (n0, n1, n2)
    layer = 1                                     # layers: 0 = 13 -> 256, 1 = 256 -> 128, then the tables, the Concat, the top MLP
    assert exact[layer].shape == (128, 256)
    assert not np.array_equal(on[layer], off[layer])
    scale = np.abs(exact[layer]).max()
    e_on, e_off = np.abs(on[layer] - exact[layer]).max() / scale, np.abs(off[layer] - exact[layer]).max() / scale
    assert e_on < 0.06 and e_off < 0.06, (e_on, e_off)          # both: the mode's accuracy on a one-step weight delta (dy itself comes through bf16 layers above)
    # the mean error of the layer's weight update: without the rounding of dy and x in this layer's own GEMM it is the smaller one
    m_on, m_off = np.abs(on[layer] - exact[layer]).mean(), np.abs(off[layer] - exact[layer]).mean()
    assert m_on < m_off, (m_on, m_off)


def test_split_mode_runs_the_narrow_chains_of_the_exact_kernels():
    """--fp32-split-bf16x3 leaves narrow layers to the exact kernels (FFH_BF16X3_MIN_FLOP), so the chain launches of round 5 serve them in that mode too
    (round 6: mlp_chain.hip chain_math_mode_ok, restated by the oracle).  Host logic on the oracle kernels: the chains run (counters), and the
    result is the per-layer calls' bit for bit; the tensor-op mode still runs none."""
    from dlrm_flexflow_amd import ffmodel

    def run(extra):
        args = ["--backend", H.oracle_backend(), "-b", "48", "--arch-sparse-feature-size", "16", "--arch-embedding-size", "30-11",
                "--arch-mlp-bot", "13-96-64-16", "--arch-mlp-top", "48-24-1", "--data-size", "48", "--mlp-chain-fwd-min-batch", "1"] + extra
        app = ffmodel.DLRM(args)
        app.warmup()
        app.train_steps(2)
        m = app.model
        m.sync()
        res = {l: m.parameter(l, 0).get_weights() for l in range(m.num_layers) if m.layer_num_weights(l)}
        res["pred"] = m.layer_output(m.num_layers - 1).get()
        cnt = (m.counter("mlp_chain_fwd_calls"), m.counter("mlp_chain_bwd_calls"))
        app.close()
        return res, cnt

    a, ca = run(["--fp32-split-bf16x3"])
    b, cb = run(["--fp32-split-bf16x3", "--no-mlp-chain"])
    c, cc = run(["--allow-tensor-op-math-conversion"])
    assert ca[0] > 0 and ca[1] > 0 and cb == (0, 0) and cc == (0, 0), (ca, cb, cc)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
