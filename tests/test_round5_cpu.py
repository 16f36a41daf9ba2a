"""CPU tests added in round 5 (host logic on the oracle backend; no GPU).

  * ffh_mlp_chain_fwd / _bwd of the oracle are exactly the per-layer calls include/ff_hip.h states (bit for bit);
  * the host layer forms chains of narrow Linear layers, runs them through the chain entry points (counters) and computes the
    same bits as with --no-mlp-chain; the chain stands back in the modes it does not serve.
"""
import numpy as np
import pytest

from dlrm_flexflow_amd import capi, ffmodel
import dlrm_helpers as H

RELU, NONE, SIG = capi.AC_MODE_RELU, capi.AC_MODE_NONE, capi.AC_MODE_SIGMOID


@pytest.mark.parametrize("widths,acts,premasked,dxflags", [
    ((13, 64, 32, 16), (RELU, RELU, RELU), False, None),
    ((20, 48, 24), (NONE, SIG), False, capi.LINEAR_DX_OVERWRITE),
    ((20, 48, 24, 8), (RELU, NONE, RELU), True, capi.LINEAR_DX_MASK_BY_X),
])
def test_oracle_chain_entries_are_the_per_layer_calls(oracle, widths, acts, premasked, dxflags):
    o = oracle.lib()
    rng = np.random.default_rng(sum(widths))
    n, B = len(widths) - 1, 37
    ws = [rng.uniform(-1, 1, (b, a)).astype(np.float32) for a, b in zip(widths[:-1], widths[1:])]
    bs = [rng.uniform(-1, 1, b).astype(np.float32) for b in widths[1:]]
    x = np.maximum(rng.uniform(-1, 1, (B, widths[0])), 0).astype(np.float32)
    ys = [np.zeros((B, w), np.float32) for w in widths[1:]]
    layers = o.chain_layers([dict(w=ws[l], bias=bs[l], y=ys[l], in_dim=widths[l], out_dim=widths[l + 1], activation=acts[l]) for l in range(n)])
    o.check(o.lib.ffh_mlp_chain_fwd(o.ctx, capi.ptr(x), widths[0], layers, n, B, None), "fwd")
    cur = x
    for l in range(n):
        cur = oracle.linear_fwd(cur, ws[l], bs[l], acts[l])
        assert cur.tobytes() == ys[l].tobytes(), f"y{l}"
    g = rng.uniform(-1, 1, (B, widths[-1])).astype(np.float32)
    dys = [np.full((B, w), 3.0, np.float32) for w in widths[1:]]
    dys[-1] = g.copy()
    dws, dbs = [np.zeros_like(w) for w in ws], [np.zeros_like(b) for b in bs]
    dx0 = rng.uniform(-1, 1, (B, widths[0])).astype(np.float32)
    dx = dx0.copy() if dxflags is not None else None
    layers = o.chain_layers([dict(w=ws[l], y=ys[l], dy=dys[l], dw=dws[l], db=dbs[l], in_dim=widths[l], out_dim=widths[l + 1], activation=acts[l]) for l in range(n)])
    flags = (capi.LINEAR_DY_PREMASKED if premasked else 0) | (dxflags or 0)
    o.check(o.lib.ffh_mlp_chain_bwd(o.ctx, capi.ptr(x), widths[0], capi.ptr(dx), widths[0], layers, n, B, flags, None), "bwd")
    dy = g
    for l in range(n - 1, -1, -1):
        xin = x if l == 0 else ys[l - 1]
        f = (capi.LINEAR_DY_PREMASKED if (premasked if l == n - 1 else acts[l] == RELU) else 0)
        if l == 0:
            f |= (dxflags or 0) if dx is not None else capi.LINEAR_ONLY_DW
        else:
            f |= capi.LINEAR_DX_OVERWRITE | (capi.LINEAR_DX_MASK_BY_X if acts[l - 1] == RELU else 0)
        dxl, dwl, dbl, dya = oracle.linear_bwd_ex(xin, ys[l], dy, ws[l], acts[l], f, dx0=dx0 if l == 0 else None)
        assert dwl.tobytes() == dws[l].tobytes() and dbl.tobytes() == dbs[l].tobytes(), f"dw / db {l}"
        assert dya.tobytes() == dys[l].tobytes(), f"dy{l}"
        if l == 0 and dx is not None:
            assert dxl.tobytes() == dx.tobytes(), "dx"
        dy = dxl


def _run(args, steps=3, trace=False):
    app = ffmodel.DLRM(args)
    app.warmup()
    app.train_steps(steps, trace=trace)
    m = app.model
    m.sync()
    res = {l: m.parameter(l, 0).get_weights() for l in range(m.num_layers) if m.layer_num_weights(l)}
    res["pred"] = m.layer_output(m.num_layers - 1).get()
    cnt = (m.counter("mlp_chain_fwd_calls"), m.counter("mlp_chain_bwd_calls"))
    app.close()
    return res, cnt


@pytest.mark.parametrize("bot,top,inter,expect", [
    ("13-96-64-16", "48-24-1", "cat", (1, 1)),          # bottom chain (3 layers); the top's two layers stay per-layer forward, and its backward leaves one layer below the loss layer
    ("13-64-32-16", "48-40-24-1", "cat", (2, 2)),       # top chain backward = 48-40-24 (the click layer stays with the fused loss launch)
    ("13-600-16", "48-40-24-1", "cat", (1, 1)),         # a 600-wide layer: no bottom chain
])
def test_host_layer_runs_narrow_mlps_as_chains_same_bits_as_per_layer(bot, top, inter, expect):
    args = ["--backend", H.oracle_backend(), "-b", "48", "--arch-sparse-feature-size", "16", "--arch-embedding-size", "30-11",
            "--arch-mlp-bot", bot, "--arch-mlp-top", top, "--data-size", "48", "--arch-interaction-op", inter, "--mlp-chain-fwd-min-batch", "1"]
    steps = 3
    a, ca = _run(args, steps)
    b, cb = _run(args + ["--no-mlp-chain"], steps)
    assert cb == (0, 0)
    per_step = (ca[0] // (steps + 1), ca[1] // (steps + 1))       # warm-up + steps
    assert per_step == expect, (ca, expect)
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_chains_stand_back_where_they_are_not_served():
    base = ["--backend", H.oracle_backend(), "-b", "48", "--arch-sparse-feature-size", "16", "--arch-embedding-size", "30-11",
            "--arch-mlp-bot", "13-96-64-16", "--arch-mlp-top", "48-24-1", "--data-size", "48", "--mlp-chain-fwd-min-batch", "1"]
    for extra in (["--profiling"], ["--mlp-chain-max-batch", "32"], ["--allow-tensor-op-math-conversion"], ["--mlp-chain-max-weights", "100"]):
        _, c = _run(base + extra, 1)
        assert c == (0, 0), (extra, c)
    _, c = _run(base + ["--deterministic"], 1)      # round 6: deterministic mode takes the chains (the weight-gradient blocks are added in split order)
    assert c[0] > 0 and c[1] > 0, c
    _, c = _run(base, 2, trace=True)          # under begin_trace / end_trace as well (the oracle backend runs traces eagerly)
    assert c[0] > 0 and c[1] > 0
    _, c = _run(base[:-2], 1)                 # the forward chain waits for 4096 samples per GPU by default; the backward chain does not
    assert c[0] == 0 and c[1] > 0, c


# ---------------------------------------------------------------------------------------------------------------------------
# the MLP gradients' all-reduce in buckets issued from inside backward() [ref: one ncclAllReduce per parameter from its own update
# task, src/runtime/optimizer.cc:93-189]
import os
import subprocess
import sys

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_worker.py")


def _run_ranks(world, outdir, mode):
    os.makedirs(outdir, exist_ok=True)
    port = 31500 + (os.getpid() % 2000) + (7 if "buckets" in mode else 0) + (13 if "mixed" in mode or "replicated" in mode else 0)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(outdir)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


@pytest.mark.parametrize("world,modes", [(2, ("golden", "buckets")), (4, ("golden", "buckets")), (2, ("replicated", "buckets-mixed"))])
def test_bucketed_allreduce_equals_the_single_bucket(tmp_path, world, modes):
    """2 / 4 gloo ranks on the oracle kernels, two steps: one all-reduce of the whole slab in update() (the default with a transport
    whose calls block the host) against buckets of >= 64 gradients issued from inside backward() with the biggest layer's weight gradient
    in two row blocks -- an element's sum over the ranks does not depend on the bucket it travels in: same bits with two ranks; with
    four, gloo's ring splits a buffer into per-rank chunks whose position decides the order of the adds, so the comparison is at
    1e-6 there.  With data-parallel tables in the slab the part no bucket covers still goes through update()."""
    a_dir, b_dir = os.path.join(tmp_path, "a"), os.path.join(tmp_path, "b")
    _run_ranks(world, a_dir, modes[0])
    _run_ranks(world, b_dir, modes[1])
    for r in range(world):
        za, zb = np.load(os.path.join(a_dir, f"rank{r}.npz")), np.load(os.path.join(b_dir, f"rank{r}.npz"))
        keys = [k for k in za.files if k.startswith("s")]
        assert keys and set(keys) == {k for k in zb.files if k.startswith("s")}
        for k in keys:
            if world == 2:
                assert np.array_equal(za[k], zb[k]), k
            else:
                np.testing.assert_allclose(za[k], zb[k], rtol=1e-6, atol=1e-7, err_msg=k)
        per_step = int(zb["bucket_calls"]) // 2
        assert per_step >= 3, per_step                    # the click layer + row blocks of the biggest layer + the bottom MLP ...
        assert int(zb["allreduce_calls"]) >= int(zb["bucket_calls"]) > int(za["allreduce_calls"])


def test_single_rank_forced_exchange_with_buckets_equals_plain_run():
    """One rank, exchange path forced (identity collectives of the C++ test transport are not available on the oracle backend: the
    golden model through the Python TorchComm needs a process group) -- the bucket bookkeeping alone: every bucket is issued exactly
    once per step whatever path a layer's backward takes (chain, pair, per-layer, fused loss), and the slab optimizer sees all of them."""
    import torch.distributed as dist
    from dlrm_flexflow_amd.comm import TorchComm
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{32100 + os.getpid() % 1000}", rank=0, world_size=1)
    comm = TorchComm(on_gpu=False)
    res = []
    for extra in ([], ["--bucket-allreduce", "--allreduce-bucket-floats", "64", "--big-dw-chunks", "2", "--big-dw-min-weights", "1"]):
        m, h = H.build_golden_dlrm(H.oracle_backend(), comm=comm.struct, overlap=True, force_exchange=True, extra_argv=extra)
        recs = H.run_steps(m, h, 3)
        res.append((recs, m.counter("allreduce_bucket_calls"), m.counter("allreduce_buckets")))
        m.close()
    (ra, ca, na), (rb, cb, nb) = res
    assert ca == 0 and nb >= 3 and cb == 3 * nb, (ca, cb, nb)
    for sa, sb in zip(ra, rb):
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), k


def test_fast_path_padding_rule_is_the_same_in_both_libraries(oracle):
    """ffh_linear_fast_in_dim (the contract of include/ff_hip.h: pad the reduction depth of a wide layer to whole 64-deep k-tiles)."""
    o = oracle.lib().lib
    for (i, n), exp in {(479, 1024): 512, (857, 1024): 896, (512, 1024): 512, (13, 512): 13, (479, 100): 479, (200, 1024): 200, (3456, 1024): 3456}.items():
        assert o.ffh_linear_fast_in_dim(i, n) == exp, (i, n)


# ---------------------------------------------------------------------------------------------------------------------------
# placements and switches added late in round 5: same bits whatever the order of issue
def test_sort_placements_and_conversion_twins_do_not_change_the_bits():
    """--early-sort / --no-early-sort (the index-only sort of the table update issued behind the gather, or in front of the apply phase) against the
    default placement by shape; tensor-op mode with and without the twin-by-conversion behind an fp32-kernel layer.  The host
    logic runs on the oracle backend: what is checked is the order / bookkeeping of the calls (a sort consumed twice or not at all
    fails loudly in the library: the one-shot contract of ffh_embedding_bwd_sort_multi)."""
    base = ["--backend", H.oracle_backend(), "-b", "2304", "--arch-sparse-feature-size", "16", "--arch-embedding-size", "3000-70000-11",
            "--arch-mlp-bot", "13-64-16", "--arch-mlp-top", "22-32-1", "--data-size", "2304", "--arch-interaction-op", "dot-tril"]
    ref, _ = _run(base, 2)
    for extra in (["--early-sort"], ["--no-early-sort"]):
        got, _ = _run(base + extra, 2)
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (extra, k)
    top = ["--backend", H.oracle_backend(), "-b", "64", "--arch-sparse-feature-size", "128", "--arch-embedding-size", "300-70",
           "--arch-mlp-bot", "13-256-128", "--arch-mlp-top", "384-256-1", "--data-size", "64", "--allow-tensor-op-math-conversion"]
    a, _ = _run(top, 2)
    b, _ = _run(top + ["--no-bf16-convert-twins"], 2)
    c, _ = _run(top + ["--no-bf16-twins"], 2)
    for k in a:
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k
