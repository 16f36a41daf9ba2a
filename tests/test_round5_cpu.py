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
    for extra in (["--deterministic"], ["--profiling"], ["--mlp-chain-max-batch", "32"], ["--allow-tensor-op-math-conversion"], ["--mlp-chain-max-weights", "100"]):
        _, c = _run(base + extra, 1)
        assert c == (0, 0), (extra, c)
    _, c = _run(base, 2, trace=True)          # under begin_trace / end_trace as well (the oracle backend runs traces eagerly)
    assert c[0] > 0 and c[1] > 0
    _, c = _run(base[:-2], 1)                 # the forward chain waits for 4096 samples per GPU by default; the backward chain does not
    assert c[0] == 0 and c[1] > 0, c
