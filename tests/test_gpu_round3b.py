"""GPU tests (-m gpu) of the two-call form of the fused table update (ABI 7).

ffh_embedding_bwd_sort_multi (index-only stable sort) + ffh_embedding_bwd_sgd_apply_multi (segmented sums, folds, SGD step) must
leave the bits ffh_embedding_bwd_sgd_fused_multi leaves -- which the other parity tests hold against the oracle's restatement of
zero_grad + embed_backward + sgd_update [ref: src/runtime/model.cc:466-490, src/ops/embedding.cu:192-217,
src/runtime/optimizer_kernel.cu:23-41] -- and the model that issues the sort behind the gather must train to the same bits as the
one that does not.
"""
import numpy as np
import pytest
import torch

from dlrm_flexflow_amd import capi, ffmodel
import dlrm_helpers as H

pytestmark = pytest.mark.gpu
HIP = capi.HIP_LIB_PATH
DEV = "cuda:0"


def _tables(hip, I, W, G, rows):
    return hip.emb_tables([(I[t], W[t], G[t], rows[t], G[t].shape[1]) for t in range(len(rows))])


@pytest.mark.parametrize("B,L,D,rows", [
    (32768, 1, 128, (4000000, 3, 977, 40000)),      # the tiled path: three / one / two / two radix passes
    (4096, 2, 64, (100000, 17)),                    # bags of two
    (1000, 1, 16, (50, 70000)),                     # small-batch path (one launch: the sort call has nothing to do)
    (2049, 1, 4, (5,)),                             # just past the small-batch limit, every row hit hundreds of times
])
def test_sort_then_apply_equals_fused_and_oracle(hip, oracle, B, L, D, rows):
    rng = np.random.default_rng(B + D)
    T = len(rows)
    ws = torch.empty(hip.lib.ffh_embedding_bwd_workspace_bytes(T, L, D, B) + 256, dtype=torch.uint8, device=DEV)
    hip.set_workspace(ws, ws.numel())
    Wn = [rng.uniform(-1, 1, (r, D)).astype(np.float32) for r in rows]
    In = [rng.integers(0, r, (B, L)) for r in rows]
    Gn = [rng.uniform(-1, 1, (B, D)).astype(np.float32) for _ in rows]
    I = [torch.from_numpy(i).to(DEV) for i in In]
    G = [torch.from_numpy(g).to(DEV) for g in Gn]
    lr = 0.05
    out = []
    for two_calls in (False, True):
        W = [torch.from_numpy(w).to(DEV) for w in Wn]
        arr = _tables(hip, I, W, G, rows)
        if two_calls:
            hip.check(hip.lib.ffh_embedding_bwd_sort_multi(hip.ctx, arr, T, L, D, B, None), "sort")
            # the gradients may change between the two calls (in a step they do not exist yet at the sort): scribble and restore
            for g, gn in zip(G, Gn):
                g.fill_(7.0)
                g.copy_(torch.from_numpy(gn))
            hip.check(hip.lib.ffh_embedding_bwd_sgd_apply_multi(hip.ctx, arr, T, L, D, B, capi.AGGR_MODE_SUM, lr, None), "apply")
        else:
            hip.check(hip.lib.ffh_embedding_bwd_sgd_fused_multi(hip.ctx, arr, T, L, D, B, capi.AGGR_MODE_SUM, lr, None), "fused")
        torch.cuda.synchronize()
        out.append([w.cpu().numpy() for w in W])
    for t in range(T):
        assert out[0][t].tobytes() == out[1][t].tobytes(), f"table {t}: sort + apply differs from the fused call"
        exp = oracle.embedding_bwd_sgd_fused(In[t], Gn[t], Wn[t], lr)
        assert out[1][t].tobytes() == exp.tobytes(), f"table {t}: differs from the oracle"


@pytest.mark.parametrize("B,L,D,rows", [
    (262144, 4, 4, (3, 70000)),        # 1 M lookups per table: 1024 blocks of 1024 -> 2048 level-1 slots, twice what the fold stages in LDS
    (40000, 16, 8, (1,)),              # every lookup on one row: one run through all 625 blocks
])
def test_folds_inside_the_reduce_launch_beyond_the_staged_slots(hip, oracle, B, L, D, rows):
    """The folds run in the tail of the reduce launch, by whichever tile / 1024-block arrives last, from slot records staged in LDS
    (1024 at a time); a table with more 1024-block slots than that reads the rest straight from memory.  Bit for bit against the
    oracle's canonical order [include/ff_hip.h, ffh_embedding_bwd_sgd_fused], SUM and AVG."""
    rng = np.random.default_rng(B + L)
    T = len(rows)
    ws = torch.empty(hip.lib.ffh_embedding_bwd_workspace_bytes(T, L, D, B) + 256, dtype=torch.uint8, device=DEV)
    hip.set_workspace(ws, ws.numel())
    Wn = [rng.uniform(-1, 1, (r, D)).astype(np.float32) for r in rows]
    In = [rng.integers(0, r, (B, L)) for r in rows]
    Gn = [rng.uniform(-1, 1, (B, D)).astype(np.float32) for _ in rows]
    I = [torch.from_numpy(i).to(DEV) for i in In]
    G = [torch.from_numpy(g).to(DEV) for g in Gn]
    for aggr in (capi.AGGR_MODE_SUM, capi.AGGR_MODE_AVG):
        W = [torch.from_numpy(w).to(DEV) for w in Wn]
        arr = _tables(hip, I, W, G, rows)
        hip.check(hip.lib.ffh_embedding_bwd_sgd_fused_multi(hip.ctx, arr, T, L, D, B, aggr, 1e-3, None), "fused")
        torch.cuda.synchronize()
        for t in range(T):
            exp = oracle.embedding_bwd_sgd_fused(In[t], Gn[t], Wn[t], 1e-3, aggr=aggr)
            assert W[t].cpu().numpy().tobytes() == exp.tobytes(), f"table {t}, aggr {aggr}: differs from the oracle"


def test_sort_call_validates_like_the_fused_call(hip):
    B, D = 4096, 16
    I = [torch.zeros(B, 1, dtype=torch.int64, device=DEV)]
    W = [torch.zeros(10, D, device=DEV)]
    G = [torch.zeros(B, D, device=DEV)]
    arr = _tables(hip, I, W, G, (10,))
    hip.set_workspace(None, 0)
    assert hip.lib.ffh_embedding_bwd_sort_multi(hip.ctx, arr, 1, 1, D, B, None) == -4      # FFH_ERR_WORKSPACE
    assert hip.lib.ffh_embedding_bwd_sort_multi(hip.ctx, arr, 1, 0, D, B, None) == -1      # FFH_ERR_BAD_ARG
    assert hip.lib.ffh_embedding_bwd_sort_multi(hip.ctx, arr, 0, 1, D, B, None) == capi.FFH_OK      # nothing to do


def _params(app):
    m = app.model
    out = {}
    for li in range(m.num_layers):
        for wi in range(m.layer_num_weights(li)):
            p = m.parameter(li, wi)
            if p.is_local:
                out[f"{m.layer_name(li)}/{wi}"] = p.get_weights()
    out["pred"] = m.layer_output(m.num_layers - 1).get()
    return out


@pytest.mark.parametrize("trace", [False, True])
def test_model_with_early_sort_trains_to_the_same_bits(hip, trace):
    """The step issues the sort behind the gather (side stream, beside the top MLP's forward) and only the apply phase where the
    whole update used to be.  --deterministic: same kernels on the same data in both runs, so any difference is an ordering bug
    (the sorted list overwritten, or read before it is complete)."""
    rows = (300000, 50, 1000000, 7, 40000)
    args = ["--backend", HIP, "-b", "16384", "--arch-sparse-feature-size", "64", "--arch-embedding-size", "-".join(map(str, rows)),
            "--arch-mlp-bot", "13-32-64", "--arch-mlp-top", "384-64-1", "--deterministic"]
    res = []
    for extra in (["--early-sort"], ["--no-early-sort"], ["--no-overlap"]):      # (by default the sort is early below 8192 samples per GPU only)
        app = ffmodel.DLRM(args + extra)
        app.warmup()
        app.train_steps(6, trace=trace)
        app.model.sync()
        res.append(_params(app))
        app.close()
    for k in res[0]:
        assert res[0][k].tobytes() == res[1][k].tobytes(), k
        assert res[0][k].tobytes() == res[2][k].tobytes(), k
