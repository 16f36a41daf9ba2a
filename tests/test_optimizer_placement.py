"""Round 4: any optimizer x any placement (VERDICT r3 "What's missing" 1).

The reference runs sgd_update / adam_update over every parameter in every placement [ref: src/runtime/optimizer_kernel.cu:23-41,
206-226; src/runtime/optimizer.cc:93-189,248-330].  Here:
  * default for momentum / weight-decay SGD and Adam: the reference's own dense path on the rank(s) that hold a table (owner-local
    dense gradient, dense per-table state, no all-reduce) -- now also behind the exchange (it used to abort);
  * --sparse-embedding-optimizer: the touched-rows rule on the sorted segments of the fused update with per-row state
    (ffh_sparse_opt: lazy semantics, the stated divergence), ABI 8.
CPU tests: the oracle kernels behind the product's host code and collectives (gloo).  GPU twins: tests/test_gpu_round4.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

import dlrm_helpers as H
from conftest import golden
from dlrm_flexflow_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dist_worker.py")


def _run_ranks(world, tmp_path, mode):
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


# ---------------------------------------------------------------------------------------------------------------------
# the kernel-level rule (oracle): element arithmetic = the dense optimizers', touched rows only
# ---------------------------------------------------------------------------------------------------------------------
def _case(seed=0, R=50, D=8, B=64, Lb=2, hot=True):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, R, (B, Lb))
    if hot:
        idx[: B // 2, 0] = 3          # a row hit many times: its sum crosses 32-blocks of the sorted list
    g = rng.normal(0, 1, (B, D)).astype(np.float32)
    w = rng.normal(0, 0.3, (R, D)).astype(np.float32)
    return idx, g, w


def test_sparse_adam_rule_equals_dense_adam_on_touched_rows_and_leaves_the_rest(oracle):
    """One step from zero state, wd = 0: the touched-rows Adam equals embed_backward + adam_update over the whole table on every
    row -- bit for bit on rows hit once (same single gradient), 1e-6 on rows hit several times (canonical order vs b order) -- and
    untouched rows keep weight and state in both."""
    idx, g, w = _case()
    R, D = w.shape
    st = oracle.AdamState(alpha=0.01); st.next()
    opt = capi.SparseOpt(capi.SPARSE_OPT_ADAM, st.alpha_t, 0.0, 0.0, 0, 0.9, 0.999, 1e-8)
    w1, m1, v1 = oracle.embedding_bwd_opt(idx, g, w, opt, np.zeros_like(w), np.zeros_like(w))
    wg = oracle.embedding_bwd_dense(idx, g, R)
    wd_, md, vd = oracle.adam_update(w, wg, np.zeros_like(w), np.zeros_like(w), st)
    cnt = np.bincount(idx.reshape(-1), minlength=R)
    once = cnt == 1
    assert once.any() and (cnt > 32).any() and (cnt == 0).any()
    assert np.array_equal(w1[once], wd_[once]) and np.array_equal(m1[once], md[once]) and np.array_equal(v1[once], vd[once])
    np.testing.assert_allclose(w1, wd_, rtol=1e-6, atol=1e-7)
    assert np.array_equal(w1[cnt == 0], w[cnt == 0]) and not m1[cnt == 0].any() and not v1[cnt == 0].any()


def test_sparse_adam_rule_three_steps_vs_torch_sparse_adam(oracle):
    """The lazy semantics are torch.optim.SparseAdam's (moments and weights of untouched rows stay; bias correction by the global
    step; epsilon outside the square root): three steps with changing ids, 1e-5."""
    import torch
    idx0, g0, w = _case(1)
    R, D = w.shape
    p = torch.nn.Parameter(torch.from_numpy(w.copy()))
    topt = torch.optim.SparseAdam([p], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    st = oracle.AdamState(alpha=0.01)
    wk, mk, vk = w.copy(), np.zeros_like(w), np.zeros_like(w)
    for step in range(3):
        idx, g, _ = _case(10 + step, hot=step != 1)
        st.next()
        opt = capi.SparseOpt(capi.SPARSE_OPT_ADAM, st.alpha_t, 0.0, 0.0, 0, 0.9, 0.999, 1e-8)
        wk, mk, vk = oracle.embedding_bwd_opt(idx, g, wk, opt, mk, vk)
        flat = torch.from_numpy(idx.reshape(-1))
        vals = torch.from_numpy(np.repeat(g, idx.shape[1], axis=0))
        p.grad = torch.sparse_coo_tensor(flat[None], vals, (R, D))
        topt.step()
        np.testing.assert_allclose(wk, p.detach().numpy(), rtol=1e-5, atol=1e-6, err_msg=f"step {step}")


@pytest.mark.parametrize("nesterov,wd", [(False, 0.0), (True, 0.0), (False, 1e-2)])
def test_sparse_momentum_rule_vs_float64_restatement(oracle, nesterov, wd):
    """sgd_update's statements [ref: src/runtime/optimizer_kernel.cu:23-41] on the touched rows, three steps, against float64."""
    _, _, w = _case(2)
    R, D = w.shape
    wk, vk = w.copy(), np.zeros_like(w)
    we, ve = w.astype(np.float64), np.zeros(w.shape)
    for step in range(3):
        idx, g, _ = _case(20 + step)
        opt = capi.SparseOpt(capi.SPARSE_OPT_SGD_MOMENTUM, 0.05, wd, 0.9, int(nesterov), 0, 0, 0)
        wk, vk, _ = oracle.embedding_bwd_opt(idx, g, wk, opt, vk, None)
        G = np.zeros((R, D)); np.add.at(G, idx.reshape(-1), np.repeat(g.astype(np.float64), idx.shape[1], axis=0))
        rows = np.unique(idx)
        gt = G[rows] + wd * we[rows]
        ve[rows] = ve[rows] * 0.9 + gt
        gt = gt + 0.9 * ve[rows] if nesterov else ve[rows]
        we[rows] -= 0.05 * gt
        np.testing.assert_allclose(wk, we, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(vk, ve, rtol=1e-5, atol=1e-6)


def test_sparse_sgd_kind_equals_the_fused_update_bit_for_bit_and_bad_args_are_named(oracle):
    idx, g, w = _case(3)
    opt = capi.SparseOpt(capi.SPARSE_OPT_SGD, 0.01, 0.0, 0.0, 0, 0, 0, 0)
    w1, _, _ = oracle.embedding_bwd_opt(idx, g, w, opt)
    assert np.array_equal(w1, oracle.embedding_bwd_sgd_fused(idx, g, w, 0.01))
    for bad in (capi.SparseOpt(7, 0.01, 0, 0, 0, 0, 0, 0), capi.SparseOpt(capi.SPARSE_OPT_SGD, 0.01, 0.1, 0, 0, 0, 0, 0),
                capi.SparseOpt(capi.SPARSE_OPT_ADAM, 0.01, 0, 0, 0, 0.9, 0.999, 1e-8)):      # the last: Adam without state
        with pytest.raises(RuntimeError):
            oracle.embedding_bwd_opt(idx, g, w, bad)


# ---------------------------------------------------------------------------------------------------------------------
# whole model, one rank: dense path = the reference's sweep (torch restatement), sparse path = the lazy rule
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,path", [("adam", "sparse"), ("mom", "sparse"), ("mom", "dense")])
def test_one_rank_model_vs_torch_restatement(kind, path):
    kw = dict(adam=H.ADAM_HP) if kind == "adam" else dict(sgd=H.MOM_HP)
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=True, extra_argv=["--sparse-embedding-optimizer"] if path == "sparse" else [], **kw)
    recs = H.run_steps(m, h, 3)
    exp = H.torch_optimizer_reference(h["g"], 3, "adam" if kind == "adam" else "sgd", path == "sparse", **(H.ADAM_HP if kind == "adam" else H.MOM_HP))
    for step in range(3):
        assert set(recs[step]) == set(exp[step])
        for k in recs[step]:
            np.testing.assert_allclose(recs[step][k], exp[step][k], rtol=2e-5, atol=2e-6, err_msg=f"step {step} {k}")
    if path == "sparse":       # rows no batch touched never moved
        g = h["g"]
        t0 = recs[2]["emb.1.weight"] - g["init/emb.1.weight"]
        touched = np.zeros(t0.shape[0], bool); touched[np.unique(g["sparse1"])] = True
        assert (~touched).any() and not t0[~touched].any() and np.abs(t0[touched]).max() > 1e-5
    m.close()


# ---------------------------------------------------------------------------------------------------------------------
# two / four ranks (gloo): every placement under Adam and momentum, dense and sparse, equals the one-rank run
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,mode", [(2, "opt:adam:dense:table"), (2, "opt:adam:sparse:table"), (2, "opt:mom:sparse:table"), (2, "opt:mom:dense:table"),
                                        (2, "opt:adam:dense:column"), (2, "opt:adam:sparse:column"), (4, "opt:adam:dense:row"), (2, "opt:adam:sparse:row"),
                                        (2, "opt:adam:sparse:mixed"), (4, "opt:adam:dense:mixed")])
def test_multi_rank_any_optimizer_any_placement_equals_one_rank(tmp_path, world, mode):
    """Adam / momentum SGD on table-wise, column-wise, row-wise and mixed (small tables data-parallel) placements over 2 or 4
    gloo ranks: used to abort ("multi-rank runs need the fused embedding update").  Every rank's slice of every table, the MLP
    replicas and the predictions equal the one-rank run of the same optimizer and path after 3 steps."""
    _, okind, path, place = mode.split(":")
    _run_ranks(world, tmp_path, mode)
    kw = dict(adam=H.ADAM_HP) if okind == "adam" else dict(sgd=H.MOM_HP)
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False, extra_argv=["--sparse-embedding-optimizer"] if path == "sparse" else [], **kw)
    ref = H.run_steps(m, h, 3)
    m.close()
    g = h["g"]
    B, D, rows = int(g["B"]), int(g["D"]), list(g["rows"])
    seen = {}
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        sl = slice(r * B // world, (r + 1) * B // world)
        for step in range(3):
            np.testing.assert_allclose(z[f"s{step}/pred"], ref[step]["pred"][sl], rtol=2e-5, atol=2e-6, err_msg=f"rank {r} step {step}")
            np.testing.assert_allclose(z[f"s{step}/top.0.weight"], ref[step]["top.0.weight"], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(z[f"s{step}/bot.0.bias"], ref[step]["bot.0.bias"], rtol=2e-5, atol=2e-6)
        for t in range(len(rows)):
            key = f"s2/emb.{t}.weight"
            if key not in z.files:
                continue
            got, full = z[key], ref[2][f"emb.{t}.weight"]
            if got.shape == full.shape:
                exp = full
            elif got.shape[1] != D:                       # column block of the giant table
                c = got.shape[1]
                exp = full[:, r * c:(r + 1) * c]
            else:                                          # row block
                exp = full[rows[t] * r // world:rows[t] * (r + 1) // world]
            assert got.shape == exp.shape
            np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-6, err_msg=f"rank {r} table {t}")
            seen.setdefault(t, []).append(r)
        assert int(z["allreduce_calls"]) == 3
    assert sorted(seen) == list(range(len(rows)))
    if place == "table":
        assert all(len(v) == 1 for v in seen.values())    # sole owner: nothing replicated, nothing all-reduced


# ---------------------------------------------------------------------------------------------------------------------
# reduction depths that are not a multiple of 64 (MLPerf's 479-wide first top layer): padded operands, same values
# ---------------------------------------------------------------------------------------------------------------------
MLPERF_SMALL = ["-b", "64", "--arch-sparse-feature-size", "128", "--arch-embedding-size", "-".join(["100"] * 26), "--arch-mlp-bot", "13-64-128",
                "--arch-mlp-top", "479-1024-64-1", "--arch-interaction-op", "dot-tril", "--data-size", "64"]


@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_padded_reduction_depth_is_invisible(opt):
    """FFModel::allocate step 4a gives the dot interaction's 479-column output (and its gradient) a leading dimension of 512 and
    stores the 1024 x 479 kernel as 1024 x 512, pad columns zero, so that the persistent GEMMs serve the layer.  The pads add exact
    zeros at the end of every k sum: three steps must leave the SAME BITS as --no-pad-linear-k in every parameter (returned in
    the reference-visible [out][in] shape) and in the predictions -- under SGD and under Adam (whose moments of the pads stay 0)."""
    from dlrm_flexflow_amd import ffmodel
    outs = []
    for flags in ([], ["--no-pad-linear-k"]):
        app = ffmodel.DLRM(["--backend", H.oracle_backend(), "--optimizer", opt] + MLPERF_SMALL + flags)
        app.warmup(); app.train_steps(3, trace=False); app.model.sync()
        m = app.model
        rec = {f"{m.layer_name(l)}/{i}": m.parameter(l, i).get_weights() for l in range(m.num_layers) for i in range(m.layer_num_weights(l))}
        rec["pred"] = m.layer_output(m.num_layers - 1).get()
        inter = [l for l in range(m.num_layers) if m.layer_name(l).startswith("DotInteraction")][0]
        rec["interaction"] = m.layer_output(inter).get()
        rec["interaction_grad"] = m.layer_output(inter).get_grad()
        rec["ld"] = np.array(m.layer_output(inter).ld)
        outs.append(rec)
        app.close()
    assert int(outs[0]["ld"]) == 512 and int(outs[1]["ld"]) == 479
    first_top = [k for k in outs[0] if k.endswith("/0") and outs[0][k].shape == (1024, 479)]
    assert len(first_top) == 1
    for k in outs[0]:
        if k != "ld":
            assert outs[0][k].shape == outs[1][k].shape and outs[0][k].tobytes() == outs[1][k].tobytes(), k


@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_bias_gradient_from_the_upper_layers_data_gradient_is_only_a_placement(opt):
    adam = H.ADAM_HP if opt == "adam" else None
    """ABI 10 (ffh_linear_bwd_set_dx_colsum): in a Linear -> Linear chain the lower layer's bias gradient is taken as the column sums of
    the data gradient the upper layer stores (its final dy), and the lower layer's call gets db = NULL.  On the oracle the sums are the
    same ascending chain over the same values: three steps keep their bits against --no-dx-colsum in every parameter and in the
    predictions, and the request is really taken (the flag is what differs, not a silent fallback)."""
    outs = []
    for flags in ([], ["--no-dx-colsum"]):
        m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=True, extra_argv=flags, adam=adam)
        outs.append(H.run_steps(m, h, 3))
        m.close()
    for step in range(3):
        for k in outs[0][step]:
            assert outs[0][step][k].tobytes() == outs[1][step][k].tobytes(), (step, k)
    # the oracle takes the request whenever the call qualifies: the chain of the golden model has such calls
    from oracle import oracle
    be = oracle.lib()
    B, IN, OUT = 8, 12, 4
    x = np.ones((B, IN), np.float32); w = np.ones((OUT, IN), np.float32); y = np.ones((B, OUT), np.float32); dy = np.ones((B, OUT), np.float32)
    dx = np.zeros((B, IN), np.float32); dw = np.zeros((OUT, IN), np.float32); db = np.zeros(OUT, np.float32); cs = np.zeros(IN, np.float32)
    from dlrm_flexflow_amd import capi
    assert be.lib.ffh_linear_bwd_set_dx_colsum(be.ctx, capi.ptr(cs), IN) == capi.FFH_OK
    be.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_NONE, capi.LINEAR_DX_OVERWRITE, None, None)
    assert be.lib.ffh_linear_dx_colsum_used(be.ctx) == 1 and np.all(cs == B * OUT)
    assert be.lib.ffh_linear_bwd_set_dx_colsum(be.ctx, None, IN) == -1
    assert be.lib.ffh_linear_bwd_set_dx_colsum(be.ctx, capi.ptr(cs), 0) == -1
    # without DX_OVERWRITE (an accumulated dx is not the lower layer's final dy) the request is dropped, not taken
    assert be.lib.ffh_linear_bwd_set_dx_colsum(be.ctx, capi.ptr(cs), IN) == capi.FFH_OK
    be.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_NONE, 0, None, None)
    assert be.lib.ffh_linear_dx_colsum_used(be.ctx) == 0 and np.all(cs == B * OUT)
