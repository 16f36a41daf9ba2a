"""GPU tests (-m gpu) added in round 4.

  * the touched-rows optimizers on the sorted segments (ABI 8) -- HIP == oracle bit for bit, every kernel path;
  * any optimizer x any placement on the HIP kernels: one rank vs the oracle backend, two ranks sharing the GPU vs one rank;
  * the one-shot contract of the sort / apply two-call form (round-3 advisor);
  * the weights' bf16 twin under a replayed step after a host write (round-3 advisor).
"""
import os
import subprocess

import numpy as np
import pytest
import torch

from dlrm_flexflow_amd import capi, ffmodel
import dlrm_helpers as H

pytestmark = pytest.mark.gpu
HIP = capi.HIP_LIB_PATH
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(kind, step=1):
    if kind == "adam":
        b1t, b2t = 0.9 ** step, 0.999 ** step
        return capi.SparseOpt(capi.SPARSE_OPT_ADAM, float(0.01 * np.sqrt(1 - b2t) / (1 - b1t)), 0.0, 0.0, 0, 0.9, 0.999, 1e-8)
    if kind == "adam-wd":
        return capi.SparseOpt(capi.SPARSE_OPT_ADAM, 0.003, 1e-2, 0.0, 0, 0.9, 0.999, 1e-8)
    if kind == "nesterov":
        return capi.SparseOpt(capi.SPARSE_OPT_SGD_MOMENTUM, 0.05, 1e-3, 0.9, 1, 0, 0, 0)
    if kind == "wd-only":
        return capi.SparseOpt(capi.SPARSE_OPT_SGD_MOMENTUM, 0.05, 1e-2, 0.0, 0, 0, 0, 0)
    return capi.SparseOpt(capi.SPARSE_OPT_SGD_MOMENTUM, 0.05, 0.0, 0.9, 0, 0, 0, 0)


@pytest.mark.parametrize("kind", ["adam", "adam-wd", "mom", "nesterov", "wd-only"])
@pytest.mark.parametrize("B,L,D,rows", [
    (32768, 1, 128, (4000000, 3, 977)),       # the tiled path: runs that cross 32- and 1024-blocks (3 rows hit ~10,000 times each), single hits
    (4096, 2, 64, (100000, 17)),              # bags of two
    (1000, 1, 16, (50, 70000)),               # the one-launch small-batch kernel
    (3000, 1, 13, (40, 5000)),                # scalar (VEC = 1) instantiations
])
def test_sparse_optimizer_rules_hip_equals_oracle_bit_for_bit(hip, oracle, kind, B, L, D, rows):
    """Three steps of ffh_embedding_bwd_opt_fused_multi (step 2 as sort + apply) with changing ids and gradients: weights and both
    state arrays equal the oracle's restatement bit for bit -- the row sums in the canonical order, the element statements those of
    sgd_update / adam_update [ref: src/runtime/optimizer_kernel.cu:23-41,206-226]."""
    rng = np.random.default_rng(B + D)
    T = len(rows)
    ws = torch.empty(hip.lib.ffh_embedding_bwd_workspace_bytes(T, L, D, B) + 256, dtype=torch.uint8, device=DEV)
    hip.set_workspace(ws, ws.numel())
    Wn = [rng.uniform(-1, 1, (r, D)).astype(np.float32) for r in rows]
    nstate = 2 if kind.startswith("adam") else (0 if kind == "wd-only" else 1)
    Sn = [[np.zeros_like(w) for _ in range(nstate)] for w in Wn]
    W = [torch.from_numpy(w).to(DEV) for w in Wn]
    S = [[torch.from_numpy(x).to(DEV) for x in st] for st in Sn]
    for step in range(3):
        In = [rng.integers(0, r, (B, L)) for r in rows]
        Gn = [rng.uniform(-1, 1, (B, D + 3)).astype(np.float32) for _ in rows]            # a leading dimension wider than the row
        opt = _opt(kind, step + 1)
        I = [torch.from_numpy(i).to(DEV) for i in In]
        G = [torch.from_numpy(g).to(DEV) for g in Gn]
        arr = hip.emb_tables([(I[t], W[t], G[t], rows[t], D + 3) for t in range(T)])
        sts = hip.emb_states([((S[t][0] if nstate > 0 else None), (S[t][1] if nstate > 1 else None)) for t in range(T)])
        import ctypes as C
        if step == 1:
            hip.check(hip.lib.ffh_embedding_bwd_sort_multi(hip.ctx, arr, T, L, D, B, None), "sort")
            hip.check(hip.lib.ffh_embedding_bwd_opt_apply_multi(hip.ctx, arr, sts, T, L, D, B, capi.AGGR_MODE_SUM, C.byref(opt), None), "apply")
        else:
            hip.check(hip.lib.ffh_embedding_bwd_opt_fused_multi(hip.ctx, arr, sts, T, L, D, B, capi.AGGR_MODE_SUM, C.byref(opt), None), "fused")
        torch.cuda.synchronize()
        for t in range(T):
            w, s0, s1 = oracle.embedding_bwd_opt(In[t], np.ascontiguousarray(Gn[t][:, :D]), Wn[t], opt, Sn[t][0] if nstate > 0 else None, Sn[t][1] if nstate > 1 else None)
            Wn[t] = w
            if nstate > 0: Sn[t][0] = s0
            if nstate > 1: Sn[t][1] = s1
            assert W[t].cpu().numpy().tobytes() == w.tobytes(), f"step {step} table {t}: weights"
            for k in range(nstate):
                assert S[t][k].cpu().numpy().tobytes() == Sn[t][k].tobytes(), f"step {step} table {t}: state {k}"


def test_sparse_optimizer_entry_points_reject_bad_arguments(hip):
    import ctypes as C
    B, D, R = 4096, 16, 100
    ws = torch.empty(hip.lib.ffh_embedding_bwd_workspace_bytes(1, 1, D, B) + 256, dtype=torch.uint8, device=DEV)
    hip.set_workspace(ws, ws.numel())
    I = torch.zeros(B, 1, dtype=torch.int64, device=DEV); W = torch.zeros(R, D, device=DEV); G = torch.zeros(B, D, device=DEV)
    arr = hip.emb_tables([(I, W, G, R, D)])
    none = hip.emb_states([(None, None)])
    for opt in (capi.SparseOpt(9, 0.01, 0, 0, 0, 0, 0, 0), capi.SparseOpt(capi.SPARSE_OPT_SGD, 0.01, 0, 0.9, 0, 0, 0, 0),
                capi.SparseOpt(capi.SPARSE_OPT_ADAM, 0.01, 0, 0, 0, 0.9, 0.999, 1e-8), capi.SparseOpt(capi.SPARSE_OPT_SGD_MOMENTUM, 0.01, 0, 0.9, 0, 0, 0, 0)):
        assert hip.lib.ffh_embedding_bwd_opt_fused_multi(hip.ctx, arr, none, 1, 1, D, B, capi.AGGR_MODE_SUM, C.byref(opt), None) == -1      # FFH_ERR_BAD_ARG
    assert hip.lib.ffh_embedding_bwd_opt_fused_multi(hip.ctx, arr, none, 1, 1, D, B, capi.AGGR_MODE_SUM, None, None) == -1
    torch.cuda.synchronize()


def test_apply_phase_is_one_shot(hip):
    """Round-3 advisor: ffh_embedding_bwd_sgd_apply_multi consumes what the sort left (sorted list + the counters only the sort
    clears).  A second apply, an apply without a sort, with another shape, or after a fused call on the same workspace used to
    run and silently skip rows whose runs cross tiles; now FFH_ERR_WORKSPACE, nothing launched."""
    B, D, R = 8192, 32, 7
    ws = torch.empty(hip.lib.ffh_embedding_bwd_workspace_bytes(1, 1, D, B) + 256, dtype=torch.uint8, device=DEV)
    hip.set_workspace(ws, ws.numel())
    I = torch.randint(0, R, (B, 1), device=DEV); G = torch.ones(B, D, device=DEV)
    W = torch.zeros(R, D, device=DEV)
    arr = hip.emb_tables([(I, W, G, R, D)])
    sort = lambda b=B: hip.lib.ffh_embedding_bwd_sort_multi(hip.ctx, arr, 1, 1, D, b, None)
    apply_ = lambda b=B: hip.lib.ffh_embedding_bwd_sgd_apply_multi(hip.ctx, arr, 1, 1, D, b, capi.AGGR_MODE_SUM, 1.0, None)
    ERR_WS = -4
    hip.set_workspace(ws, ws.numel())                 # (a fresh attach does not carry a note)
    assert sort() == 0 and apply_() == 0
    assert apply_() == ERR_WS                          # second apply on the same sort
    assert b"one apply per sort" in hip.lib.ffh_last_error_string(hip.ctx)
    assert sort() == 0 and apply_(B // 2) == ERR_WS    # another shape
    assert sort() == 0
    assert hip.lib.ffh_embedding_bwd_sgd_fused_multi(hip.ctx, arr, 1, 1, D, B, capi.AGGR_MODE_SUM, 1.0, None) == 0
    assert apply_() == ERR_WS                          # the fused call overwrote the sorted list
    ws2 = torch.empty_like(ws)
    assert sort() == 0
    hip.set_workspace(ws2, ws2.numel())
    assert apply_() == ERR_WS                          # another workspace attached
    hip.set_workspace(ws, ws.numel())
    assert apply_() == 0                               # ... the note stays with the workspace it describes
    torch.cuda.synchronize()
    counts = torch.bincount(I.reshape(-1), minlength=R).float()
    assert torch.equal(W[:, 0].cpu(), -3.0 * counts.cpu())      # exactly the three updates that were accepted: apply, fused, apply


@pytest.mark.parametrize("kind,path", [("adam", "dense"), ("adam", "sparse"), ("mom", "sparse"), ("mom", "dense")])
def test_one_rank_model_any_optimizer_hip_vs_oracle_backend(hip, kind, path):
    """The golden model under Adam / momentum SGD, tables on the reference's dense path or on the touched-rows rule: three steps on
    the HIP kernels (overlapped streams) against the same host code on the oracle (1e-5; tables of the sparse path bit for bit
    would need bit-equal gradients, which the MLP's atomics do not give)."""
    kw = dict(adam=H.ADAM_HP) if kind == "adam" else dict(sgd=H.MOM_HP)
    extra = ["--sparse-embedding-optimizer"] if path == "sparse" else []
    a, ha = H.build_golden_dlrm(HIP, overlap=True, extra_argv=extra, **kw)
    b, hb = H.build_golden_dlrm(H.oracle_backend(), overlap=False, extra_argv=extra, **kw)
    ra, rb = H.run_steps(a, ha, 3), H.run_steps(b, hb, 3)
    for step in range(3):
        for k in rb[step]:
            np.testing.assert_allclose(ra[step][k], rb[step][k], rtol=2e-5, atol=2e-6, err_msg=f"step {step} {k}")
    a.close(); b.close()


def _ranks_on_one_gpu(tmp_path, world, mode):
    worker = os.path.join(ROOT, "tests", "_dist_worker_gpu.py")
    port = 29650 + (os.getpid() % 300)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(["python", worker, str(tmp_path), "staged", mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("world,mode", [(2, "opt:adam:dense:table"), (2, "opt:adam:sparse:table"), (2, "opt:mom:sparse:column"), (4, "opt:adam:sparse:row"), (2, "opt:adam:dense:mixed")])
def test_ranks_on_one_gpu_any_optimizer_equals_one_rank(hip, tmp_path, world, mode):
    """VERDICT r3 item 3 "Done": 2 (4) ranks on one GPU (HIP kernels, host-staged collectives) under Adam / momentum, table-wise,
    column-wise, row-wise and mixed placements, equal the one-rank HIP run of the same optimizer after three steps."""
    _, okind, path, place = mode.split(":")
    z = _ranks_on_one_gpu(tmp_path, world, mode)
    kw = dict(adam=H.ADAM_HP) if okind == "adam" else dict(sgd=H.MOM_HP)
    m, h = H.build_golden_dlrm(HIP, overlap=False, extra_argv=["--sparse-embedding-optimizer"] if path == "sparse" else [], **kw)
    ref = H.run_steps(m, h, 3)
    m.close()
    g = h["g"]
    B, D, rows = int(g["B"]), int(g["D"]), list(g["rows"])
    seen = set()
    for r in range(world):
        sl = slice(r * B // world, (r + 1) * B // world)
        for step in range(3):
            np.testing.assert_allclose(z[r][f"s{step}/pred"], ref[step]["pred"][sl], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(z[r][f"s{step}/top.0.weight"], ref[step]["top.0.weight"], rtol=2e-5, atol=2e-6)
        for t in range(len(rows)):
            key = f"s2/emb.{t}.weight"
            if key not in z[r].files:
                continue
            got, full = z[r][key], ref[2][f"emb.{t}.weight"]
            if got.shape == full.shape: exp = full
            elif got.shape[1] != D: exp = full[:, r * got.shape[1]:(r + 1) * got.shape[1]]
            else: exp = full[rows[t] * r // world:rows[t] * (r + 1) // world]
            np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-6, err_msg=f"rank {r} table {t}")
            seen.add(t)
    assert seen == set(range(len(rows)))


@pytest.mark.parametrize("mode", ["tensor-op", "split"])
def test_weight_twin_is_fresh_under_a_replayed_step_after_a_host_write(hip, mode):
    """Round-3 advisor (medium): tensor-op mode + hipGraph replay -- forward() returns at once when replaying, so the bf16 twin of
    the weights was reconverted only AFTER the replayed step had used the stale one.  Kaggle widths (432->512, 512->256 read twins):
    two traced steps, then every MLP weight is overwritten from the host, then one more replayed step -- against the same
    sequence with eager launches.  (round 6) "split": the same for the three-plane image of the weights, on a model whose layers are
    big enough for the mode to take them (16384 samples, 384 -> 1024 -> 512 -> 1: 1.3e10 / 1.7e10 flop per GEMM)."""
    if mode == "tensor-op":
        args = H.KAGGLE_ARGS(2048) + ["--allow-tensor-op-math-conversion"]
    else:
        args = ["-b", "16384", "--arch-sparse-feature-size", "128", "--arch-embedding-size", "3000-700", "--arch-mlp-bot", "13-256-128",
                "--arch-mlp-top", "384-1024-512-1", "--data-size", "16384", "--fp32-split-bf16x3"]
    outs = []
    for trace in (True, False):
        app = ffmodel.DLRM(["--backend", HIP] + args + ([] if trace else ["--no-trace"]))
        app.warmup()
        app.train_steps(2, trace=trace)
        app.model.sync()
        m = app.model
        for l in range(m.num_layers):
            if m.layer_name(l).startswith("Dense"):
                p = m.parameter(l, 0)
                p.set_weights((p.get_weights() * 0.25).astype(np.float32))      # a host write to the slab: the twin is stale now
        app.train_steps(1, trace=trace)
        app.model.sync()
        outs.append(m.layer_output(m.num_layers - 1).get())
        assert bool(m.uses_graph) == trace
        app.close()
    d = np.abs(outs[0] - outs[1]).max()
    assert d < 1e-4, d          # a stale twin (weights 4x larger in the GEMMs) moves the sigmoid outputs by ~1e-1


from conftest import exact_routes


def _route(hip):
    return hip.lib.ffh_linear_last_route(hip.ctx).decode()


@pytest.mark.parametrize("B", [4096, 32768])
def test_exchange_mode_first_top_layer_backward_takes_the_persistent_kernels(hip, oracle, B):
    """Round 4: with a column map pending (the exchange path: the first top layer stores its data gradient into the bottom MLP's
    gradient and the all-to-all send buffer, ffh_linear_bwd_set_dx_scatter) BOTH GEMMs of the layer used to fall back to the
    register-staged kernels (3456 -> 1024 at 4096 samples: 658 us for the pair instead of 523).  The persistent data-gradient kernel
    now takes the map in its epilogue: route asserted, result against the oracle at 1e-5 of the term mass, destinations of a
    27-way Concat (128 columns each) plus one unaligned destination that must take the one-by-one store path."""
    import ctypes
    IN, OUT = 3456, 1024
    rng = np.random.default_rng(B)
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    dy = (rng.uniform(-1, 1, (B, OUT)) / B).astype(np.float32)
    dy[rng.uniform(0, 1, dy.shape) < 0.5] = 0.0                                   # premasked by the layer above
    xd, wd, dyd = (torch.from_numpy(a).to(DEV) for a in (x, w, dy))
    yd = torch.zeros(B, OUT, device=DEV)
    # destinations: the bottom MLP's gradient [B][128]; 25 tables' slots in a send buffer of width 25 * 128; one table alone in a
    # buffer that starts 4 bytes off a 16-byte boundary (its groups of four go one by one)
    bot = torch.full((B, 128), -7.0, device=DEV); send = torch.full((B, 25 * 128), -7.0, device=DEV); odd = torch.full((B * 128 + 1,), -7.0, device=DEV)
    ent = np.zeros((IN, 2), np.int64)
    for n in range(IN):
        if n < 128: ent[n] = (bot.data_ptr() + 4 * n, 128)
        elif n < 128 + 25 * 128: ent[n] = (send.data_ptr() + 4 * (n - 128), 25 * 128)
        else: ent[n] = (odd.data_ptr() + 4 + 4 * (n - 26 * 128), 128)
    cmap = torch.from_numpy(ent).to(DEV)
    dx = torch.full((B, IN), 3.0, device=DEV); dw = torch.zeros(OUT, IN, device=DEV); db = torch.zeros(OUT, device=DEV)
    s2 = torch.cuda.Stream()
    flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DY_PREMASKED
    hip.check(hip.lib.ffh_linear_bwd_set_dx_scatter(hip.ctx, ctypes.c_void_p(cmap.data_ptr()), IN, None), "set map")
    hip.call("ffh_linear_bwd_ex", xd, IN, dx, IN, yd, OUT, dyd, OUT, wd, dw, db, IN, OUT, B, capi.AC_MODE_RELU, flags, None, s2.cuda_stream)
    route = _route(hip)
    torch.cuda.synchronize()
    assert int(hip.lib.ffh_linear_dx_scatter_used(hip.ctx)) == 1
    assert route.count("|sk_128x128x64") == 2 and "colmap" in route, route
    got = torch.cat([bot, send, odd[1:].reshape(B, 128)], 1).cpu().numpy()
    exp = dy.astype(np.float64) @ w.astype(np.float64)
    mass = np.abs(dy).astype(np.float64) @ np.abs(w).astype(np.float64)
    assert np.all(np.abs(got - exp) <= 1e-5 * mass + 1e-9)
    assert bool((dx == 3.0).all()) and float(odd[0]) == -7.0
    dw_e = dy.astype(np.float64).T @ x.astype(np.float64)
    mdw = np.abs(dy).astype(np.float64).T @ np.abs(x).astype(np.float64)
    assert np.all(np.abs(dw.cpu().numpy() - dw_e) <= 1e-5 * mdw + 1e-9)


# ---------------------------------------------------------------------------------------------------------------------
# the benched step itself, tightened (VERDICT r3 "What's weak" 1, "Next round" 5)
# ---------------------------------------------------------------------------------------------------------------------
TERABYTE_ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155, 4, 976, 14,
                 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
TB_ARGS = lambda rows, B=32768: ["-b", str(B), "--arch-sparse-feature-size", "128", "--arch-embedding-size", "-".join(str(r) for r in rows),
                                 "--arch-mlp-bot", "13-512-256-128", "--arch-mlp-top", "3456-1024-1024-512-256-1", "--data-size", str(B)]


def _release_cached_device_memory():
    """The 96 GB / 3-step tests run late in a process whose earlier tests went through torch's caching allocator (the 204.8 GB
    table of test_gpu_round2 among them): hand the cached blocks back before the shim asks HIP for its own."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    print(f"device memory free {free / 2**30:.1f} of {total / 2**30:.1f} GiB")


def _weights(m):
    out = {}
    for li in range(m.num_layers):
        for wi in range(m.layer_num_weights(li)):
            p = m.parameter(li, wi)
            if p.is_local:
                out[f"{m.layer_name(li)}/{wi}"] = p.get_weights()
    return out


def _delta_check(name, dh, dc, tol_mass, tol_ulp, worst, explained=None, extra_tol=0.0):
    """|delta_hip - delta_cpu| <= 1e-5 of the update's term mass (+ the rounding of w itself).  An element beyond that bound must be
    EXPLAINED by the ReLU's discontinuity, element by element (round 5; round 4 only bounded how many there were): a sample whose
    pre-activation lies within one rounding error of zero gets its relu' mask 1 from one backend and 0 from the other, which adds or
    drops ONE whole term of that unit's weight-gradient row (1/16,000 of the row's mass at this batch: 6e-5) and perturbs the gradient
    that sample sends down to the tables (a row hit once carries it undiluted).  `explained` marks the elements such a flip -- derived
    from the two runs' own activations, step by step -- can reach: the rows of the flipped units, the table rows the flipped samples
    hit.  Nothing outside it may leave the 1e-5 bound; inside it the error stays within 5e-2 of the mass.  `extra_tol` widens the bound
    of an MLP weight by what the samples with a flip ABOVE its layer contribute to it (4 x their own term mass): the flip changes the
    gradient such a sample hands down by a whole term of a 128..1024-term sum, and where a weight's total mass is small that one sample
    is more than 1e-5 of it (seen: 3 of 131,072 elements of the 512 -> 256 layer, off by 2x the plain bound)."""
    err = np.abs(dh - dc)
    tol = tol_mass + tol_ulp + extra_tol
    bad = err > tol
    if explained is None:
        explained = np.zeros(err.shape, bool)
    explained = np.broadcast_to(explained, err.shape)
    unexplained = bad & ~explained
    worst[name] = (int(bad.sum()), int(unexplained.sum()), float((err / tol).max()))
    at = int(np.where(unexplained, err / tol, 0).argmax())
    msg = (f"{name}: {int(unexplained.sum())} of {err.size} deltas beyond 1e-5 of the term mass that no relu' flip explains ({int(bad.sum())} beyond it in all); "
           f"worst unexplained at flat index {at}: off by {err.flat[at]:.3e}, bound {tol.flat[at]:.3e} (delta hip {dh.flat[at]:.3e} cpu {dc.flat[at]:.3e})")
    assert not unexplained.any(), msg
    assert np.all(err <= 5000 * tol_mass + tol_ulp), msg


@pytest.mark.timeout(3000)
def test_bench_workload_three_steps_at_b32768_weight_deltas_vs_oracle(hip):
    """FOUR steps (the driver's warm-up iteration + three) of the workload bench.py times at N = 1 (26 tables, emb_dim 128, bot 13-512-256-128, top 3456-1024-1024-512-256-1,
    batch 32768, eager launches on three streams, early sort: the next gather, the sort behind it and the update are all inside
    the window) on the HIP kernels against the same host code on the oracle, compared on the weight DELTAS: a weight is ~3e-2
    and one update ~1e-4, so round 3's rtol 2e-5 on the weights saw a gradient error only above ~0.5 % of the update.  Bound per
    element (see _delta_check: the relu' mask is a discontinuity; every element beyond the bound must lie where a flip between the two runs -- derived from their own activations, step by step -- reaches): 1e-5 of the update's term mass -- lr * sum over the steps of sum_b |dy[b][o]| |x[b][i]| for an MLP weight (taken as
    steps x the last step's, computed in float64 from the oracle run's own activations and gradients), lr * sum_b |dy| for a bias, lr *
    the summed |dZ| mass of the hits for a table row -- plus two ulps of the weight per step for the rounding of w itself.  Row
    counts capped at 100,000 (the oracle's tables must fit the host; full-size tables: the next test)."""
    _release_cached_device_memory()
    rows = [min(r, 100000) for r in TERABYTE_ROWS]
    steps, lr = 4, 0.01                        # the driver's warm-up iteration + three more
    runs = {}
    for name, backend in (("hip", HIP), ("cpu", H.oracle_backend())):
        app = ffmodel.DLRM(["--backend", backend] + TB_ARGS(rows))
        m = app.model
        w0 = _weights(m)                      # before the driver's warm-up iteration (a full training step): the common starting point
        relu_layers = [li for li in range(m.num_layers) if m.layer_name(li).startswith("Dense")][:-1]        # every Dense but the click layer (sigmoid)
        masks = []                            # per step: layer -> packed (y > 0) of the step's forward: where the backends' relu' masks differ

        def snap():
            m.sync()
            masks.append({m.layer_name(li): np.packbits(m.layer_output(li).get() > 0) for li in relu_layers})
        app.warmup()
        snap()
        for _ in range(steps - 1):
            app.train_steps(1, trace=False)
            snap()
        m.sync()
        rec = {"w0": w0, "w1": _weights(m), "pred": m.layer_output(m.num_layers - 1).get(), "masks": masks}
        if name == "cpu":
            # activations and activation gradients of the LAST step: the operands of every weight gradient
            rec["x"], rec["dy"], rec["ids"] = {}, {}, {}
            for li in range(m.num_layers):
                nm = m.layer_name(li)
                if nm.startswith("Dense"):
                    rec["dy"][nm] = m.layer_output(li).get_grad()
                    rec["x"][nm] = app.dense_input().get() if li == 0 else None
                    rec["x_layer"] = rec.get("x_layer", {}); rec["x_layer"][nm] = li - 1
                if nm.startswith("Embedding"):
                    rec["ids"][nm] = app.sparse_input(len(rec["ids"])).get(np.int64)
            rec["outs"] = {m.layer_name(li): m.layer_output(li).get() for li in range(m.num_layers) if not m.layer_name(li).startswith("Embedding")}
            rec["names"] = [m.layer_name(li) for li in range(m.num_layers)]
        runs[name] = rec
        app.close()
    h, c = runs["hip"], runs["cpu"]
    assert h["w0"].keys() == c["w0"].keys()
    for k in h["w0"]:
        assert h["w0"][k].tobytes() == c["w0"][k].tobytes(), f"{k}: the two backends start from different weights"
    names = c["names"]
    # the relu' flips between the two runs, from their own activations: (sample, unit) pairs whose y > 0 differs in any of the steps
    Bn = 32768
    flip_units, flip_samples, flip_rows = {}, np.zeros(Bn, bool), {}
    nflips = {}
    concat0 = [n for n in names if n.startswith("Concat")][0]
    for nm in h["masks"][0]:
        f = np.zeros(0, bool)
        for mh, mc in zip(h["masks"], c["masks"]):
            d = np.unpackbits(mh[nm] ^ mc[nm]).astype(bool)
            f = d if f.size == 0 else (f | d)
        f = f[:Bn * (f.size // Bn)].reshape(Bn, -1)
        flip_units[nm] = f.any(0)
        nflips[nm] = int(f.sum())
        flip_rows[nm] = f.any(1)                                                      # samples with a flip in this layer
        if names.index(nm) > names.index(concat0): flip_samples |= f.any(1)          # a top-MLP flip perturbs what that sample sends down to the tables
    print("relu' flips between the backends over the steps, per layer:", nflips, "; samples with a top-MLP flip:", int(flip_samples.sum()))
    dd = lambda a: torch.from_numpy(np.abs(a)).to(DEV).double()
    # input of every Dense layer: the dense input, the layer before it, or (first top layer) the Concat output
    worst = {}
    concat = [n for n in names if n.startswith("Concat")][0]
    first_top = names[names.index(concat) + 1]
    for nm in [n for n in names if n.startswith("Dense")]:
        li = names.index(nm)
        x = c["x"][nm] if li == 0 else c["outs"][names[li - 1]]
        dy = c["dy"][nm]
        mass_w = (dd(dy).T @ dd(x)).cpu().numpy()                    # [out][in], float64
        mass_b = np.abs(dy).astype(np.float64).sum(0)
        for wi, mass in ((0, mass_w), (1, mass_b)):
            k = f"{nm}/{wi}"
            dh = h["w1"][k].astype(np.float64) - h["w0"][k].astype(np.float64)
            dc = c["w1"][k].astype(np.float64) - c["w0"][k].astype(np.float64)
            fu = flip_units.get(nm, np.zeros(dh.shape[0], bool))                      # rows (kernel) / entries (bias) of the units that flipped
            # samples with a flip in a layer whose gradient flows into this one: the Dense layers above it in its own MLP, and for the bottom
            # MLP every top layer as well
            above = np.zeros(Bn, bool)
            for n2, fr in flip_rows.items():
                i2 = names.index(n2)
                if i2 > li: above |= fr
            if above.any():
                xs, dys = np.abs(x[above]).astype(np.float64), np.abs(dy[above]).astype(np.float64)
                extra = 4.0 * lr * steps * (dys.T @ xs if wi == 0 else dys.sum(0))
            else:
                extra = 0.0
            _delta_check(k, dh, dc, 1e-5 * lr * steps * mass.reshape(dh.shape), steps * 2 * np.spacing(np.abs(c["w0"][k]).astype(np.float32)).astype(np.float64), worst,
                         explained=fu[:, None] if dh.ndim == 2 else fu, extra_tol=extra)
            assert np.abs(dc).max() > 0
    # tables: a row's update is lr * the sum of its hits' gradient rows = rows of dZ = dy1 W1[:, table's columns]
    dy1, w1 = c["dy"][first_top], c["w0"][f"{first_top}/0"]
    dzmass = dd(dy1) @ dd(w1)                                          # [B][3456] on the device, float64
    t = 0
    for nm in names:
        if not nm.startswith("Embedding"):
            continue
        k = f"{nm}/0"
        ids = torch.from_numpy(c["ids"][nm].reshape(-1)).to(DEV)
        R = h["w0"][k].shape[0]
        rowmass = torch.zeros(R, 128, dtype=torch.float64, device=DEV).index_add_(0, ids, dzmass[:, 128 * (t + 1):128 * (t + 2)]).cpu().numpy()
        dh = h["w1"][k].astype(np.float64) - h["w0"][k].astype(np.float64)
        dc = c["w1"][k].astype(np.float64) - c["w0"][k].astype(np.float64)
        hit = np.zeros(R, bool); hit[c["ids"][nm].reshape(Bn, -1)[flip_samples].reshape(-1)] = True      # rows the flipped samples hit
        _delta_check(k, dh, dc, 1e-5 * lr * steps * rowmass, steps * 2 * np.spacing(np.abs(c["w0"][k]).astype(np.float32)).astype(np.float64), worst, explained=hit[:, None])
        untouched = np.ones(R, bool); untouched[c["ids"][nm].reshape(-1)] = False
        assert not dh[untouched].any() and not dc[untouched].any()
        t += 1
    np.testing.assert_allclose(h["pred"], c["pred"], rtol=2e-5, atol=2e-6)
    print("per tensor: elements beyond 1e-5 of the mass, of those unexplained by a relu' flip, worst error / bound:", {k: (v[0], v[1], round(v[2], 2)) for k, v in sorted(worst.items(), key=lambda kv: -kv[1][2])[:10]})


def _digest_of_touched_rows(hip, app, steps_ids=None):
    """Tables too large to copy out: the rows the resident batch touches (the only rows a step writes), gathered on the device by
    the product's own gather kernel, as bytes."""
    m = app.model
    out = {}
    t = 0
    for li in range(m.num_layers):
        nm = m.layer_name(li)
        if not nm.startswith("Embedding"):
            continue
        p = m.parameter(li, 0)
        ids = app.sparse_input(t).get(np.int64)
        idt = torch.from_numpy(np.ascontiguousarray(ids)).to(DEV)
        R, D = p.dims
        rows = torch.empty(ids.shape[0], D, device=DEV)
        m.sync()
        hip.check(hip.lib.ffh_embedding_fwd(hip.ctx, idt.data_ptr(), rows.data_ptr(), p.device_ptr, ids.shape[1], D, ids.shape[0], R, D, capi.AGGR_MODE_SUM, None), "gather rows")
        torch.cuda.synchronize()
        out[nm] = rows.cpu().numpy()
        t += 1
    return out


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("full_size", [False, True, 4096, "8192-mlperf", "split", "tensor-op"])
def test_benched_step_overlapped_equals_serial_bit_for_bit_under_deterministic(hip, full_size):
    """The composition the per-layer tests cannot see -- aliasing into the Concat buffer, premasked dy across layers, forked weight
    gradients, the early sort, the next gather beside the last weight-gradient GEMM -- at the size the driver times: three steps
    under --deterministic (no floating-point atomics: same kernels => same bits), three-stream overlap with the early sort against
    --no-overlap --no-early-sort --serial-dw on one stream.  Any difference is a race or an ordering bug.  full_size: the
    uncapped Terabyte row counts (96 GB of tables; the serial run is the reference), compared on the MLP, the predictions and
    every table row the batch touches."""
    _release_cached_device_memory()
    # (round 5) 4096: the per-rank batch of the 8-GPU job, where the first top layer's data gradient runs as stream-K WITH FIX-UP -- the
    # form deterministic mode now takes (its parts are added in k order whoever arrives last): the cross-workgroup slot / counter protocol
    # runs beside the other streams' kernels here
    # (round 6) 4096 and the MLPerf shape at 8192 samples (BASELINE configs[3] per GPU): the bottom MLP's backward runs as the CHAIN launches there
    # (mlp_chain_dx_kernel / mlp_chain_dw_kernel beside the biggest weight-gradient GEMM and the table update) -- deterministic mode now takes
    # them (the splits of a weight-gradient block are added in split order by a second launch), and the test asserts that it did
    # (round 6) "split" / "tensor-op": the benched shape in the two bf16-pipe math modes -- the three-plane images / bf16 twins add writers and
    # readers on three streams (the gather's image on the side stream, the optimizer's refresh of the weights' image, conversions behind fp32-kernel
    # layers, the early sort at every batch): a stale or half-written image is exactly the kind of bug only the composition shows
    mode_flags = {"split": ["--fp32-split-bf16x3"], "tensor-op": ["--allow-tensor-op-math-conversion"]}.get(full_size, [])
    mlperf = full_size == "8192-mlperf"
    batch = 4096 if full_size == 4096 else (8192 if mlperf else 32768)
    rows = TERABYTE_ROWS if full_size is True else [min(r, 100000) for r in TERABYTE_ROWS]
    args = TB_ARGS(rows, batch)
    if mlperf:
        args = ["-b", str(batch), "--arch-sparse-feature-size", "128", "--arch-embedding-size", "-".join(str(r) for r in rows), "--arch-mlp-bot", "13-512-256-128",
                "--arch-mlp-top", "479-1024-1024-512-256-1", "--arch-interaction-op", "dot-tril", "--data-size", str(batch)]
    runs = []
    for flags in ([], ["--no-overlap", "--no-early-sort", "--serial-dw"]):
        app = ffmodel.DLRM(["--backend", HIP, "--deterministic"] + mode_flags + args + flags)
        app.warmup()
        app.train_steps(3, trace=False)
        app.model.sync()
        m = app.model
        if batch <= 8192:
            assert m.counter("mlp_chain_bwd_calls") >= 3, "the chain launches did not run under --deterministic"
        rec = {f"{m.layer_name(li)}/{wi}": m.parameter(li, wi).get_weights() for li in range(m.num_layers) if m.layer_name(li).startswith("Dense")
               for wi in range(m.layer_num_weights(li))}
        rec["pred"] = m.layer_output(m.num_layers - 1).get()
        rec.update(_digest_of_touched_rows(hip, app))
        runs.append(rec)
        app.close()
    assert runs[0].keys() == runs[1].keys() and len(runs[0]) > 30
    for k in runs[0]:
        assert runs[0][k].tobytes() == runs[1][k].tobytes(), f"{k}: overlapped and serial runs differ under --deterministic"


# ---------------------------------------------------------------------------------------------------------------------
# the per-rank batch of the 8-GPU job (4096 samples) and the MLPerf batch (8192): stream-K with fix-up (linear_sk.hip SPLIT)
# ---------------------------------------------------------------------------------------------------------------------
def _close(got, exp, mass, what, tol=1e-5):
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    bad = err > tol * mass + 1e-6
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} beyond {tol} of the term mass, worst {err.max():.3e} (mass there {mass.flat[err.argmax()]:.3e})"


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("B,IN,OUT,expect", [
    (4096, 3456, 1024, {"dx": "streamk"}),        # 864 data-gradient tiles on 256 workgroups: 3.4 rounds
    (4096, 1024, 512, {"fwd": "t64"}),            # 128 forward tiles of 128 rows = 256 of 64 rows, one per workgroup with the whole reduction (round 5; round 4: stream-K)
    (4096, 1024, 1024, {}),                       # 256 tiles: whole tiles, one per workgroup
    (8192, 512, 256, {"fwd": "t64"}),             # MLPerf batch: 128 tiles of 8 k-tiles -> 256 of 64 rows
    (8192, 1024, 256, {"fwd": "t64"}),            # 128 tiles of 16 k-tiles -> 256 of 64 rows
    (4096, 1024, 1280, {"fwd": "streamk"}),       # 320 forward tiles: 1.25 rounds -> stream-K with fix-up
    (2560, 1280, 768, {"fwd": "t64", "dx": "streamk"}),   # forward: 120 tiles -> 240 of 64 rows; data gradient 200 tiles of 12 k-tiles: ranges that end mid-tile everywhere, a short last range
])
def test_linear_layers_at_the_per_rank_batch_vs_oracle_with_routes(hip, oracle, B, IN, OUT, expect):
    """Forward, plain backward and the model's backward form (premasked dy, relu'-by-x mask, stored dX, forked dW) of the layers of
    the 4096-sample step against the oracle at 1e-5 of the term mass, with the route each call took: the uneven tile counts must
    run as stream-K with fix-up ("streamk" in the route token), and twice in a row (the second launch finds the first one's flags:
    the epoch must tell them apart)."""
    act = capi.AC_MODE_RELU
    rng = np.random.default_rng(IN + OUT + B)
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = (rng.uniform(-1, 1, (B, OUT)) / B).astype(np.float32)
    xd, wd, bd = (torch.from_numpy(a).to(DEV) for a in (x, w, b))
    ax, aw = np.abs(x).astype(np.float64), np.abs(w).astype(np.float64)
    routes = {}
    y_e = oracle.linear_fwd(x, w, b, act)
    for rep in range(2):
        y = torch.full((B, OUT), 3.0, device=DEV)
        hip.call("ffh_linear_fwd", xd, IN, y, OUT, wd, bd, IN, OUT, B, act, None)
        routes["fwd"] = _route(hip)
        _close(y.cpu().numpy(), y_e, ax @ aw.T + np.abs(b), f"{IN}->{OUT} y (launch {rep})")
    yd = torch.from_numpy(y_e).to(DEV)
    dx_e, dw_e, db_e, dy_e = oracle.linear_bwd(x, y_e, gy, w, act)
    a = np.abs(dy_e).astype(np.float64)
    m_dw, m_db, m_dx = a.T @ ax, a.sum(0), a @ aw
    flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X | capi.LINEAR_DY_PREMASKED
    dx_e2, dw_e2, db_e2, _ = oracle.linear_bwd_ex(x, y_e, dy_e, w, act, flags, dx0=None)
    s2 = torch.cuda.Stream()
    for rep in range(2):
        dy2 = torch.from_numpy(dy_e).to(DEV)
        dx2 = torch.full((B, IN), 9.0, device=DEV); dw2 = torch.zeros(OUT, IN, device=DEV); db2 = torch.zeros(OUT, device=DEV)
        hip.call("ffh_linear_bwd_ex", xd, IN, dx2, IN, yd, OUT, dy2, OUT, wd, dw2, db2, IN, OUT, B, act, flags, None, s2.cuda_stream)
        routes["bwd_ex"] = _route(hip)
        torch.cuda.synchronize()
        _close(dw2.cpu().numpy(), dw_e2, m_dw, f"{IN}->{OUT} dw (model form, launch {rep})")
        _close(db2.cpu().numpy(), db_e2, m_db, f"{IN}->{OUT} db (model form)")
        _close(dx2.cpu().numpy(), dx_e2, m_dx, f"{IN}->{OUT} dx (model form, launch {rep})")
    dy = torch.from_numpy(gy).to(DEV)
    dx = torch.zeros(B, IN, device=DEV); dw = torch.zeros(OUT, IN, device=DEV); db = torch.zeros(OUT, device=DEV)
    hip.call("ffh_linear_bwd", xd, IN, dx, IN, yd, OUT, dy, OUT, wd, dw, db, IN, OUT, B, act, None)
    routes["bwd"] = _route(hip)
    torch.cuda.synchronize()
    _close(dw.cpu().numpy(), dw_e, m_dw, f"{IN}->{OUT} dw (plain)")
    _close(dx.cpu().numpy(), dx_e, m_dx, f"{IN}->{OUT} dx (plain)")
    print(f"routes {IN}->{OUT} @{B}:", routes)
    if not exact_routes(hip):
        return                                    # the split-mode run (conftest.SPLIT_MODE_TESTS): the numbers above, other kernels
    tok = lambda r, which: [t for t in r.split(";") if which in t.split("|")[0]]
    if expect.get("fwd") == "streamk":
        assert "streamk" in routes["fwd"] and "|sk_128x128x64" in routes["fwd"], routes
    elif expect.get("fwd") == "t64":
        assert "|sk_64x128x64" in routes["fwd"] and "streamk" not in routes["fwd"], routes
    else:
        assert "streamk" not in routes["fwd"], routes
    dx_tok = tok(routes["bwd_ex"], "dx")
    assert len(dx_tok) == 1
    assert ("streamk" in dx_tok[0]) == ("dx" in expect), routes


@pytest.mark.timeout(1200)
def test_mlperf_first_top_layer_padded_to_512_hip_vs_unpadded_oracle(hip):
    """the 479-wide layer behind the dot interaction.  The shim pads the interaction output and the kernel to 512
    columns of zeros (FFModel::allocate step 4a) so that the persistent GEMMs take the layer; two steps of an MLPerf-shaped model
    (26 tables, emb_dim 128, bot 13-512-256-128, top 479-1024-1024-512-256-1, 4096 samples) on the HIP kernels with the padding
    against the oracle backend WITHOUT it, parameters in the reference-visible shapes."""
    args = ["-b", "4096", "--arch-sparse-feature-size", "128", "--arch-embedding-size", "-".join(["1000"] * 26), "--arch-mlp-bot", "13-512-256-128",
            "--arch-mlp-top", "479-1024-1024-512-256-1", "--arch-interaction-op", "dot-tril", "--data-size", "4096"]
    outs = []
    for backend, flags in ((HIP, []), (H.oracle_backend(), ["--no-pad-linear-k"])):
        app = ffmodel.DLRM(["--backend", backend] + args + flags)
        app.warmup(); app.train_steps(2, trace=False); app.model.sync()
        m = app.model
        rec = {f"{m.layer_name(l)}/{i}": m.parameter(l, i).get_weights() for l in range(m.num_layers) for i in range(m.layer_num_weights(l))}
        rec["pred"] = m.layer_output(m.num_layers - 1).get()
        inter = [l for l in range(m.num_layers) if m.layer_name(l).startswith("DotInteraction")][0]
        rec["ld"] = np.array(m.layer_output(inter).ld)
        outs.append(rec)
        app.close()
    assert int(outs[0]["ld"]) == 512 and int(outs[1]["ld"]) == 479
    for k in outs[0]:
        if k != "ld":
            np.testing.assert_allclose(outs[0][k], outs[1][k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_exchange_step_captured_as_a_graph_equals_eager_bit_for_bit(hip, tmp_path):
    """with RcclComm the collectives are stream enqueues from C++, so the per-rank step of a multi-rank job --
    kernels, both all-to-alls, the all-reduce, the side-stream branches -- can be captured and replayed as one hipGraph
    (--capture-exchange), as the reference traces every iteration on any GPU count [ref: examples/cpp/DLRM/dlrm.cc:174-181].  One
    forced RCCL rank, Kaggle shape, three steps under --deterministic: the replayed run must leave the bits of the eager one.
    (The captured step keeps its embedding branch on the compute stream: with RCCL captured on a stream that joined the capture by
    an event, ROCm 7.0's hipStreamEndCapture recurses without end -- profiles/r04_capture_exchange_endcapture_backtrace.txt.)"""
    worker = os.path.join(ROOT, "tests", "_dist_worker_gpu.py")
    outs = []
    for mode, flags in (("kaggle-graph", ["--capture-exchange", "--deterministic"]), ("kaggle", ["--deterministic", "--no-trace"])):
        d = os.path.join(tmp_path, mode); os.makedirs(d)
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90))
        r = subprocess.run(["python", worker, d, "direct", mode, *flags], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        outs.append(np.load(os.path.join(d, "rank0.npz")))
    assert int(outs[0]["uses_graph"]) == 1 and int(outs[1]["uses_graph"]) == 0
    assert int(outs[0]["alltoall_calls"]) < int(outs[1]["alltoall_calls"])      # replayed steps issue no host-side collective calls
    for k in outs[1].files:
        if k.startswith("p") or k == "pred":
            assert outs[0][k].tobytes() == outs[1][k].tobytes(), k


def test_stream_k_split_routes_under_graph_replay(hip):
    """The SPLIT GEMMs' cross-workgroup flags carry no per-launch argument (the consumer clears them), so a captured step replays
    correctly: the headline model at 4096 samples (3456->1024's dX and 1024->512's forward take the stream-K route there), four
    steps replayed from a hipGraph against four eager steps."""
    rows = [min(r, 2000) for r in TERABYTE_ROWS]
    outs = []
    for trace in (True, False):
        app = ffmodel.DLRM(["--backend", HIP] + TB_ARGS(rows, 4096) + ([] if trace else ["--no-trace"]))
        app.warmup(); app.train_steps(4, trace=trace); app.model.sync()
        assert bool(app.model.uses_graph) == trace
        rec = _weights(app.model)
        rec["pred"] = app.model.layer_output(app.model.num_layers - 1).get()
        outs.append(rec)
        app.close()
    for k in outs[0]:
        np.testing.assert_allclose(outs[0][k], outs[1][k], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("B,IN,OUT,served", [
    (4096, 1024, 512, True),        # whole tiles on the persistent kernel
    (4096, 3456, 1024, True),       # stream-K with fix-up: only the workgroup that completes a tile runs the epilogue
    (2560, 1280, 768, True),        # ranges that end mid-tile everywhere
    (512, 96, 64, False),           # not a shape of the persistent kernel: declined, colsum untouched
])
def test_lower_layers_bias_gradient_from_the_data_gradient_epilogue(hip, oracle, B, IN, OUT, served):
    """ABI 10, ffh_linear_bwd_set_dx_colsum: the data gradient a layer stores (relu'-by-x mask applied) is the lower layer's final dy,
    whose column sums are that layer's bias gradient [ref: src/ops/linear.cu:644-651].  The persistent data-gradient kernel adds them
    to `colsum` from its store epilogue; checked against the oracle's twin and against the sums of the dx the same call stored, at
    1e-5 of the term mass; one call only (the second call leaves colsum alone); a call the kernel does not serve says so."""
    act = capi.AC_MODE_RELU
    rng = np.random.default_rng(7 * IN + OUT + B)
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    y = np.maximum(rng.uniform(-1, 1, (B, OUT)), 0).astype(np.float32)
    dy = (rng.uniform(-1, 1, (B, OUT)) * (y > 0) / B).astype(np.float32)          # premasked, as the model hands it down
    flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X | capi.LINEAR_DY_PREMASKED
    xd, wd, yd = (torch.from_numpy(a).to(DEV) for a in (x, w, y))
    res = {}
    for name, be in (("hip", hip), ("oracle", oracle.lib())):
        dev = name == "hip"
        mk = (lambda a: torch.from_numpy(a).to(DEV)) if dev else (lambda a: a.copy())
        host = (lambda a: a.cpu().numpy().copy()) if dev else (lambda a: a.copy())
        dyt = mk(dy); dx = mk(np.full((B, IN), 9.0, np.float32)); dw = mk(np.zeros((OUT, IN), np.float32)); db = mk(np.zeros(OUT, np.float32))
        cs = mk(np.full(IN, 0.5, np.float32))
        xx, ww, yy = (xd, wd, yd) if dev else (x, w, y)
        be.check(be.lib.ffh_linear_bwd_set_dx_colsum(be.ctx, capi.ptr(cs), IN), "set colsum")
        be.call("ffh_linear_bwd_ex", xx, IN, dx, IN, yy, OUT, dyt, OUT, ww, dw, db, IN, OUT, B, act, flags, None, None)
        used = be.lib.ffh_linear_dx_colsum_used(be.ctx)
        route = be.lib.ffh_linear_last_route(be.ctx).decode() if dev else ""
        if dev: torch.cuda.synchronize()
        cs1 = host(cs); dx1 = host(dx)
        be.call("ffh_linear_bwd_ex", xx, IN, dx, IN, yy, OUT, dyt, OUT, ww, dw, db, IN, OUT, B, act, flags, None, None)   # the request is gone
        used2 = be.lib.ffh_linear_dx_colsum_used(be.ctx)
        if dev: torch.cuda.synchronize()
        res[name] = (used, used2, cs1, host(cs), dx1, route)
    used, used2, cs1, cs2, dx1, route = res["hip"]
    assert used2 == 0 and np.array_equal(cs1, cs2), "the request must be consumed by one call"
    if not served:
        assert used == 0 and np.all(cs1 == 0.5), (used, route)
        return
    assert used == 1 and "|colsum" in route and "|sk_128x128x64" in route, route
    mass = (np.abs(dy).astype(np.float64) @ np.abs(w).astype(np.float64)).sum(0) + 0.5
    want = 0.5 + dx1.astype(np.float64).sum(0)
    assert np.max(np.abs(cs1 - want) / mass) < 1e-5, np.max(np.abs(cs1 - want) / mass)
    o_used, o_used2, o_cs1, o_cs2, _, _ = res["oracle"]
    assert o_used == 1 and o_used2 == 0 and np.array_equal(o_cs1, o_cs2)
    assert np.max(np.abs(cs1 - o_cs1) / mass) < 1e-5


def test_stream_with_priority_is_an_ordinary_stream(hip):
    """ABI 11, ffh_stream_create_with_priority: priorities outside the device's range are clamped, the stream takes kernels and events
    like any other, a NULL result pointer is a bad argument."""
    C = capi.C
    for prio in (-1, 0, 1, -100, 100):
        s = C.c_void_p()
        hip.check(hip.lib.ffh_stream_create_with_priority(hip.ctx, C.byref(s), prio), f"create with priority {prio}")
        assert s.value
        t = torch.full((1 << 16,), 3.0, device=DEV)
        torch.cuda.synchronize()
        hip.check(hip.lib.ffh_zero(hip.ctx, capi.ptr(t), t.numel() * 4, s), "zero on the stream")
        hip.check(hip.lib.ffh_stream_sync(hip.ctx, s), "sync")
        assert float(t.abs().sum()) == 0.0
        hip.check(hip.lib.ffh_stream_destroy(hip.ctx, s), "destroy")
    assert hip.lib.ffh_stream_create_with_priority(hip.ctx, None, 0) == -1


@pytest.mark.parametrize("B,IN,OUT", [(32768, 256, 1), (20000, 256, 1), (16384, 64, 16), (17000, 1024, 3)])
def test_narrow_layer_backward_with_partial_rows_and_last_arriver(hip, B, IN, OUT):
    """Round 4: from 16384 samples up the one-launch narrow-layer backward (the click-probability layer) no longer ends every weight in
    a chain of one atomic per workgroup: the workgroups leave partial dW / db rows in ctx-owned scratch and the last one to arrive adds
    them up in block order.  dw, db (accumulated onto what the buffers held), dx against float64 at 1e-5 of the term mass; three
    launches in a row (the arrival counter must be back at 0 each time); the same bits run to run (the block order is fixed)."""
    g = torch.Generator(device="cpu"); g.manual_seed(B + IN + OUT)
    x = torch.randn(B, IN, generator=g).to(DEV); w = (torch.randn(OUT, IN, generator=g) * 0.1).to(DEV)
    y = torch.rand(B, OUT, generator=g).to(DEV); dy0 = torch.randn(B, OUT, generator=g).to(DEV)
    d64, x64, w64 = dy0.double(), x.double(), w.double()
    refs = {"dw": (0.5 + d64.t() @ x64, 0.5 + d64.abs().t() @ x64.abs()), "db": (0.25 + d64.sum(0), 0.25 + d64.abs().sum(0)),
            "dx": (d64 @ w64, d64.abs() @ w64.abs() + 1e-30)}
    seen = []
    for rep in range(3):
        dy = dy0.clone(); dx = torch.full((B, IN), 7.0, device=DEV); dw = torch.full((OUT, IN), 0.5, device=DEV); db = torch.full((OUT,), 0.25, device=DEV)
        hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_NONE, capi.LINEAR_DX_OVERWRITE, None, None)
        assert "skinny" in _route(hip)
        torch.cuda.synchronize()
        for nm, got in (("dw", dw), ("db", db), ("dx", dx)):
            ref, mass = refs[nm]
            assert float(((got.double() - ref).abs() / mass).max()) < 1e-5, (nm, rep)
        seen.append((dw.cpu().numpy().tobytes(), db.cpu().numpy().tobytes()))
    assert seen[0] == seen[1] == seen[2], "partial rows are added in block order: no run-to-run differences"
