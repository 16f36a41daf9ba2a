"""CPU tests: pin the oracle (oracle/ffh_oracle.c) against
  * the committed golden vectors (tests/golden/*.npz; the embedding ones come from the
    reference's own compiled AVX2 lookup, the rest from PyTorch-CPU / numpy, the oracle
    of the reference's op tests), and
  * float64 mathematics for the pieces the reference has no test for.
Tolerances follow the reference's harness [ref: tests/ops/test_harness.py:203,302,425,504]
or tighter, and are written next to each check.
"""
import numpy as np
import pytest

from conftest import golden
from dlrm_flexflow_amd import capi


# ---------------------------------------------------------------------------
# Embedding forward: bit-exact vs the reference's own function
# ---------------------------------------------------------------------------
def test_embedding_fwd_matches_reference_golden(oracle):
    g = golden("embedding_fwd_ref")
    n = int(g["n_cases"])
    assert n == 14
    for k in range(n):
        out = oracle.embedding_fwd(g[f"c{k}_idx"], g[f"c{k}_w"])
        exp = g[f"c{k}_out"]
        # bit-exact, including the sign of zero
        assert out.tobytes() == exp.tobytes(), f"case {k} D={exp.shape[1]}"


def test_embedding_fwd_ragged_reference_golden(oracle):
    """Ragged / empty bags: the reference signature takes per-bag lengths; our ABI has a
    fixed bag size, so each bag is evaluated on its own (L = its length)."""
    g = golden("embedding_fwd_ref")
    w, flat, lens = g["ragged_w"], g["ragged_idx"], g["ragged_len"]
    exp, exp_mean = g["ragged_out"], g["ragged_out_mean"]
    pos = 0
    for b, ln in enumerate(lens):
        if ln == 0:
            assert not exp[b].any()          # an empty bag is all zeros in the reference
            continue
        idx = flat[pos:pos + ln].reshape(1, ln)
        assert oracle.embedding_fwd(idx, w).tobytes() == exp[b:b + 1].tobytes()
        got_mean = oracle.embedding_fwd(idx, w, aggr=capi.AGGR_MODE_AVG)
        assert got_mean.tobytes() == exp_mean[b:b + 1].tobytes()
        pos += ln


@pytest.mark.skipif(not __import__("oracle.oracle", fromlist=["x"]).ref_available(),
                    reason="oracle/_ref not built (needs /root/reference)")
def test_embedding_fwd_matches_live_reference(oracle):
    """Where oracle/_ref exists, compare on fresh seeded inputs too (bigger than the fixtures)."""
    rng = np.random.default_rng(123)
    for D, L in ((128, 1), (64, 4), (16, 1), (256, 2), (7, 3)):
        w = rng.uniform(-1, 1, (5000, D)).astype(np.float32)
        idx = rng.integers(0, 5000, (777, L))
        assert oracle.embedding_fwd(idx, w).tobytes() == oracle.ref_embedding_fwd(idx, w).tobytes()


def test_embedding_fwd_rejects_out_of_range(oracle):
    w = np.zeros((4, 8), np.float32)
    with pytest.raises(capi.FFHError):
        oracle.embedding_fwd(np.array([[4]]), w)
    with pytest.raises(capi.FFHError):
        oracle.embedding_fwd(np.array([[-1]]), w)


# ---------------------------------------------------------------------------
# Embedding backward (dense) and the fused backward + SGD
# ---------------------------------------------------------------------------
def test_embedding_bwd_dense_vs_float64(oracle):
    rng = np.random.default_rng(1)
    B, L, D, R = 300, 2, 16, 11
    idx = rng.integers(0, R, (B, L))
    g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    wg = oracle.embedding_bwd_dense(idx, g, R)
    exact = np.zeros((R, D))
    np.add.at(exact, idx.reshape(-1), np.repeat(g.astype(np.float64), L, axis=0))
    # fp32 running sums of ~55 terms each: 1e-5 relative to the L1 mass (north_star tolerance)
    mass = np.zeros((R, D))
    np.add.at(mass, idx.reshape(-1), np.repeat(np.abs(g.astype(np.float64)), L, axis=0))
    assert np.all(np.abs(wg - exact) <= 1e-5 * mass + 1e-12)


@pytest.mark.parametrize("R,B,L", [(3, 1000, 1), (50, 600, 2), (100000, 500, 1), (1, 700, 1)])
def test_embedding_fused_vs_float64_and_dense_path(oracle, R, B, L):
    """The fused op must equal the reference's three-step dense path
    (zero_grad -> embed_backward -> sgd_update): compare with (a) that path run through the
    oracle's own dense functions and (b) float64 maths; untouched rows keep their bits."""
    rng = np.random.default_rng(R)
    D, lr = 12, 0.01
    idx = rng.integers(0, R, (B, L))
    g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
    fused = oracle.embedding_bwd_sgd_fused(idx, g, w, lr)
    dense_grad = oracle.embedding_bwd_dense(idx, g, R)
    three_step = oracle.sgd_update(w.reshape(-1), dense_grad.reshape(-1), lr).reshape(R, D)
    exact_g = np.zeros((R, D))
    np.add.at(exact_g, idx.reshape(-1), np.repeat(g.astype(np.float64), L, axis=0))
    mass = np.zeros((R, D))
    np.add.at(mass, idx.reshape(-1), np.repeat(np.abs(g.astype(np.float64)), L, axis=0))
    exact = w.astype(np.float64) - lr * exact_g
    tol = 1e-5 * (lr * mass + np.abs(w)) + 1e-12
    assert np.all(np.abs(fused - exact) <= tol)
    assert np.all(np.abs(three_step - exact) <= tol)
    hit = np.zeros(R, bool)
    hit[idx.reshape(-1)] = True
    assert fused[~hit].tobytes() == w[~hit].tobytes()
    # rows hit exactly once in one chunk: bit-identical to the dense three-step path
    counts = np.bincount(idx.reshape(-1), minlength=R)
    once = counts == 1
    assert fused[once].tobytes() == three_step[once].tobytes()


def test_embedding_fused_canonical_order_is_two_level(oracle):
    """The documented canonical order: left-to-right sums inside 32-blocks, 32-block partials added
    left to right inside 1024-blocks, 1024-block partials added left to right.  Re-derived in numpy
    for one hot row spanning several 1024-blocks, with a second row shifting the block alignment."""
    rng = np.random.default_rng(7)
    n_hot, n_other = 2 * capi.EMB_CHUNK1 + 5 * capi.EMB_CHUNK + 17, 11
    D, lr = 4, 0.5
    idx = np.concatenate([np.zeros(n_other, np.int64), np.ones(n_hot, np.int64)]).reshape(-1, 1)   # row 0 first: row 1's run starts at sorted position 11
    B = idx.shape[0]
    g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    w = rng.uniform(-1, 1, (2, D)).astype(np.float32)
    got = oracle.embedding_bwd_sgd_fused(idx, g, w, lr)

    def fold(parts):
        acc = parts[0]
        for p in parts[1:]:
            acc = (acc + p).astype(np.float32)
        return acc

    def canonical(lo, hi):          # sorted positions [lo, hi) of one row; sorted order == batch order here
        bigs = []
        a1 = lo
        while a1 < hi:
            z1 = min((a1 // capi.EMB_CHUNK1 + 1) * capi.EMB_CHUNK1, hi)
            smalls = []
            a = a1
            while a < z1:
                z = min((a // capi.EMB_CHUNK + 1) * capi.EMB_CHUNK, z1)
                smalls.append(fold([g[q] for q in range(a, z)]))
                a = z
            bigs.append(fold(smalls))
            a1 = z1
        return fold(bigs)

    for row, (lo, hi) in enumerate(((0, n_other), (n_other, B))):
        tot = canonical(lo, hi)
        exp = np.array([np.float32(np.float64(w[row, d]) + np.float64(np.float32(-lr)) * np.float64(tot[d])) for d in range(D)], np.float32)
        assert got[row].tobytes() == exp.tobytes()


# ---------------------------------------------------------------------------
# Linear: torch golden (the reference harness's own oracle)
# ---------------------------------------------------------------------------
def test_linear_matches_torch_golden(oracle):
    g = golden("linear_torch")
    for k in range(int(g["n_cases"])):
        x, w, b, gy = g[f"c{k}_x"], g[f"c{k}_w"], g[f"c{k}_b"], g[f"c{k}_gy"]
        act = int(g[f"c{k}_act"])
        y = oracle.linear_fwd(x, w, b, act)
        # fp32 GEMM vs torch: 1e-5 relative to the magnitude (north_star tolerance)
        np.testing.assert_allclose(y, g[f"c{k}_y"], rtol=1e-5, atol=1e-5)
        dx, dw, db, _ = oracle.linear_bwd(x, y, gy, w, act)
        np.testing.assert_allclose(dw, g[f"c{k}_dw"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(db, g[f"c{k}_db"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(dx, g[f"c{k}_dx"], rtol=1e-5, atol=2e-5)
        # one SGD step (lr 0.01), what LinearTest checks [ref: tests/ops/test_harness.py:216-244]
        w_after = oracle.sgd_update(w.reshape(-1), dw.reshape(-1), 0.01).reshape(w.shape)
        np.testing.assert_allclose(w_after, g[f"c{k}_w_after"], rtol=1e-5, atol=1e-6)


def test_linear_gelu_forward_matches_torch_tanh_form(oracle):
    """AC_MODE_GELU is the tanh form with the reference's constants [ref: src/ops/linear.cu:454-459]; no backward, as there."""
    import torch
    from dlrm_flexflow_amd import capi
    rng = np.random.default_rng(14)
    x = rng.uniform(-3, 3, (50, 40)).astype(np.float32)
    w = rng.uniform(-0.5, 0.5, (24, 40)).astype(np.float32)
    b = rng.uniform(-1, 1, 24).astype(np.float32)
    y = oracle.linear_fwd(x, w, b, capi.AC_MODE_GELU)
    t = torch.nn.functional.gelu(torch.from_numpy(x) @ torch.from_numpy(w).T + torch.from_numpy(b), approximate="tanh").numpy()
    np.testing.assert_allclose(y, t, rtol=1e-5, atol=1e-5)
    with pytest.raises(Exception):
        oracle.linear_bwd(x, y, y, w, capi.AC_MODE_GELU)


def test_linear_reference_harness_shape(oracle):
    """LinearTest (10, 2000, 1000) regenerated from np.random.seed(0) with torch as the
    oracle, exactly as the reference harness does; its tolerance is mean signed error < 1e-3."""
    import torch
    np.random.seed(0)
    B, IN, OUT = 10, 2000, 1000
    x = np.random.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = np.random.uniform(-1, 1, (OUT, IN)).astype(np.float32)
    y = oracle.linear_fwd(x, w, np.zeros(OUT, np.float32))
    exp = torch.nn.functional.linear(torch.from_numpy(x), torch.from_numpy(w)).numpy()
    assert abs(float((y - exp).mean())) < 1e-3                      # the reference's own bar
    np.testing.assert_allclose(y, exp, rtol=1e-4, atol=1e-3 * 1e-1)  # and a far tighter one


# ---------------------------------------------------------------------------
# Concat / BatchMatmul / SGD / MSE / metrics
# ---------------------------------------------------------------------------
def test_concat_matches_numpy_golden(oracle):
    g = golden("concat_numpy")
    for k in range(int(g["n_cases"])):
        parts = [g[f"c{k}_in{i}"] for i in range(int(g[f"c{k}_n"]))]
        out = oracle.concat_fwd(parts)
        assert out.tobytes() == g[f"c{k}_out"].tobytes()            # pure copy: bit-exact
        back = oracle.concat_bwd(out, [p.shape[1] for p in parts])
        for p, q in zip(parts, back):
            assert np.array_equal(p, q)                             # 0 + x


def test_bmm_matches_torch_golden(oracle):
    g = golden("bmm_torch")
    for k in range(int(g["n_cases"])):
        a, b, go = g[f"c{k}_a"], g[f"c{k}_b"], g[f"c{k}_go"]
        np.testing.assert_allclose(oracle.bmm_fwd(a, b), g[f"c{k}_o"], rtol=1e-5, atol=1e-5)  # harness: 1e-5
        ga, gb = oracle.bmm_bwd(go, a, b)
        np.testing.assert_allclose(ga, g[f"c{k}_ga"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gb, g[f"c{k}_gb"], rtol=1e-5, atol=1e-5)


def test_bmm_reference_harness_large_shape(oracle):
    """(d,m,n,k) = (145,265,15,64), tolerance 1e-4 [ref: tests/ops/test_harness.py:500-510]."""
    np.random.seed(0)
    d, m, n, k = 145, 265, 15, 64
    a = np.random.uniform(0, 1, (d, n, k)).astype(np.float32)
    b = np.random.uniform(0, 1, (d, k, m)).astype(np.float32)
    np.testing.assert_allclose(oracle.bmm_fwd(a, b), np.matmul(a, b), rtol=1e-4, atol=1e-4)


def test_bmm_seq_length(oracle):
    """seq_length truncation [ref: src/ops/batch_matmul.cu:216-236]: strides stay full-size."""
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (3, 4, 6)).astype(np.float32)
    b = rng.uniform(0, 1, (3, 6, 5)).astype(np.float32)
    o = oracle.bmm_fwd(a, b, a_seq=0, b_seq=1, seq=4)               # k: 6 -> 4
    np.testing.assert_allclose(o, np.matmul(a[:, :, :4], b[:, :4, :]), rtol=1e-5, atol=1e-6)
    o = oracle.bmm_fwd(a, b, a_seq=1, b_seq=-1, seq=2)              # n: 4 -> 2 (rows 2,3 untouched = 0)
    np.testing.assert_allclose(o[:, :2], np.matmul(a[:, :2], b), rtol=1e-5, atol=1e-6)
    assert not o[:, 2:].any()


def test_sgd_and_mse_match_torch_golden(oracle):
    g = golden("sgd_mse_torch")
    for k in range(int(g["n_cases"])):
        lr, wd, mom, nest = g[f"c{k}_hp"]
        w = g[f"c{k}_w0"].copy()
        v = np.zeros_like(w) if mom > 0 else None
        for step in range(3):
            w = oracle.sgd_update(w, g[f"c{k}_g"][step], lr, wd, mom, bool(nest), v)
        np.testing.assert_allclose(w, g[f"c{k}_w3"], rtol=1e-5, atol=1e-6)
    grad = oracle.mse_bwd(g["mse_p"], g["mse_y"], 1.0 / 37)
    np.testing.assert_allclose(grad, g["mse_grad"], rtol=1e-6, atol=1e-8)
    perf = oracle.metrics_update(g["mse_p"], g["mse_y"], capi.METRIC_ACCURACY | capi.METRIC_MSE)
    assert perf.train_all == 2 * 37 and perf.train_correct == 37    # the reference's double count, 1 class
    assert abs(perf.mse_loss - float(g["mse_sum"])) <= 1e-5 * float(g["mse_sum"])


def test_embedding_backward_matches_compiled_reference(oracle):
    """Pins the dense scatter-add (a-3) and, through it, the fused update (a-3 + a-4) to the reference's own CPU
    embed_backward [ref: src/ops/embedding.cc:344-374], compiled from its source (fixture: make_golden.py,
    embedding_bwd_from_reference).  ffh_embedding_bwd_dense: same ascending-b accumulation => bit for bit.
    Fused update: W - lr * (reference gradient from zero) within 1e-6 relative (only the order inside duplicate rows
    differs), and bit for bit when no row is hit more than twice (a + b is order-free)."""
    g = golden("embedding_bwd_ref")
    for k in range(int(g["n_cases"])):
        idx, gr, wg0, wg = g[f"c{k}_idx"], g[f"c{k}_g"], g[f"c{k}_wg0"], g[f"c{k}_wg"]
        R = wg0.shape[0]
        got = oracle.embedding_bwd_dense(idx, gr, R, wgrad=wg0.copy())
        assert got.tobytes() == wg.tobytes(), f"case {k}"
        # three-step reference path with the reference's own gradient: zero_grad -> embed_backward -> sgd_update
        dense = wg.astype(np.float64) - wg0.astype(np.float64)          # the gradient itself, up to fp32 rounding of wg0 + .
        rng = np.random.default_rng(k)
        w = rng.uniform(-1, 1, wg0.shape).astype(np.float32)
        fused = oracle.embedding_bwd_sgd_fused(idx, gr, w, 0.05)
        from_zero = oracle.embedding_bwd_dense(idx, gr, R)               # bit-equal to the reference run from a zero table: checked next
        np.testing.assert_allclose(from_zero, dense, rtol=0, atol=4e-6 * max(1.0, np.abs(dense).max()))
        three_step = oracle.sgd_update(w.reshape(-1), from_zero.reshape(-1), 0.05).reshape(w.shape)
        mass = np.zeros(wg0.shape, np.float64)
        np.add.at(mass, idx.reshape(-1), np.abs(gr.astype(np.float64)))
        bound = 8 * 2.0 ** -24 * (np.abs(w) + 0.05 * mass) + 1e-12      # a few ulps of the accumulated magnitude
        assert (np.abs(fused.astype(np.float64) - three_step) <= bound).all(), f"case {k}" 
        counts = np.bincount(idx.reshape(-1), minlength=R)
        rows = counts <= 2
        assert np.array_equal(fused[rows], three_step[rows])
    if oracle.ref_available():                                             # in this container: call the compiled function itself
        idx, gr = g["c0_idx"], g["c0_g"]
        z = oracle.ref_embedding_bwd(idx, gr, np.zeros_like(g["c0_wg0"]))
        assert z.tobytes() == oracle.embedding_bwd_dense(idx, gr, z.shape[0]).tobytes()


def test_concat_bwd_overwrite_flag(oracle):
    """ffh_concat_bwd_ex: 0 = add_with_stride [ref: src/runtime/cuda_helper.cu:110-126], FFH_CONCAT_BWD_OVERWRITE = plain store."""
    import ctypes as C
    rng = np.random.default_rng(2)
    nb, widths = 5, [3, 8, 1]
    og = rng.uniform(-1, 1, (nb, sum(widths))).astype(np.float32)
    lib = oracle.lib()
    for flags in (0, capi.CONCAT_BWD_OVERWRITE):
        grads = [np.full((nb, w), 2.0, np.float32) for w in widths]
        pa = (C.c_void_p * 3)(*[g.ctypes.data for g in grads])
        ba = (C.c_int64 * 3)(*widths)
        lib.check(lib.lib.ffh_concat_bwd_ex(lib.ctx, og.ctypes.data, sum(widths), pa, ba, None, 3, nb, flags, None), "concat_bwd_ex")
        off = 0
        for g, w in zip(grads, widths):
            assert np.array_equal(g, og[:, off:off + w] + (0 if flags else np.float32(2.0)))
            off += w


def test_relu_mask_moved_to_the_producer_equals_plain_calls(oracle):
    """FFH_LINEAR_DX_MASK_BY_X on the upper layer + FFH_LINEAR_DY_PREMASKED on the lower one compute exactly what two
    plain Linear::backward calls compute (reluBackward [ref: src/runtime/cuda_helper.cu:71-78] applied where its
    operand is produced) -- bit for bit in the oracle, and equal to torch autograd through relu."""
    import torch
    rng = np.random.default_rng(11)
    B, I0, H, O = 37, 12, 20, 9
    x0 = rng.uniform(-1, 1, (B, I0)).astype(np.float32)
    w1, b1 = rng.uniform(-1, 1, (H, I0)).astype(np.float32), rng.uniform(-1, 1, H).astype(np.float32)
    w2, b2 = rng.uniform(-1, 1, (O, H)).astype(np.float32), rng.uniform(-1, 1, O).astype(np.float32)
    g2 = rng.uniform(-1, 1, (B, O)).astype(np.float32)
    y1 = oracle.linear_fwd(x0, w1, b1, capi.AC_MODE_RELU)
    y2 = oracle.linear_fwd(y1, w2, b2, capi.AC_MODE_NONE)
    # plain: upper layer hands down dy1 unmasked, lower layer masks it in place
    dy1, dw2, db2, _ = oracle.linear_bwd(y1, y2, g2, w2, capi.AC_MODE_NONE)
    dx0, dw1, db1, dy1_after = oracle.linear_bwd(x0, y1, dy1, w1, capi.AC_MODE_RELU)
    # moved: the mask is applied by the upper layer's data-gradient step
    F = capi
    dy1_m, dw2_m, db2_m, _ = oracle.linear_bwd_ex(y1, y2, g2, w2, capi.AC_MODE_NONE, F.LINEAR_DX_OVERWRITE | F.LINEAR_DX_MASK_BY_X,
                                                 dx0=np.full((B, H), 7.0, np.float32))
    dx0_m, dw1_m, db1_m, dy1_kept = oracle.linear_bwd_ex(x0, y1, dy1_m, w1, capi.AC_MODE_RELU, F.LINEAR_DY_PREMASKED)
    assert np.array_equal(dy1_m, dy1_after) and np.array_equal(dy1_kept, dy1_m)
    for a, b in ((dw2, dw2_m), (db2, db2_m), (dx0, dx0_m), (dw1, dw1_m), (db1, db1_m)):
        assert np.array_equal(a, b)
    # split calls carry the flags too
    dxs, _, _, _ = oracle.linear_bwd_ex(x0, y1, dy1_m, w1, capi.AC_MODE_RELU, F.LINEAR_DY_PREMASKED | F.LINEAR_ONLY_DX)
    _, dws, dbs, _ = oracle.linear_bwd_ex(x0, y1, dy1_m, w1, capi.AC_MODE_RELU, F.LINEAR_DY_PREMASKED | F.LINEAR_ONLY_DW)
    assert np.array_equal(dxs, dx0) and np.array_equal(dws, dw1) and np.array_equal(dbs, db1)
    t = {k: torch.tensor(v, requires_grad=True) for k, v in dict(x0=x0, w1=w1, b1=b1, w2=w2, b2=b2).items()}
    out = torch.relu(t["x0"] @ t["w1"].T + t["b1"]) @ t["w2"].T + t["b2"]
    out.backward(torch.from_numpy(g2))
    for got, key in ((dx0_m, "x0"), (dw1_m, "w1"), (db1_m, "b1"), (dw2_m, "w2"), (db2_m, "b2")):
        np.testing.assert_allclose(got, t[key].grad.numpy(), rtol=1e-5, atol=1e-5, err_msg=key)


def test_adam_matches_torch_golden(oracle):
    """adam_update [ref: src/runtime/optimizer_kernel.cu:206-226] + AdamOptimizer::next [ref: optimizer.cc:248-254]
    against torch.optim.Adam over 5 steps (fixture: tests/golden/make_golden.py:adam_from_torch)."""
    g = golden("adam_torch")
    for k in range(int(g["n_cases"])):
        alpha, b1, b2, wd, eps = g[f"c{k}_hp"]
        st = oracle.AdamState(alpha, b1, b2, wd, eps)
        w = g[f"c{k}_w0"].copy()
        m, v = np.zeros_like(w), np.zeros_like(w)
        for step in range(5):
            st.next()
            w, m, v = oracle.adam_update(w, g[f"c{k}_g"][step], m, v, st)
        np.testing.assert_allclose(w, g[f"c{k}_w5"], rtol=1e-5, atol=1e-6)
    # the statement-by-statement float32 evaluation in numpy, one step (independent of the C code)
    rng = np.random.default_rng(0)
    w0, g0 = rng.uniform(-1, 1, 1000).astype(np.float32), rng.uniform(-1, 1, 1000).astype(np.float32)
    m0, v0 = rng.uniform(-0.1, 0.1, 1000).astype(np.float32), rng.uniform(0, 0.1, 1000).astype(np.float32)
    st = oracle.AdamState(0.01, 0.9, 0.999, 1e-3, 1e-8)
    st.next()
    f = np.float32
    gt = (g0.astype(np.float64) + np.float64(f(1e-3)) * w0).astype(f)                          # single rounding = fma
    mt = (np.float64(f(0.9)) * m0 + ((f(1) - f(0.9)) * gt).astype(np.float64)).astype(f)
    vt = (np.float64(f(0.999)) * v0 + (((f(1) - f(0.999)) * gt) * gt).astype(np.float64)).astype(f)
    wn = w0 - (f(st.alpha_t) * mt) / (np.sqrt(vt) + f(1e-8))
    w1, m1, v1, g1 = oracle.adam_update(w0, g0, m0, v0, st, zero_grad=True)
    np.testing.assert_array_equal(m1, mt)
    np.testing.assert_array_equal(v1, vt)
    np.testing.assert_array_equal(w1, wn.astype(f))
    assert not g1.any()
    w2, g2 = oracle.sgd_update_zero_grad(w0, g0, 0.01)
    np.testing.assert_array_equal(w2, oracle.sgd_update(w0, g0, 0.01))
    assert not g2.any()


# ---------------------------------------------------------------------------
# RNG contract (include/ffh_rng.h)
# ---------------------------------------------------------------------------
def _mix64(z):
    M = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def test_rng_contract(oracle):
    M = (1 << 64) - 1
    seed, first, R = 42, 1000, 12345
    idx = oracle.gen_indices(64, seed, first, R)
    u = oracle.gen_uniform01(64, seed, first)
    y = oracle.gen_bernoulli(64, seed, first)
    for i in range(64):
        h = _mix64((_mix64(seed) + first + i) & M)
        assert idx[i] == h % R
        assert u[i] == np.float32((h >> 40) / 16777216.0)
        assert y[i] == float((h >> 33) & 1)
    w = oracle.init_uniform(1000, 5, -0.25, 0.25)
    assert w.min() >= -0.25 and w.max() < 0.25 and abs(float(w.mean())) < 0.02


def test_row_sharded_table_equals_whole_table(oracle):
    """The row-wise sharding arithmetic (this build's extension, SURVEY 8e 'reduce-scatter variant') on plain arrays:
    G row blocks, each followed by a zero row; ids localized per block; the blocks' partial bag sums add up to the
    whole-table gather (bit-exact for bag 1, 1e-6 for longer bags whose adds are re-associated), and per-block fused
    updates with the zero row discarded rebuild the whole-table update (1e-6: a row's gradients are the same and in the
    same order, but the canonical order cuts runs at fixed positions of the sorted list, which differ per block)."""
    rng = np.random.default_rng(3)
    R, D, B, G = 53, 8, 64, 4
    w = rng.standard_normal((R, D)).astype(np.float32)
    for L in (1, 3):
        idx = rng.integers(0, R, (B, L))
        g = rng.standard_normal((B, D)).astype(np.float32)
        whole = oracle.embedding_fwd(idx, w)
        whole_upd = oracle.embedding_bwd_sgd_fused(idx, g, w, 0.05)
        total = np.zeros((B, D), np.float32)
        rebuilt = []
        for r in range(G):
            r0, r1 = R * r // G, R * (r + 1) // G
            loc = oracle.embedding_localize_rows(idx, r0, r1 - r0)
            exp = np.where((idx >= r0) & (idx < r1), idx - r0, r1 - r0)
            assert np.array_equal(loc, exp)
            block = np.concatenate([w[r0:r1], np.zeros((1, D), np.float32)])
            total = total + oracle.embedding_fwd(loc, block)
            rebuilt.append(oracle.embedding_bwd_sgd_fused(loc, g, block, 0.05)[:-1])
        if L == 1:
            assert np.array_equal(total, whole)
        else:
            np.testing.assert_allclose(total, whole, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(np.concatenate(rebuilt), whole_upd, rtol=1e-6, atol=1e-7)
        untouched = np.setdiff1d(np.arange(R), idx.reshape(-1))
        assert np.array_equal(np.concatenate(rebuilt)[untouched], w[untouched])


@pytest.mark.parametrize("n", [2, 5, 27, 64])
def test_tril_matches_torch(oracle, n):
    """Strict lower triangle of the pairwise-dot matrix (no reference op): torch's Z[:, li, lj] with
    tril_indices(n, n, -1); the backward adds into the kept entries only.  Pure copies and x + g: bit-exact."""
    import torch
    rng = np.random.default_rng(n)
    B, P = 7, n * (n - 1) // 2
    z = rng.standard_normal((B, n, n)).astype(np.float32)
    li, lj = torch.tril_indices(n, n, offset=-1)
    got = oracle.tril_fwd(z, out_ld=P + 3, col_off=2)
    assert np.array_equal(got[:, 2:2 + P], z[:, li.numpy(), lj.numpy()])
    assert (got[:, :2] == 777).all() and (got[:, 2 + P:] == 777).all()
    g = rng.standard_normal((B, P)).astype(np.float32)
    base = rng.standard_normal((B, n, n)).astype(np.float32)
    zt = torch.from_numpy(z).requires_grad_(True)
    (zt[:, li, lj] * torch.from_numpy(g)).sum().backward()
    assert np.array_equal(oracle.tril_bwd(g, base), base + zt.grad.numpy())


@pytest.mark.parametrize("B,c,d", [(3, 2, 1), (5, 4, 8), (4, 27, 128), (2, 32, 36), (3, 9, 130)])
def test_dot_interaction_matches_torch(oracle, B, c, d):
    """The fused pairwise-dot interaction against the torch composition the reference's op tests use
    (cat -> reshape -> bmm with the transpose, tests/ops/test_harness.py:125-177) + MLPerf's tril pick; 1e-5."""
    import torch
    rng = np.random.default_rng(B * 100 + c)
    z = rng.uniform(-1, 1, (B, c, d)).astype(np.float32)
    zt = torch.from_numpy(z).requires_grad_(True)
    p = torch.bmm(zt, zt.transpose(1, 2))
    li, lj = torch.tril_indices(c, c, offset=-1)
    exp = torch.cat([zt[:, 0, :], p[:, li, lj]], dim=1)
    got = oracle.dot_interaction_fwd(z)
    np.testing.assert_allclose(got, exp.detach().numpy(), rtol=1e-5, atol=1e-5)
    g = rng.uniform(-1, 1, got.shape).astype(np.float32)
    (exp * torch.from_numpy(g)).sum().backward()
    np.testing.assert_allclose(oracle.dot_interaction_bwd(z, g), zt.grad.numpy(), rtol=1e-5, atol=1e-5)
    base = rng.uniform(-1, 1, z.shape).astype(np.float32)
    np.testing.assert_allclose(oracle.dot_interaction_bwd(z, g, base), base + zt.grad.numpy(), rtol=1e-5, atol=1e-5)
