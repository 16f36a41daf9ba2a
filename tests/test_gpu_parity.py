"""GPU parity tests (-m gpu): the hand-written HIP kernels, called through the C-ABI
(dlrm_flexflow_amd/csrc/libffhip.so), against
  * the committed golden vectors (reference's compiled AVX2 lookup / torch / numpy),
  * the CPU oracle on the same seeded inputs, and
  * size-independent properties at BASELINE.json's full sizes.
Bar: bit-exact for index / gather / copy / canonical-order work; 1e-5 relative for fp32
reductions whose order differs (GEMMs, atomics), tolerance written at each check.
torch is used only to own device memory and, in the full-size property tests, as an
independent checker of pure gathers.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import golden
from dlrm_flexflow_amd import capi

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def sync():
    torch.cuda.synchronize()


@pytest.fixture(scope="module")
def ws(hip):
    """128 MiB of scratch attached to the ctx (FFHandler.workSpace analogue)."""
    buf = torch.empty(128 << 20, dtype=torch.uint8, device=DEV)
    hip.set_workspace(buf, buf.numel())
    return buf


def bits_equal(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


# ---------------------------------------------------------------------------
# RNG / init kernels: bit-exact with the oracle
# ---------------------------------------------------------------------------
def test_rng_kernels_bit_exact(hip, oracle):
    n = 100003
    t = torch.empty(n, dtype=torch.float32, device=DEV)
    hip.call("ffh_init_uniform", t, n, 7, -0.125, 0.25, None)
    assert bits_equal(host(t), oracle.init_uniform(n, 7, -0.125, 0.25))
    hip.call("ffh_gen_uniform01", t, n, 9, 555, None)
    assert bits_equal(host(t), oracle.gen_uniform01(n, 9, 555))
    hip.call("ffh_gen_bernoulli", t, n, 9, 555, None)
    assert bits_equal(host(t), oracle.gen_bernoulli(n, 9, 555))
    i = torch.empty(n, dtype=torch.int64, device=DEV)
    for R in (3, 10131227, 200_000_000):
        hip.call("ffh_gen_indices", i, n, 11, 1 << 33, R, None)
        assert bits_equal(host(i), oracle.gen_indices(n, 11, 1 << 33, R))
    hip.call("ffh_fill_f32", t, n, 1.5, None)
    assert (host(t) == 1.5).all()
    hip.call("ffh_fill_f32", t[1:], n - 1, -2.0, None)       # unaligned base
    h = host(t)
    assert h[0] == 1.5 and (h[1:] == -2.0).all()


def test_embedding_localize_rows_bit_exact(hip, oracle):
    """Row-wise sharded table: ids relative to the local block, everything else -> the zero row; in place too."""
    rng = np.random.default_rng(8)
    for n, R, r0, nloc in ((1, 10, 0, 10), (100003, 200_000_000, 75_000_000, 25_000_000), (4096, 50, 12, 13), (777, 9, 9, 0)):
        idx = rng.integers(0, R, n)
        d = dev(idx)
        out = torch.empty_like(d)
        hip.call("ffh_embedding_localize_rows", d, out, n, r0, nloc, None)
        assert np.array_equal(host(out), oracle.embedding_localize_rows(idx, r0, nloc))
        hip.call("ffh_embedding_localize_rows", d, d, n, r0, nloc, None)
        assert np.array_equal(host(d), host(out))
    with pytest.raises(capi.FFHError):
        hip.call("ffh_embedding_localize_rows", None, None, 5, 0, 1, None)


# ---------------------------------------------------------------------------
# Embedding forward
# ---------------------------------------------------------------------------
def gpu_emb_fwd(hip, idx, w, aggr=capi.AGGR_MODE_SUM, out_ld=None, col_off=0):
    B, L = idx.shape
    R, D = w.shape
    out_ld = out_ld or D
    out = torch.full((B, out_ld), 777.0, dtype=torch.float32, device=DEV)
    hip.call("ffh_embedding_fwd", dev(idx), out[:, col_off:], dev(w), L, D, B, R, out_ld, aggr, None)
    return host(out)


def test_embedding_fwd_reference_golden_bit_exact(hip):
    """Golden vectors produced by the reference's own compiled AVX2 lookup."""
    g = golden("embedding_fwd_ref")
    for k in range(int(g["n_cases"])):
        out = gpu_emb_fwd(hip, g[f"c{k}_idx"], g[f"c{k}_w"])
        assert bits_equal(out, g[f"c{k}_out"]), f"case {k}"
    # ragged bags of the fixture, evaluated bag by bag (fixed bag size in this ABI), SUM and mean
    w, flat, lens = g["ragged_w"], g["ragged_idx"], g["ragged_len"]
    pos = 0
    for b, ln in enumerate(lens):
        if ln:
            idx = flat[pos:pos + ln].reshape(1, ln)
            assert bits_equal(gpu_emb_fwd(hip, idx, w), g["ragged_out"][b:b + 1])
            assert bits_equal(gpu_emb_fwd(hip, idx, w, capi.AGGR_MODE_AVG), g["ragged_out_mean"][b:b + 1])
            pos += ln


@pytest.mark.parametrize("B,L,D,R", [(4096, 1, 128, 100000), (2048, 1, 16, 10131227 // 50), (1000, 3, 64, 5000),
                                     (777, 2, 13, 300), (513, 1, 512, 2000), (64, 4, 20, 9), (1, 1, 128, 5), (100, 1, 48, 77)])
def test_embedding_fwd_vs_oracle_bit_exact(hip, oracle, B, L, D, R):
    rng = np.random.default_rng(B + D)
    w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
    idx = rng.integers(0, R, (B, L))
    assert bits_equal(gpu_emb_fwd(hip, idx, w), oracle.embedding_fwd(idx, w))
    if L > 1:
        assert bits_equal(gpu_emb_fwd(hip, idx, w, capi.AGGR_MODE_AVG), oracle.embedding_fwd(idx, w, capi.AGGR_MODE_AVG))


def test_embedding_fwd_empty_batch_and_bad_args(hip):
    w = torch.zeros(4, 8, device=DEV)
    idx = torch.zeros(0, 1, dtype=torch.int64, device=DEV)
    out = torch.zeros(1, 8, device=DEV)
    hip.call("ffh_embedding_fwd", idx, out, w, 1, 8, 0, 4, 8, capi.AGGR_MODE_SUM, None)     # empty: no-op
    with pytest.raises(capi.FFHError):
        hip.call("ffh_embedding_fwd", idx, out, w, 1, 8, 1, 4, 4, capi.AGGR_MODE_SUM, None)  # ld < D
    with pytest.raises(capi.FFHError):
        hip.call("ffh_embedding_fwd", idx, out, w, 1, 8, 1, 4, 8, capi.AGGR_MODE_NONE, None)


def test_embedding_fwd_multi_table_into_concat_buffer(hip, oracle):
    """26 Criteo-Kaggle-shaped tables in one launch, writing straight into the [B][16+26*16]
    concat buffer (out_ld = 432): gathers bit-exact, other columns untouched."""
    rows = [1460, 583, 10131227 // 100, 2202608 // 100, 305, 24, 12517, 633, 3, 93145, 5683, 8351593 // 100, 3194, 27,
            14992, 5461306 // 100, 10, 5652, 2173, 4, 7046547 // 100, 18, 15, 286181, 105, 142572]
    B, D = 2048, 16
    rng = np.random.default_rng(0)
    ws_, idxs = [], []
    Z = torch.full((B, 16 + 26 * D), -5.0, dtype=torch.float32, device=DEV)
    entries = []
    for t, R in enumerate(rows):
        w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
        idx = rng.integers(0, R, (B, 1))
        ws_.append(w); idxs.append(idx)
        entries.append((dev(idx), dev(w), Z[:, 16 + t * D:], R, Z.shape[1]))
    keep = entries
    arr = hip.emb_tables(keep)
    hip.check(hip.lib.ffh_embedding_fwd_multi(hip.ctx, arr, len(rows), 1, D, B, capi.AGGR_MODE_SUM, None), "fwd_multi")
    z = host(Z)
    assert (z[:, :16] == -5.0).all()
    for t in range(26):
        assert bits_equal(z[:, 16 + t * D:16 + (t + 1) * D].copy(), oracle.embedding_fwd(idxs[t], ws_[t]))


# ---------------------------------------------------------------------------
# Embedding backward: dense atomics (reference's own form) and the fused sparse SGD
# ---------------------------------------------------------------------------
def test_embedding_bwd_dense(hip, oracle):
    rng = np.random.default_rng(2)
    B, L, D, R = 3000, 2, 16, 50
    idx = rng.integers(0, R, (B, L))
    g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    wg = torch.zeros(R, D, device=DEV)
    hip.call("ffh_embedding_bwd_dense", dev(idx), dev(g), wg, L, D, B, R, D, capi.AGGR_MODE_SUM, None)
    exp = oracle.embedding_bwd_dense(idx, g, R)
    mass = np.zeros((R, D))
    np.add.at(mass, idx.reshape(-1), np.repeat(np.abs(g.astype(np.float64)), L, axis=0))
    # atomics arrive in any order: 1e-5 of the L1 mass of each sum (north_star tolerance)
    assert np.all(np.abs(host(wg) - exp) <= 1e-5 * mass)
    # unique rows: a single add each -> bit-exact
    idx_u = rng.permutation(4000)[:B].reshape(B, 1)
    wg = torch.zeros(4000, D, device=DEV)
    hip.call("ffh_embedding_bwd_dense", dev(idx_u), dev(g), wg, 1, D, B, 4000, D, capi.AGGR_MODE_SUM, None)
    assert np.array_equal(host(wg), oracle.embedding_bwd_dense(idx_u, g, 4000))


def gpu_fused(hip, idx, g, w, lr, aggr=capi.AGGR_MODE_SUM, gld=None):
    B, L = idx.shape
    R, D = w.shape
    gt = dev(g)
    wt = dev(w)
    hip.call("ffh_embedding_bwd_sgd_fused", dev(idx), gt, wt, L, D, B, R, gld or g.shape[1], aggr, float(lr), None)
    return host(wt)


@pytest.mark.parametrize("B,L,D,R", [
    (32768, 1, 16, 3), (32768, 1, 128, 36), (8192, 1, 16, 1460), (4096, 1, 128, 100000), (2048, 1, 16, 10131227),
    (3000, 2, 64, 500), (1000, 3, 13, 40), (5000, 1, 20, 7), (1, 1, 16, 10), (129, 1, 8, 1), (2047, 1, 256, 2000),
    (4096, 1, 512, 50),
    # batch * bag <= 2048: the single-launch small-batch kernel (all radix passes LDS-resident, then the same reduce / fold bodies)
    (1024, 2, 16, 3), (1025, 1, 16, 5), (600, 3, 20, 17), (2048, 1, 64, 1), (512, 4, 32, 1 << 20), (33, 1, 16, 100000), (2048, 1, 16, 513)])
def test_embedding_fused_bwd_sgd_bit_exact_vs_oracle(hip, oracle, ws, B, L, D, R):
    """Canonical-order reduction: the GPU result equals the oracle bit for bit, for heavy
    duplicates (R=3), rare duplicates, bags, odd D (scalar path) and rows untouched."""
    rng = np.random.default_rng(R + B)
    idx = rng.integers(0, R, (B, L))
    g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    if R <= 200000:
        w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
        got = gpu_fused(hip, idx, g, w, 0.01)
        assert bits_equal(got, oracle.embedding_bwd_sgd_fused(idx, g, w, 0.01))
        if L > 1:
            got = gpu_fused(hip, idx, g, w, 0.01, capi.AGGR_MODE_AVG)
            assert bits_equal(got, oracle.embedding_bwd_sgd_fused(idx, g, w, 0.01, capi.AGGR_MODE_AVG))
    else:
        # big table: compare only the touched rows (the oracle copy of the table would be 650 MB)
        wt = torch.empty(R, D, device=DEV)
        hip.call("ffh_init_uniform", wt, R * D, 3, -0.1, 0.1, None)
        before = wt.clone()
        hip.call("ffh_embedding_bwd_sgd_fused", dev(idx), dev(g), wt, L, D, B, R, D, capi.AGGR_MODE_SUM, 0.01, None)
        rows = np.unique(idx)
        sub = host(before[torch.from_numpy(rows).to(DEV)])
        remap = np.searchsorted(rows, idx)
        exp = oracle.embedding_bwd_sgd_fused(remap, g, sub, 0.01)
        assert bits_equal(host(wt[torch.from_numpy(rows).to(DEV)]), exp)
        mask = torch.ones(R, dtype=torch.bool, device=DEV)
        mask[torch.from_numpy(rows).to(DEV)] = False
        assert torch.equal(wt[mask], before[mask])           # untouched rows keep their bits


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_embedding_fused_random_shapes_bit_exact(hip, oracle, ws, seed):
    """Random (batch, bag, width, rows): the small-batch single launch and the tiled multi-launch form, vector and scalar
    row paths, SUM and AVG, 1..4 radix passes -- always the oracle's bits; and the gather on the same draws."""
    rng = np.random.default_rng(seed)
    for case in range(10):
        B = int(rng.choice([1, 7, 33, 500, 1024, 2048, 2049, 3000, 5000, 9000]))
        L = int(rng.choice([1, 1, 2, 3]))
        D = int(rng.choice([4, 8, 13, 16, 20, 32, 64, 128]))
        R = int(rng.choice([1, 2, 17, 300, 513, 5000, 70000, 300000]))
        if R * D > 2e7:
            R = int(2e7 // D)
        idx = rng.integers(0, R, (B, L))
        if rng.integers(0, 3) == 0:
            idx[:] = idx[0, 0]                                    # one row takes everything
        g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
        w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
        aggr = capi.AGGR_MODE_AVG if (L > 1 and rng.integers(0, 2)) else capi.AGGR_MODE_SUM
        what = f"seed {seed} case {case}: B={B} L={L} D={D} R={R} aggr={aggr}"
        assert bits_equal(gpu_fused(hip, idx, g, w, 0.03, aggr), oracle.embedding_bwd_sgd_fused(idx, g, w, 0.03, aggr)), what
        assert bits_equal(gpu_emb_fwd(hip, idx, w, aggr), oracle.embedding_fwd(idx, w, aggr)), what


@pytest.mark.parametrize("B", [4096, 2048, 300])
def test_embedding_fused_multi_table_strided_grad(hip, oracle, ws, B):
    """Several tables in one call, gradients read as column slices of one [B][ld] buffer
    (the concat gradient), exactly how the FFModel shim calls it.  B <= 2048: the small-batch kernel with a different
    number of radix passes per table."""
    rng = np.random.default_rng(5)
    D = 16
    rows = [3, 100, 5000, 250000, 17]
    ld = 16 + len(rows) * D
    G = rng.uniform(-1, 1, (B, ld)).astype(np.float32)
    Gt = dev(G)
    ents, wt, ws_, idxs = [], [], [], []
    for t, R in enumerate(rows):
        w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
        idx = rng.integers(0, R, (B, 1))
        ws_.append(w); idxs.append(idx)
        wt.append(dev(w))
        ents.append((dev(idx), wt[-1], Gt[:, 16 + t * D:], R, ld))
    arr = hip.emb_tables(ents)
    hip.check(hip.lib.ffh_embedding_bwd_sgd_fused_multi(hip.ctx, arr, len(rows), 1, D, B, capi.AGGR_MODE_SUM, 0.05, None), "fused_multi")
    for t, R in enumerate(rows):
        gs = np.ascontiguousarray(G[:, 16 + t * D:16 + (t + 1) * D])
        assert bits_equal(host(wt[t]), oracle.embedding_bwd_sgd_fused(idxs[t], gs, ws_[t], 0.05)), f"table {t}"


def test_embedding_backward_reference_fixture_on_gpu(hip, oracle, ws):
    """The reference's compiled CPU embed_backward (tests/golden/embedding_bwd_ref.npz) against the HIP kernels:
    ffh_embedding_bwd_dense within 1e-5 of it (atomics: no fixed order), the fused update equal to
    W - lr * (that gradient) within 1e-6 and bit-equal to the oracle's fused update."""
    g = golden("embedding_bwd_ref")
    for k in range(int(g["n_cases"])):
        idx, gr, wg0, wg = g[f"c{k}_idx"], g[f"c{k}_g"], g[f"c{k}_wg0"], g[f"c{k}_wg"]
        R, D = wg0.shape
        B = idx.shape[0]
        wgt = dev(wg0)
        hip.call("ffh_embedding_bwd_dense", dev(idx), dev(gr), wgt, 1, D, B, R, D, capi.AGGR_MODE_SUM, None)
        mass = np.abs(wg0).astype(np.float64)
        np.add.at(mass, idx.reshape(-1), np.abs(gr.astype(np.float64)))
        assert_gemm_close(host(wgt), wg, mass, f"dense case {k}")
        w = np.random.default_rng(k).uniform(-1, 1, (R, D)).astype(np.float32)
        fused = gpu_fused(hip, idx, gr, w, 0.05)
        assert bits_equal(fused, oracle.embedding_bwd_sgd_fused(idx, gr, w, 0.05))
        ref_grad = wg.astype(np.float64) - wg0.astype(np.float64)
        np.testing.assert_allclose(fused, w - 0.05 * ref_grad, rtol=1e-5, atol=2e-6 * max(1.0, np.abs(ref_grad).max()), err_msg=f"fused case {k}")


def test_embedding_fused_equals_reference_three_step_path(hip, oracle, ws):
    """zero_grad -> embed_backward (atomics) -> sgd_update, all three on the GPU, against the
    fused kernel: same table within 1e-5 (the dense path's atomics have no fixed order)."""
    rng = np.random.default_rng(9)
    B, D, R, lr = 8192, 32, 1000, 0.01
    idx = rng.integers(0, R, (B, 1))
    g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
    fused = gpu_fused(hip, idx, g, w, lr)
    wt, wg = dev(w), torch.empty(R, D, device=DEV)
    hip.call("ffh_zero", wg, R * D * 4, None)
    hip.call("ffh_embedding_bwd_dense", dev(idx), dev(g), wg, 1, D, B, R, D, capi.AGGR_MODE_SUM, None)
    hip.call("ffh_sgd_update", wt, wg, None, R * D, lr, 0.0, 0.0, 0, None)
    mass = np.zeros((R, D))
    np.add.at(mass, idx.reshape(-1), np.abs(g.astype(np.float64)))
    assert np.all(np.abs(host(wt) - fused) <= 1e-5 * (lr * mass + np.abs(w)))


def test_embedding_fused_needs_workspace(hip, ws):
    hip.set_workspace(None, 0)
    w = torch.zeros(4, 8, device=DEV)
    idx = torch.zeros(16, 1, dtype=torch.int64, device=DEV)
    g = torch.zeros(16, 8, device=DEV)
    try:
        with pytest.raises(capi.FFHError, match="workspace"):
            hip.call("ffh_embedding_bwd_sgd_fused", idx, g, w, 1, 8, 16, 4, 8, capi.AGGR_MODE_SUM, 0.1, None)
    finally:
        hip.set_workspace(ws, ws.numel())


# ---------------------------------------------------------------------------
# Linear (exact-fp32 MFMA GEMM)
# ---------------------------------------------------------------------------
def gpu_linear_fwd(hip, x, w, b, act, ldx=None, ldy=None):
    B, IN = x.shape
    OUT = w.shape[0]
    ldx, ldy = ldx or IN, ldy or OUT
    xt = torch.zeros(B, ldx, device=DEV); xt[:, :IN] = dev(x)
    yt = torch.full((B, ldy), 3.0, device=DEV)
    hip.call("ffh_linear_fwd", xt, ldx, yt, ldy, dev(w), None if b is None else dev(b), IN, OUT, B, act, None)
    return host(yt)[:, :OUT].copy()


def gpu_linear_bwd(hip, x, y, gy, w, act, want_dx=True, use_bias=True):
    B, IN = x.shape
    OUT = w.shape[0]
    dx = torch.zeros(B, IN, device=DEV) if want_dx else None
    dw = torch.zeros(OUT, IN, device=DEV)
    db = torch.zeros(OUT, device=DEV) if use_bias else None
    dy = dev(gy)
    hip.call("ffh_linear_bwd", dev(x), IN, dx, IN, dev(y), OUT, dy, OUT, dev(w), dw, db, IN, OUT, B, act, None)
    return (host(dx) if want_dx else None), host(dw), (host(db) if use_bias else None), host(dy)


def assert_gemm_close(got, exp, absmass, what):
    """|got-exp| <= 1e-5 * sum_k |a_k b_k|  (1e-5 relative, north_star) + 1e-6 absolute"""
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    assert np.all(err <= 1e-5 * absmass + 1e-6), f"{what}: max err {err.max()} vs mass {absmass.max()}"


def test_linear_torch_golden(hip):
    g = golden("linear_torch")
    for k in range(int(g["n_cases"])):
        x, w, b, gy = g[f"c{k}_x"], g[f"c{k}_w"], g[f"c{k}_b"], g[f"c{k}_gy"]
        act = int(g[f"c{k}_act"])
        y = gpu_linear_fwd(hip, x, w, b, act)
        np.testing.assert_allclose(y, g[f"c{k}_y"], rtol=1e-5, atol=1e-5)
        dx, dw, db, _ = gpu_linear_bwd(hip, x, g[f"c{k}_y"], gy, w, act)
        np.testing.assert_allclose(dw, g[f"c{k}_dw"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(db, g[f"c{k}_db"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(dx, g[f"c{k}_dx"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,IN,OUT,act", [
    (2048, 13, 512, capi.AC_MODE_RELU), (2048, 512, 256, capi.AC_MODE_RELU), (2048, 256, 64, capi.AC_MODE_RELU),
    (2048, 64, 16, capi.AC_MODE_RELU), (2048, 432, 512, capi.AC_MODE_RELU), (2048, 256, 1, capi.AC_MODE_SIGMOID),
    (128, 144, 64, capi.AC_MODE_NONE), (4096, 1024, 1024, capi.AC_MODE_RELU), (333, 77, 45, capi.AC_MODE_SIGMOID),
    (10, 2000, 1000, capi.AC_MODE_NONE), (4096, 1024, 1, capi.AC_MODE_SIGMOID), (515, 300, 4, capi.AC_MODE_RELU),
    (64, 1500, 2, capi.AC_MODE_NONE), (300, 20, 100, capi.AC_MODE_RELU), (64, 32, 64, capi.AC_MODE_NONE), (4096, 13, 512, capi.AC_MODE_RELU),
    (37, 5, 16, capi.AC_MODE_SIGMOID), (1001, 16, 1024, capi.AC_MODE_NONE), (33, 1, 64, capi.AC_MODE_RELU), (2048, 13, 96, capi.AC_MODE_RELU)])
def test_linear_vs_oracle(hip, oracle, B, IN, OUT, act):
    """DLRM layer shapes of C1/C2/C4 plus ragged sizes and the reference harness shape
    (10,2000,1000) [ref: tests/ops/test_harness.py:201-283]."""
    rng = np.random.default_rng(IN * OUT)
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    y = gpu_linear_fwd(hip, x, w, b, act)
    y_exp = oracle.linear_fwd(x, w, b, act)
    mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
    assert_gemm_close(y, y_exp, mass, "y")
    dx, dw, db, dy_after = gpu_linear_bwd(hip, x, y_exp, gy, w, act)
    dx_e, dw_e, db_e, dy_e = oracle.linear_bwd(x, y_exp, gy, w, act)
    np.testing.assert_allclose(dy_after, dy_e, rtol=1e-6, atol=1e-7)       # elementwise activation gradient
    a = np.abs(dy_e).astype(np.float64)
    assert_gemm_close(dw, dw_e, a.T @ np.abs(x).astype(np.float64), "dw")
    assert_gemm_close(db, db_e, a.sum(0), "db")
    assert_gemm_close(dx, dx_e, a @ np.abs(w).astype(np.float64), "dx")


def test_linear_strided_operands_and_accumulate(hip, oracle):
    """x and y as column slices of wider buffers (the concat buffer), dx accumulated on top of
    existing values (beta = 1, the reference's semantics), no bias, dx = NULL."""
    rng = np.random.default_rng(3)
    B, IN, OUT = 300, 24, 40
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = rng.uniform(-1, 1, (OUT, IN)).astype(np.float32)
    y = gpu_linear_fwd(hip, x, w, None, capi.AC_MODE_NONE, ldx=IN + 8, ldy=OUT + 4)
    assert_gemm_close(y, oracle.linear_fwd(x, w, None), np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T, "y")
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    dx0 = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    dxt, dwt = dev(dx0), torch.zeros(OUT, IN, device=DEV)
    hip.call("ffh_linear_bwd", dev(x), IN, dxt, IN, dev(y), OUT, dev(gy), OUT, dev(w), dwt, None, IN, OUT, B, capi.AC_MODE_NONE, None)
    dx_e, dw_e, _, _ = oracle.linear_bwd(x, y, gy, w, capi.AC_MODE_NONE, use_bias=False)
    np.testing.assert_allclose(host(dxt), dx0 + dx_e, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(host(dwt), dw_e, rtol=1e-5, atol=1e-4)
    dwt.zero_()
    hip.call("ffh_linear_bwd", dev(x), IN, None, IN, dev(y), OUT, dev(gy), OUT, dev(w), dwt, None, IN, OUT, B, capi.AC_MODE_NONE, None)
    np.testing.assert_allclose(host(dwt), dw_e, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("OUT", [80, 1, 3, 12])
@pytest.mark.parametrize("act", [capi.AC_MODE_RELU, capi.AC_MODE_SIGMOID, capi.AC_MODE_NONE])
def test_linear_bwd_ex_forms_equal_reference_form(hip, oracle, act, OUT):
    """ffh_linear_bwd_ex: overwrite-mode dx, the forked weight-gradient stream, and the split
    ONLY_DX / ONLY_DW calls all give what one ffh_linear_bwd call gives (and what the oracle gives)."""
    rng = np.random.default_rng(act)
    B, IN = 777, 96            # OUT = 1, 3, 12: the one-launch skinny-output kernels (linear_skinny_*), OUT = 80: the GEMMs
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / 8).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    y = oracle.linear_fwd(x, w, b, act)
    dx_e, dw_e, db_e, dy_e = oracle.linear_bwd(x, y, gy, w, act)
    s2 = torch.cuda.Stream()

    def run(mode):
        dx = torch.full((B, IN), 5.0, device=DEV) if mode != "plain" else torch.zeros(B, IN, device=DEV)
        dw, db, dy = torch.zeros(OUT, IN, device=DEV), torch.zeros(OUT, device=DEV), dev(gy)
        args = (dev(x), IN, dx, IN, dev(y), OUT, dy, OUT, dev(w), dw, db, IN, OUT, B, act)
        if mode == "plain":
            hip.call("ffh_linear_bwd", *args, None)
        elif mode == "overwrite+fork":
            hip.call("ffh_linear_bwd_ex", *args, 1, None, s2.cuda_stream)
        else:   # split calls; sigmoid needs ONLY_DX first
            hip.call("ffh_linear_bwd_ex", *args, 1 | 4, None, None)
            hip.call("ffh_linear_bwd_ex", *args, 2, None, None)
        torch.cuda.synchronize()
        return host(dx), host(dw), host(db), host(dy)

    for mode in ("plain", "overwrite+fork", "split"):
        dx, dw, db, dy = run(mode)
        a = np.abs(dy_e).astype(np.float64)
        np.testing.assert_allclose(dy, dy_e, rtol=1e-6, atol=1e-7, err_msg=mode)
        assert_gemm_close(dx, dx_e, a @ np.abs(w).astype(np.float64), f"dx {mode}")
        assert_gemm_close(dw, dw_e, a.T @ np.abs(x).astype(np.float64), f"dw {mode}")
        assert_gemm_close(db, db_e, a.sum(0), f"db {mode}")


@pytest.mark.parametrize("B,IN,OUT", [(2048, 432, 512), (2048, 512, 256), (2050, 436, 260), (300, 1024, 512), (4096, 200, 256),
                                      (1000, 256, 640)])
def test_linear_lds_dma_kernel_shapes(hip, oracle, B, IN, OUT):
    """The LDS-DMA GEMM (linear.hip, gemm_glds_kernel) on its own: the two Kaggle shapes it exists for, ragged M / N
    (rows and columns served from the zero page), a k extent that is not a multiple of the 64-wide k-tile, the 32x64
    tile form (few tiles), split-K over the batch for dW with db from the LDS image, strided operands, and the
    producer-side relu' flags -- against the oracle's plain calls."""
    rng = np.random.default_rng(B + IN + OUT)
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    for act in (capi.AC_MODE_RELU, capi.AC_MODE_NONE):
        y = gpu_linear_fwd(hip, x, w, b, act, ldx=IN + 4, ldy=OUT + 8)
        mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
        assert_gemm_close(y, oracle.linear_fwd(x, w, b, act), mass, f"y act {act}")
    # layer taken as the LOWER one of a chain: dy arrives premasked; and as the UPPER one: dx masked by x > 0
    y = oracle.linear_fwd(x, w, b, capi.AC_MODE_RELU)
    xr = np.maximum(x, 0)                       # pretend x is itself a ReLU output
    yr = oracle.linear_fwd(xr, w, b, capi.AC_MODE_RELU)
    gm = np.where(yr > 0, gy, 0).astype(np.float32)
    flags = capi.LINEAR_DY_PREMASKED | capi.LINEAR_DX_MASK_BY_X
    dx_e, dw_e, db_e, _ = oracle.linear_bwd_ex(xr, yr, gm, w, capi.AC_MODE_RELU, flags | capi.LINEAR_DX_OVERWRITE)
    a = np.abs(gm).astype(np.float64)
    # overwrite / accumulate / fork: both GEMMs in one launch (gemm_glds_bwd_kernel); split: the two single-GEMM kernels
    for mode in ("overwrite", "accumulate", "fork", "split"):
        dx0 = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
        dx, dw, db, dy = dev(dx0), torch.zeros(OUT, IN, device=DEV), torch.zeros(OUT, device=DEV), dev(gm)
        f = flags | (0 if mode == "accumulate" else capi.LINEAR_DX_OVERWRITE)
        s2 = torch.cuda.Stream()
        args = (dev(xr), IN, dx, IN, dev(yr), OUT, dy, OUT, dev(w), dw, db, IN, OUT, B, capi.AC_MODE_RELU)
        if mode == "split":
            hip.call("ffh_linear_bwd_ex", *args, f | capi.LINEAR_ONLY_DX, None, None)
            hip.call("ffh_linear_bwd_ex", *args, f | capi.LINEAR_ONLY_DW, None, None)
        else:
            hip.call("ffh_linear_bwd_ex", *args, f, None, s2.cuda_stream if mode == "fork" else None)
        torch.cuda.synchronize()
        assert bits_equal(host(dy), gm)                                       # premasked dy is not touched
        exp_dx = dx_e + (dx0 if mode == "accumulate" else 0)
        assert_gemm_close(host(dx), exp_dx, a @ np.abs(w).astype(np.float64) + (np.abs(dx0) if mode == "accumulate" else 0), f"dx {mode}")
        assert (host(dx)[xr <= 0] == (dx0[xr <= 0] if mode == "accumulate" else 0)).all()    # the mask is exact
        assert_gemm_close(host(dw), dw_e, a.T @ np.abs(xr).astype(np.float64), f"dw {mode}")
        assert_gemm_close(host(db), db_e, a.sum(0), f"db {mode}")
    # and the whole chain semantics on the GPU: plain two-call form == flagged form (1e-5)
    dxp, dwp, dbp, _ = gpu_linear_bwd(hip, xr, yr, gy, w, capi.AC_MODE_RELU)
    assert_gemm_close(dwp, dw_e, a.T @ np.abs(xr).astype(np.float64), "dw plain vs flagged")
    assert_gemm_close(dbp, db_e, a.sum(0), "db plain vs flagged")


def test_linear_unsupported_activation_is_an_error(hip):
    t = torch.zeros(4, 4, device=DEV)
    with pytest.raises(capi.FFHError):
        hip.call("ffh_linear_fwd", t, 4, t, 4, t, None, 4, 4, 4, capi.AC_MODE_TANH, None)
    # GELU: forward only, as in the reference [ref: src/ops/linear.cu:454-459 forward, :632-635 backward asserts]
    with pytest.raises(capi.FFHError):
        hip.call("ffh_linear_bwd", t, 4, t, 4, t, 4, t, 4, t, t, t, 4, 4, 4, capi.AC_MODE_GELU, None)


@pytest.mark.gpu
@pytest.mark.parametrize("B,IN,OUT", [(64, 32, 16), (300, 432, 512), (2048, 512, 256), (4096, 1024, 1024), (1000, 13, 512), (777, 100, 3)])
def test_linear_gelu_forward(hip, oracle, B, IN, OUT):
    """AC_MODE_GELU [ref: gelu_forward_kernel, src/runtime/cuda_helper.cu:81-90: x * (0.5 + 0.5 tanh(x (C x^2 + B)))] through every
    forward kernel family (thin input, skinny output, LDS-DMA, register-staged); oracle and torch's tanh-form gelu as checkers."""
    rng = np.random.default_rng(IN + OUT)
    x = rng.uniform(-2, 2, (B, IN)).astype(np.float32)
    w = (rng.uniform(-2, 2, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    y = gpu_linear_fwd(hip, x, w, b, capi.AC_MODE_GELU)
    y_exp = oracle.linear_fwd(x, w, b, capi.AC_MODE_GELU)
    mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
    assert_gemm_close(y, y_exp, mass, "y")                      # |gelu'| <= 1.13: the pre-activation bound carries over
    t = torch.nn.functional.gelu(torch.from_numpy(x) @ torch.from_numpy(w).T + torch.from_numpy(b), approximate="tanh").numpy()
    np.testing.assert_allclose(y, t, rtol=1e-4, atol=2e-5)


# ---------------------------------------------------------------------------
# Concat / BatchMatmul / loss / metrics / SGD
# ---------------------------------------------------------------------------
def test_concat_golden_and_alias(hip):
    g = golden("concat_numpy")
    for k in range(int(g["n_cases"])):
        parts = [g[f"c{k}_in{i}"] for i in range(int(g[f"c{k}_n"]))]
        widths = [p.shape[1] for p in parts]
        nb = parts[0].shape[0]
        pt = [dev(p) for p in parts]
        out = torch.empty(nb, sum(widths), device=DEV)
        hip.concat("ffh_concat_fwd", out, sum(widths), pt, widths, None, nb)
        assert bits_equal(host(out), g[f"c{k}_out"])               # pure copy: bit-exact
        grads = [torch.zeros(nb, w_, device=DEV) for w_ in widths]
        hip.concat("ffh_concat_bwd", out, sum(widths), grads, widths, None, nb)
        for p, q in zip(parts, grads):
            assert np.array_equal(p, host(q))
        # ffh_concat_bwd_ex: accumulate on top of existing values (the reference's add_with_stride) / overwrite them
        for flags in (0, capi.CONCAT_BWD_OVERWRITE):
            grads = [torch.full((nb, w_), 2.0, device=DEV) for w_ in widths]
            n = len(parts)
            pa = (C.c_void_p * n)(*[q.data_ptr() for q in grads])
            ba = (C.c_int64 * n)(*widths)
            hip.check(hip.lib.ffh_concat_bwd_ex(hip.ctx, out.data_ptr(), sum(widths), pa, ba, None, n, nb, flags, None), "concat_bwd_ex")
            for p, q in zip(parts, grads):
                assert np.array_equal(p if flags else p + 2.0, host(q))
    # aliased producers: inputs that already live in the output are skipped, others copied
    nb, widths = 64, [16, 16, 16]
    Z = torch.zeros(nb, 48, device=DEV)
    Z[:, 16:32] = 7.0
    a, c = torch.ones(nb, 16, device=DEV), torch.full((nb, 16), 2.0, device=DEV)
    hip.concat("ffh_concat_fwd", Z, 48, [a, Z[:, 16:], c], widths, [16, 48, 16], nb)
    z = host(Z)
    assert (z[:, :16] == 1).all() and (z[:, 16:32] == 7).all() and (z[:, 32:] == 2).all()


def test_concat_maximum_fan_in_and_ragged_widths(hip, oracle):
    """256 inputs (MAX_NUM_INPUTS, [ref: include/config.h:30-37]) of ragged widths 1..7 in one launch each way, bit-exact;
    an empty batch is a no-op; 257 inputs are refused."""
    rng = np.random.default_rng(4)
    nb = 37
    widths = [int(w_) for w_ in rng.integers(1, 8, 256)]
    parts = [rng.standard_normal((nb, w_)).astype(np.float32) for w_ in widths]
    out = torch.empty(nb, sum(widths), device=DEV)
    hip.concat("ffh_concat_fwd", out, sum(widths), [dev(p) for p in parts], widths, None, nb)
    assert bits_equal(host(out), np.concatenate(parts, 1)) and bits_equal(host(out), oracle.concat_fwd(parts))
    grads = [torch.zeros(nb, w_, device=DEV) for w_ in widths]
    hip.concat("ffh_concat_bwd", out, sum(widths), grads, widths, None, nb)
    for p, q in zip(parts, grads):
        assert np.array_equal(p, host(q))
    hip.concat("ffh_concat_fwd", out, sum(widths), [dev(p) for p in parts], widths, None, 0)      # empty batch
    with pytest.raises(capi.FFHError):
        hip.concat("ffh_concat_fwd", out, sum(widths) + 1, [dev(p) for p in parts] + [dev(parts[0])], widths + [1], None, nb)


def test_bmm_golden_and_harness_shape(hip, oracle):
    g = golden("bmm_torch")
    for k in range(int(g["n_cases"])):
        a, b, go = g[f"c{k}_a"], g[f"c{k}_b"], g[f"c{k}_go"]
        d, n, kk = a.shape
        m = b.shape[2]
        o = torch.empty(d, n, m, device=DEV)
        hip.call("ffh_bmm_fwd", o, dev(a), dev(b), m, n, kk, d, -1, -1, -1, None)
        np.testing.assert_allclose(host(o), g[f"c{k}_o"], rtol=1e-5, atol=1e-5)     # harness: 1e-5
        ga, gb = torch.zeros(d, n, kk, device=DEV), torch.zeros(d, kk, m, device=DEV)
        hip.call("ffh_bmm_bwd", dev(go), dev(a), ga, dev(b), gb, m, n, kk, d, None)
        np.testing.assert_allclose(host(ga), g[f"c{k}_ga"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(host(gb), g[f"c{k}_gb"], rtol=1e-5, atol=1e-5)
    np.random.seed(0)
    d, m, n, kk = 145, 265, 15, 64                                                  # harness: 1e-4 here
    a = np.random.uniform(0, 1, (d, n, kk)).astype(np.float32)
    b = np.random.uniform(0, 1, (d, kk, m)).astype(np.float32)
    o = torch.empty(d, n, m, device=DEV)
    hip.call("ffh_bmm_fwd", o, dev(a), dev(b), m, n, kk, d, -1, -1, -1, None)
    np.testing.assert_allclose(host(o), np.matmul(a, b), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(o), oracle.bmm_fwd(a, b), rtol=1e-5, atol=1e-4)
    # seq_length truncation keeps full-size strides
    o.zero_()
    hip.call("ffh_bmm_fwd", o, dev(a), dev(b), m, n, kk, d, 0, 1, 40, None)
    np.testing.assert_allclose(host(o), oracle.bmm_fwd(a, b, 0, 1, 40), rtol=1e-5, atol=1e-4)
    # DLRM dot-interaction shape: Z [B][27][128] x Z^T
    rng = np.random.default_rng(0)
    z = rng.uniform(-1, 1, (64, 27, 128)).astype(np.float32)
    zt = np.ascontiguousarray(z.transpose(0, 2, 1))
    o = torch.empty(64, 27, 27, device=DEV)
    hip.call("ffh_bmm_fwd", o, dev(z), dev(zt), 27, 27, 128, 64, -1, -1, -1, None)
    np.testing.assert_allclose(host(o), oracle.bmm_fwd(z, zt), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dims,perm", [((64, 27, 128), (0, 2, 1)), ((5, 33, 70), (0, 2, 1)), ((3, 4, 5, 6), (0, 3, 1, 2)),
                                       ((17, 9), (1, 0)), ((2, 3, 4), (2, 0, 1)), ((2048, 27, 16), (0, 2, 1))])
def test_transpose_bit_exact(hip, dims, perm):
    """Transpose forward is a pure permutation (bit-exact vs numpy); backward accumulates the inverse permutation."""
    rng = np.random.default_rng(len(dims))
    a = rng.uniform(-1, 1, dims).astype(np.float32)
    out = torch.empty([dims[p] for p in perm], device=DEV)
    hip.transpose("ffh_transpose_fwd", out, dev(a), dims, perm)
    assert bits_equal(host(out), np.ascontiguousarray(np.transpose(a, perm)))
    g = rng.uniform(-1, 1, out.shape).astype(np.float32)
    ig0 = rng.uniform(-1, 1, dims).astype(np.float32)
    ig = dev(ig0)
    hip.transpose("ffh_transpose_bwd", ig, dev(g), dims, perm)
    np.testing.assert_array_equal(host(ig), ig0 + np.transpose(g, np.argsort(perm)))


@pytest.mark.parametrize("B,n", [(1, 2), (33, 5), (2048, 27), (257, 64)])
def test_tril_bit_exact(hip, oracle, B, n):
    """Strict lower triangle of [B][n][n] (MLPerf-DLRM's pick of the pairwise dots), into a column slice of a wider
    buffer; backward adds into the kept entries only.  Copies and single adds: bit-exact with the oracle."""
    rng = np.random.default_rng(B + n)
    P = n * (n - 1) // 2
    z = rng.standard_normal((B, n, n)).astype(np.float32)
    out = torch.full((B, P + 5), 777.0, dtype=torch.float32, device=DEV)
    hip.call("ffh_tril_fwd", out[:, 3:], P + 5, dev(z), B, n, None)
    assert bits_equal(host(out), oracle.tril_fwd(z, out_ld=P + 5, col_off=3))
    g = rng.standard_normal((B, P + 5)).astype(np.float32)
    base = rng.standard_normal((B, n, n)).astype(np.float32)
    gd, bd = dev(g), dev(base)
    hip.call("ffh_tril_bwd", bd, gd[:, 3:], P + 5, B, n, None)
    assert bits_equal(host(bd), oracle.tril_bwd(np.ascontiguousarray(g[:, 3:3 + P]), base))
    with pytest.raises(capi.FFHError):
        hip.call("ffh_tril_fwd", out, P - 1, dev(z), B, n, None)      # out_ld too small
    with pytest.raises(capi.FFHError):
        hip.call("ffh_tril_fwd", out, P + 5, dev(z), B, 65, None)     # n > 64


@pytest.mark.parametrize("B,c,d", [(1, 2, 1), (7, 4, 8), (2048, 27, 128), (130, 32, 36), (33, 9, 130), (64, 27, 16),
                                   (5, 2, 128), (131, 32, 128), (9, 31, 128), (1030, 17, 128)])   # d = 128: the straight-line backward (round 4), smallest / largest c, odd batches
def test_dot_interaction_vs_oracle(hip, oracle, B, c, d):
    """The fused pairwise-dot interaction (one wave per sample, Z Z^T on fp32 MFMA) against the oracle's plain loops:
    pass-through columns bit-exact, dot products within 1e-5 (the MFMA adds the k terms in another order); strided
    sample rows and output rows (a concat buffer), accumulate and overwrite backward."""
    rng = np.random.default_rng(B + c + d)
    P = c * (c - 1) // 2
    ldz, ldo = c * d + 4, d + P + 8
    z = rng.uniform(-1, 1, (B, ldz)).astype(np.float32)
    zz = np.ascontiguousarray(z[:, :c * d]).reshape(B, c, d)
    out = torch.full((B, ldo), 777.0, dtype=torch.float32, device=DEV)
    hip.call("ffh_dot_interaction_fwd", dev(z), ldz, out, ldo, B, c, d, None)
    got, exp = host(out), oracle.dot_interaction_fwd(zz)
    assert bits_equal(got[:, :d], exp[:, :d])
    np.testing.assert_allclose(got[:, d:d + P], exp[:, d:], rtol=1e-5, atol=1e-5)
    assert (got[:, d + P:] == 777).all()
    g = rng.uniform(-1, 1, (B, ldo)).astype(np.float32)
    gg = np.ascontiguousarray(g[:, :d + P])
    base = rng.uniform(-1, 1, (B, ldz)).astype(np.float32)
    for flags in (0, 1):
        zg = dev(base)
        hip.call("ffh_dot_interaction_bwd", dev(z), ldz, dev(g), ldo, zg, ldz, B, c, d, flags, None)
        h = host(zg)
        want = oracle.dot_interaction_bwd(zz, gg, None if flags else base[:, :c * d].reshape(B, c, d))
        np.testing.assert_allclose(h[:, :c * d].reshape(B, c, d), want, rtol=1e-5, atol=2e-5)
        assert bits_equal(h[:, c * d:], base[:, c * d:])
    with pytest.raises(capi.FFHError):
        hip.call("ffh_dot_interaction_fwd", dev(z), ldz, out, ldo, B, 33, d, None)


@pytest.mark.parametrize("seed", range(8))
def test_dot_interaction_random_shapes(hip, oracle, seed):
    """Random (batch, vectors, width, strides) for the fused interaction and the tril kernels against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    B, c, d = int(rng.integers(1, 300)), int(rng.integers(2, 33)), int(rng.integers(1, 200))
    P = c * (c - 1) // 2
    ldz, ldo = c * d + int(rng.integers(0, 3)) * 4, d + P + int(rng.integers(0, 9))
    z = rng.uniform(-1, 1, (B, ldz)).astype(np.float32)
    zz = np.ascontiguousarray(z[:, :c * d]).reshape(B, c, d)
    out = torch.full((B, ldo), 777.0, dtype=torch.float32, device=DEV)
    hip.call("ffh_dot_interaction_fwd", dev(z), ldz, out, ldo, B, c, d, None)
    got, exp = host(out), oracle.dot_interaction_fwd(zz)
    assert bits_equal(got[:, :d], exp[:, :d]) and (got[:, d + P:] == 777).all()
    np.testing.assert_allclose(got[:, d:d + P], exp[:, d:], rtol=1e-5, atol=1e-5 * max(1, d // 16))
    g = rng.uniform(-1, 1, (B, ldo)).astype(np.float32)
    zg = torch.zeros(B, ldz, device=DEV)
    hip.call("ffh_dot_interaction_bwd", dev(z), ldz, dev(g), ldo, zg, ldz, B, c, d, 1, None)
    np.testing.assert_allclose(host(zg)[:, :c * d].reshape(B, c, d), oracle.dot_interaction_bwd(zz, np.ascontiguousarray(g[:, :d + P])),
                               rtol=1e-5, atol=2e-5 * max(1, c // 8))
    n = c if c >= 2 else 2
    m = rng.standard_normal((B, n, n)).astype(np.float32)
    t = torch.full((B, n * (n - 1) // 2 + 2), 777.0, dtype=torch.float32, device=DEV)
    hip.call("ffh_tril_fwd", t[:, 1:], t.shape[1], dev(m), B, n, None)
    assert bits_equal(host(t), oracle.tril_fwd(m, out_ld=t.shape[1], col_off=1))


def test_adam_and_zero_grad(hip, oracle):
    """ffh_adam_update / ffh_sgd_update_ex: bit-exact with the oracle, within 1e-5 of torch.optim.Adam (fixture)."""
    g = golden("adam_torch")
    for k in range(int(g["n_cases"])):
        alpha, b1, b2, wd, eps = g[f"c{k}_hp"]
        st = oracle.AdamState(alpha, b1, b2, wd, eps)
        w = dev(g[f"c{k}_w0"]); m = torch.zeros_like(w); v = torch.zeros_like(w)
        w_o = g[f"c{k}_w0"].copy(); m_o = np.zeros_like(w_o); v_o = np.zeros_like(w_o)
        for step in range(5):
            st.next()
            gd = dev(g[f"c{k}_g"][step])
            hip.call("ffh_adam_update", w, gd, m, v, w.numel(), float(st.alpha_t), float(b1), float(b2), float(wd), float(eps),
                     capi.OPT_ZERO_GRAD if step % 2 else 0, None)
            assert (host(gd) == 0).all() if step % 2 else bits_equal(host(gd), g[f"c{k}_g"][step])
            w_o, m_o, v_o = oracle.adam_update(w_o, g[f"c{k}_g"][step], m_o, v_o, st)
        assert bits_equal(host(w), w_o) and bits_equal(host(m), m_o) and bits_equal(host(v), v_o)
        np.testing.assert_allclose(host(w), g[f"c{k}_w5"], rtol=1e-5, atol=1e-6)
    n = (1 << 20) + 4                                                 # vector path, large; and an odd count (scalar path)
    rng = np.random.default_rng(1)
    for cnt in (n, 1001):
        w0, gr = rng.uniform(-1, 1, cnt).astype(np.float32), rng.uniform(-1, 1, cnt).astype(np.float32)
        m0, v0 = rng.uniform(-0.1, 0.1, cnt).astype(np.float32), rng.uniform(0, 0.1, cnt).astype(np.float32)
        st = oracle.AdamState(0.01, 0.9, 0.999, 1e-3, 1e-8); st.next(); st.next()
        w, gd, m, v = dev(w0), dev(gr), dev(m0), dev(v0)
        hip.call("ffh_adam_update", w, gd, m, v, cnt, float(st.alpha_t), 0.9, 0.999, 1e-3, 1e-8, capi.OPT_ZERO_GRAD, None)
        w_o, m_o, v_o = oracle.adam_update(w0, gr, m0, v0, st)
        assert bits_equal(host(w), w_o) and bits_equal(host(m), m_o) and bits_equal(host(v), v_o) and not host(gd).any()
        w, gd = dev(w0), dev(gr)
        hip.call("ffh_sgd_update_ex", w, gd, None, cnt, 0.01, 0.0, 0.0, 0, capi.OPT_ZERO_GRAD, None)
        assert bits_equal(host(w), oracle.sgd_update(w0, gr, 0.01)) and not host(gd).any()
    with pytest.raises(capi.FFHError):
        hip.call("ffh_adam_update", w, gd, None, None, 4, 0.01, 0.9, 0.999, 0.0, 1e-8, 0, None)


def test_sgd_mse_metrics(hip, oracle):
    g = golden("sgd_mse_torch")
    for k in range(int(g["n_cases"])):
        lr, wd, mom, nest = g[f"c{k}_hp"]
        w = dev(g[f"c{k}_w0"])
        v = torch.zeros_like(w) if mom > 0 else None
        w_o = g[f"c{k}_w0"].copy()
        v_o = np.zeros_like(w_o) if mom > 0 else None
        for step in range(3):
            hip.call("ffh_sgd_update", w, dev(g[f"c{k}_g"][step]), v, w.numel(), float(lr), float(wd), float(mom), int(nest), None)
            w_o = oracle.sgd_update(w_o, g[f"c{k}_g"][step], lr, wd, mom, bool(nest), v_o)
        assert bits_equal(host(w), w_o)                               # same FMA sequence: bit-exact
        np.testing.assert_allclose(host(w), g[f"c{k}_w3"], rtol=1e-5, atol=1e-6)
    n = 1 << 20                                                       # vector path, large
    rng = np.random.default_rng(1)
    w0, gr = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-1, 1, n).astype(np.float32)
    w = dev(w0)
    hip.call("ffh_sgd_update", w, dev(gr), None, n, 0.01, 0.0, 0.0, 0, None)
    assert bits_equal(host(w), oracle.sgd_update(w0, gr, 0.01))
    p, y = dev(g["mse_p"]), dev(g["mse_y"])
    lg = torch.empty_like(p)
    hip.call("ffh_mse_bwd", lg, p, y, p.numel(), 1.0 / 37, None)
    assert bits_equal(host(lg), oracle.mse_bwd(g["mse_p"], g["mse_y"], 1.0 / 37))
    np.testing.assert_allclose(host(lg), g["mse_grad"], rtol=1e-6, atol=1e-8)
    # compute_metrics + loss backward in one launch == the two separate entry points
    perf2, lg2 = torch.zeros(8, dtype=torch.int32, device=DEV), torch.empty_like(p)
    hip.call("ffh_mse_bwd_metrics", lg2, p, y, perf2, 37, 1, 1.0 / 37, capi.METRIC_ACCURACY | capi.METRIC_MSE, None)
    assert bits_equal(host(lg2), host(lg))
    perf = torch.zeros(8, dtype=torch.int32, device=DEV)
    hip.call("ffh_metrics_update", p, y, perf, 37, 1, capi.METRIC_ACCURACY | capi.METRIC_MSE, None)
    assert host(perf2)[:2].tolist() == [74, 37] and abs(host(perf2)[4:5].view(np.float32)[0] - float(g["mse_sum"])) <= 1e-5 * float(g["mse_sum"])
    hp = host(perf)
    assert hp[0] == 74 and hp[1] == 37                                # train_all double count (1 class + accuracy)
    mse = hp[4:5].view(np.float32)[0]
    assert abs(mse - float(g["mse_sum"])) <= 1e-5 * float(g["mse_sum"])
    d, s = dev(w0), dev(gr)
    hip.call("ffh_add_scaled", d, s, n, 0.5, None)
    np.testing.assert_array_equal(host(d), (w0.astype(np.float64) + gr.astype(np.float64) * 0.5).astype(np.float32))


@pytest.mark.parametrize("B,IN,OUT,act", [(2048, 256, 1, capi.AC_MODE_SIGMOID), (100, 64, 3, capi.AC_MODE_NONE), (777, 1024, 4, capi.AC_MODE_RELU),
                                         (8192, 256, 1, capi.AC_MODE_SIGMOID)])
def test_linear_bwd_mse_equals_the_two_calls(hip, oracle, B, IN, OUT, act):
    """ffh_linear_bwd_mse (loss step + metrics folded into the last layer's one-launch backward) against
    ffh_mse_bwd_metrics followed by ffh_linear_bwd_ex: dy and dX bit-exact (same operations per element), dW / db / the
    metric sums within 1e-5 (atomics); the oracle's restatement (the two calls) agrees."""
    rng = np.random.default_rng(B + OUT)
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = rng.uniform(-1, 1, (OUT, IN)).astype(np.float32)
    y = rng.uniform(0.05, 0.95, (B, OUT)).astype(np.float32)
    label = (rng.uniform(0, 1, (B, OUT)) > 0.5).astype(np.float32)
    flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X
    mf = capi.METRIC_ACCURACY | capi.METRIC_MSE
    res = []
    for fused in (True, False):
        dx = torch.full((B, IN), 3.0, dtype=torch.float32, device=DEV)
        dy = torch.full((B, OUT), 9.0, dtype=torch.float32, device=DEV)
        dw, db = torch.zeros(OUT, IN, device=DEV), torch.zeros(OUT, device=DEV)
        perf = torch.zeros(8, dtype=torch.int32, device=DEV)
        if fused:
            hip.call("ffh_linear_bwd_mse", dev(x), IN, dx, IN, dev(y), OUT, dy, OUT, dev(w), dw, db, IN, OUT, B, act, flags,
                     dev(label), 1.0 / B, perf, mf, None)
        else:
            hip.call("ffh_mse_bwd_metrics", dy, dev(y), dev(label), perf, B, OUT, 1.0 / B, mf, None)
            hip.call("ffh_linear_bwd_ex", dev(x), IN, dx, IN, dev(y), OUT, dy, OUT, dev(w), dw, db, IN, OUT, B, act, flags, None, None)
        res.append([host(t) for t in (dx, dy, dw, db, perf)])
    f, t = res
    assert bits_equal(f[0], t[0]) and bits_equal(f[1], t[1])
    np.testing.assert_allclose(f[2], t[2], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f[3], t[3], rtol=1e-5, atol=1e-6)
    assert f[4][:2].tolist() == t[4][:2].tolist()
    np.testing.assert_allclose(f[4][4:5].view(np.float32), t[4][4:5].view(np.float32), rtol=1e-5)
    # the oracle's own restatement
    lg = oracle.mse_bwd(y, label, 1.0 / B)
    odx, odw, odb, ody = oracle.linear_bwd_ex(x, y, lg, w, act, flags)
    assert bits_equal(f[1], ody)
    np.testing.assert_allclose(f[0], odx, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(f[2], odw, rtol=1e-5, atol=1e-6)
    with pytest.raises(capi.FFHError):     # a layer the one-launch backward does not serve: nothing launched, caller makes two calls
        hip.call("ffh_linear_bwd_mse", dev(x), IN, dx, IN, dev(y), OUT, dy, OUT, dev(w), dw, db, IN, OUT, B, act, capi.LINEAR_ONLY_DX,
                 dev(label), 1.0 / B, perf, mf, None)


@pytest.mark.parametrize("B,INL,MID,OUTU,act_l,act_u", [
    (2048, 256, 64, 16, capi.AC_MODE_RELU, capi.AC_MODE_RELU), (100, 256, 32, 5, capi.AC_MODE_NONE, capi.AC_MODE_SIGMOID),
    (33, 128, 64, 1, capi.AC_MODE_SIGMOID, capi.AC_MODE_NONE), (4097, 512, 64, 16, capi.AC_MODE_RELU, capi.AC_MODE_RELU)])
def test_linear_pair_fwd_equals_the_two_calls(hip, oracle, B, INL, MID, OUTU, act_l, act_u):
    """ffh_linear_pair_fwd (two narrow layers forward in one launch, the middle activation kept in LDS) against two
    ffh_linear_fwd calls on the GPU and in the oracle; the upper output goes into a wider buffer (a concat slice)."""
    rng = np.random.default_rng(B + INL + MID)
    x = rng.uniform(-1, 1, (B, INL)).astype(np.float32)
    wl = (rng.uniform(-1, 1, (MID, INL)) / np.sqrt(INL)).astype(np.float32)
    bl = rng.uniform(-1, 1, MID).astype(np.float32)
    wu = (rng.uniform(-1, 1, (OUTU, MID)) / np.sqrt(MID)).astype(np.float32)
    bu = rng.uniform(-1, 1, OUTU).astype(np.float32)
    ldu = OUTU + 7
    yl = torch.full((B, MID), 9.0, dtype=torch.float32, device=DEV)
    yu = torch.full((B, ldu), 777.0, dtype=torch.float32, device=DEV)
    hip.call("ffh_linear_pair_fwd", dev(x), INL, dev(wl), dev(bl), INL, act_l, yl, MID, MID, dev(wu), dev(bu), OUTU, act_u, yu[:, 3:], ldu, B, None)
    yl_e = oracle.linear_fwd(x, wl, bl, act_l)
    yu_e = oracle.linear_fwd(yl_e, wu, bu, act_u)
    assert_gemm_close(host(yl), yl_e, np.abs(x).astype(np.float64) @ np.abs(wl).astype(np.float64).T + np.abs(bl), "y_l")
    got = host(yu)
    assert_gemm_close(got[:, 3:3 + OUTU], yu_e, np.abs(yl_e).astype(np.float64) @ np.abs(wu).astype(np.float64).T + np.abs(bu) + 1e-2, "y_u")
    assert (got[:, :3] == 777).all() and (got[:, 3 + OUTU:] == 777).all()
    with pytest.raises(capi.FFHError):                             # mid = 48: not served
        hip.call("ffh_linear_pair_fwd", dev(x), INL, dev(wl), dev(bl), INL, act_l, yl, MID, 48, dev(wu), dev(bu), OUTU, act_u, yu, ldu, B, None)


@pytest.mark.parametrize("B,INL,INU,OUTU,act_u,act_l,fu,fl", [
    (2048, 256, 64, 16, capi.AC_MODE_RELU, capi.AC_MODE_RELU, capi.LINEAR_DY_PREMASKED, capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X),
    (2048, 256, 64, 16, capi.AC_MODE_RELU, capi.AC_MODE_RELU, 0, capi.LINEAR_DX_OVERWRITE),
    (100, 96, 32, 5, capi.AC_MODE_SIGMOID, capi.AC_MODE_NONE, 0, 0),
    (33, 32, 64, 16, capi.AC_MODE_NONE, capi.AC_MODE_RELU, 0, capi.LINEAR_DX_MASK_BY_X),
    (4097, 512, 64, 1, capi.AC_MODE_RELU, capi.AC_MODE_RELU, 0, capi.LINEAR_DX_OVERWRITE)])
def test_linear_pair_bwd_equals_the_two_calls(hip, oracle, B, INL, INU, OUTU, act_u, act_l, fu, fl):
    """ffh_linear_pair_bwd (upper layer's backward + lower layer's dX in one launch, the gradient between them kept in
    LDS) against the two ffh_linear_bwd_ex calls it stands for, on the GPU and in the oracle: activation gradients
    bit-exact, everything that sums (dX_l, dW_u, db_u, dy_l) within 1e-5 of its absolute mass."""
    rng = np.random.default_rng(B + INL + OUTU)
    xl = rng.uniform(-1, 1, (B, INL)).astype(np.float32)
    xu = rng.uniform(-1, 1, (B, INU)).astype(np.float32)          # the lower layer's output (relu: some entries <= 0)
    yu = rng.uniform(-1, 1, (B, OUTU)).astype(np.float32)
    if act_u == capi.AC_MODE_SIGMOID:
        yu = np.abs(yu) * 0.9 + 0.05
    gu = rng.uniform(-1, 1, (B, OUTU)).astype(np.float32)
    wu = (rng.uniform(-1, 1, (OUTU, INU)) / np.sqrt(INU)).astype(np.float32)
    wl = (rng.uniform(-1, 1, (INU, INL)) / np.sqrt(INL)).astype(np.float32)
    dx0 = rng.uniform(-1, 1, (B, INL)).astype(np.float32)         # accumulate form starts from this

    def run(fused):
        dyu, dxl = dev(gu), dev(dx0)
        dyl = torch.full((B, INU), 5.0, dtype=torch.float32, device=DEV)
        dwu, dbu = torch.zeros(OUTU, INU, device=DEV), torch.zeros(OUTU, device=DEV)
        dwl, dbl = torch.zeros(INU, INL, device=DEV), torch.zeros(INU, device=DEV)
        X = dict(xu=dev(xu), yu=dev(yu), wu=dev(wu), xl=dev(xl), wl=dev(wl))
        if fused:
            hip.call("ffh_linear_pair_bwd", X["xu"], INU, X["yu"], OUTU, dyu, OUTU, X["wu"], dwu, dbu, INU, OUTU, act_u, fu,
                     X["xl"], INL, dxl, INL, dyl, INU, X["wl"], INL, act_l, fl, B, None)
        else:
            up = fu | capi.LINEAR_DX_OVERWRITE | (capi.LINEAR_DX_MASK_BY_X if act_l == capi.AC_MODE_RELU else 0)
            hip.call("ffh_linear_bwd_ex", X["xu"], INU, dyl, INU, X["yu"], OUTU, dyu, OUTU, X["wu"], dwu, dbu, INU, OUTU, B, act_u, up, None, None)
            hip.call("ffh_linear_bwd_ex", X["xl"], INL, dxl, INL, X["xu"], INU, dyl, INU, X["wl"], dwl, dbl, INL, INU, B, act_l,
                     fl | capi.LINEAR_ONLY_DX | capi.LINEAR_DY_PREMASKED, None, None)
        return [host(t) for t in (dyu, dyl, dxl, dwu, dbu)]

    f, t = run(True), run(False)
    assert bits_equal(f[0], t[0])                                  # dy_u after the upper activation gradient
    au = np.abs(t[0]).astype(np.float64)
    assert_gemm_close(f[1], t[1], au @ np.abs(wu).astype(np.float64), "dy_l")
    al = np.abs(t[1]).astype(np.float64)
    assert_gemm_close(f[2], t[2], al @ np.abs(wl).astype(np.float64) + np.abs(dx0), "dx_l")
    assert_gemm_close(f[3], t[3], au.T @ np.abs(xu).astype(np.float64), "dw_u")
    assert_gemm_close(f[4], t[4], au.sum(0), "db_u")
    if act_l == capi.AC_MODE_RELU:
        assert (f[1][xu <= 0] == 0).all()                          # premasked for the lower layer
    with pytest.raises(capi.FFHError):                             # in_u = 48: not served, nothing launched
        hip.call("ffh_linear_pair_bwd", dev(xu), INU, dev(yu), OUTU, dev(gu), OUTU, dev(wu), torch.zeros(OUTU, INU, device=DEV), None, 48, OUTU,
                 act_u, fu, dev(xl), INL, dev(dx0), INL, torch.zeros(B, INU, device=DEV), INU, dev(wl), INL, act_l, fl, B, None)


# ---------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("B", [32768, 65536])
def test_full_size_terabyte_shape_properties(hip, ws, B):
    """Criteo-Terabyte / MLPerf per-table shape (B = 32768 and 65536, D = 128, R = 39,884,406 rows = 20.4 GB):
    gather == pure row copy (checked with torch indexing), fused update touches exactly the
    indexed rows, is linear in the gradient, and leaves every other row's bits alone."""
    D, R, lr = 128, 39884406, 0.01
    W = torch.empty(R, D, device=DEV)
    hip.call("ffh_init_uniform", W, R * D, 1, -(1.0 / R) ** 0.5, (1.0 / R) ** 0.5, None)
    idx = torch.empty(B, 1, dtype=torch.int64, device=DEV)
    hip.call("ffh_gen_indices", idx, B, 2, 0, R, None)
    sync()
    assert int(idx.min()) >= 0 and int(idx.max()) < R
    out = torch.empty(B, D, device=DEV)
    hip.call("ffh_embedding_fwd", idx, out, W, 1, D, B, R, D, capi.AGGR_MODE_SUM, None)
    sync()
    assert torch.equal(out, W[idx[:, 0]])                              # bit-exact gather at full size
    g = torch.empty(B, D, device=DEV)
    hip.call("ffh_gen_uniform01", g, B * D, 3, 0, None)
    rows = torch.unique(idx)
    before = W[rows].clone()
    csum_before = W.sum(dtype=torch.float64)
    hip.call("ffh_embedding_bwd_sgd_fused", idx, g, W, 1, D, B, R, D, capi.AGGR_MODE_SUM, lr, None)
    sync()
    # linearity / checksum: sum(W_after) - sum(W_before) == -lr * sum(g)
    delta = float(W.sum(dtype=torch.float64) - csum_before)
    expect = -lr * float(g.sum(dtype=torch.float64))
    assert abs(delta - expect) <= 1e-5 * lr * float(g.abs().sum(dtype=torch.float64)) + 1e-3
    # per-row: W_after[row] = W_before[row] - lr * sum of its gradients (float64 check, 1e-5 rel)
    acc = torch.zeros(rows.numel(), D, dtype=torch.float64, device=DEV)
    inv = torch.searchsorted(rows, idx[:, 0])
    acc.index_add_(0, inv, g.double())
    exp = before.double() - lr * acc
    assert torch.all((W[rows].double() - exp).abs() <= 1e-5 * (before.abs().double() + lr * acc.abs()) + 1e-12)
    # untouched rows keep their bits: regenerate the first 4096 rows of the same counter-based stream
    lo, hi = -(1.0 / R) ** 0.5, (1.0 / R) ** 0.5
    full = torch.empty(4096, D, device=DEV)
    hip.call("ffh_init_uniform", full, 4096 * D, 1, lo, hi, None)     # first 4096 rows of the same stream
    sync()
    small = torch.arange(0, 4096, device=DEV)
    small = small[~torch.isin(small, rows)]
    assert torch.equal(W[small], full[small])                          # untouched rows keep their bits


def test_full_size_kaggle_step_shapes(hip, oracle, ws):
    """Criteo-Kaggle shape (C2): all 26 tables with their true row counts, B = 2048, D = 16.
    Forward into the concat buffer and fused update, each against the oracle bit for bit."""
    rows = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
            10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
    B, D = 2048, 16
    Wt, It = [], []
    Z = torch.zeros(B, 16 + 26 * D, device=DEV)
    G = torch.empty(B, 16 + 26 * D, device=DEV)
    hip.call("ffh_gen_uniform01", G, G.numel(), 77, 0, None)
    for t, R in enumerate(rows):
        w = torch.empty(R, D, device=DEV)
        hip.call("ffh_init_uniform", w, R * D, 100 + t, -(1.0 / R) ** 0.5, (1.0 / R) ** 0.5, None)
        i = torch.empty(B, 1, dtype=torch.int64, device=DEV)
        hip.call("ffh_gen_indices", i, B, 200 + t, 0, R, None)
        Wt.append(w); It.append(i)
    arr = hip.emb_tables([(It[t], Wt[t], Z[:, 16 + t * D:], rows[t], Z.shape[1]) for t in range(26)])
    hip.check(hip.lib.ffh_embedding_fwd_multi(hip.ctx, arr, 26, 1, D, B, capi.AGGR_MODE_SUM, None), "fwd")
    sync()
    for t in range(26):
        assert torch.equal(Z[:, 16 + t * D:16 + (t + 1) * D], Wt[t][It[t][:, 0]])
    before = [Wt[t][torch.unique(It[t])].clone() for t in range(26)]
    arr = hip.emb_tables([(It[t], Wt[t], G[:, 16 + t * D:], rows[t], G.shape[1]) for t in range(26)])
    hip.check(hip.lib.ffh_embedding_bwd_sgd_fused_multi(hip.ctx, arr, 26, 1, D, B, capi.AGGR_MODE_SUM, 0.01, None), "bwd")
    sync()
    Gh = host(G)
    for t in range(26):
        rws = torch.unique(It[t])
        idx_h = host(It[t])
        remap = np.searchsorted(host(rws), idx_h)
        gs = np.ascontiguousarray(Gh[:, 16 + t * D:16 + (t + 1) * D])
        exp = oracle.embedding_bwd_sgd_fused(remap, gs, host(before[t]), 0.01)
        assert bits_equal(host(Wt[t][rws]), exp), f"table {t} R={rows[t]}"
