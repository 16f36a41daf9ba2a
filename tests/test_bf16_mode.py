"""Tensor-op math mode (--allow-tensor-op-math-conversion -> ffh_ctx_set_math_mode(FFH_MATH_TENSOR_OP_BF16))
[ref: cublasSetMathMode(CUBLAS_TENSOR_OP_MATH), src/runtime/model.cu:81-83; call sites src/ops/linear.cu:436-453,624-659].

Numerics contract (include/ff_hip.h): Linear GEMMs with in_dim, out_dim >= 128 round BOTH operands to bfloat16
(nearest even), multiply exactly, accumulate in fp32; everything else is unchanged fp32.
  * CPU: the oracle in that mode against a numpy restatement of the rounding (independent code), and against the
    fp32 result inside the stated bound 2^-8 * sum |a_k b_k|;
  * GPU (-m gpu): the HIP kernels (v_mfma_f32_32x32x16_bf16) against the oracle IN THE SAME MODE at the fp32-mode
    tolerance (1e-5 of the term mass: only the fp32 summation order differs), forward / dX / dW, every flag form,
    ragged shapes; and the whole DLRM step with the driver flag, HIP vs the oracle backend.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import ROOT
from dlrm_flexflow_amd import capi

MATH_BF16 = 1


def bf16_round(a):
    """float32 -> bfloat16 (round to nearest even) -> float32, on the bits"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def act_fwd(v, act):
    if act == capi.AC_MODE_RELU:
        return np.maximum(v, 0)
    if act == capi.AC_MODE_SIGMOID:
        return 1 / (1 + np.exp(-v))
    return v


def emulated(x, w, b, gy, act):
    """float64 sums over bf16-rounded operands (products of two bf16 numbers are exact in fp32 and in float64)"""
    xb, wb = bf16_round(x).astype(np.float64), bf16_round(w).astype(np.float64)
    y = act_fwd(xb @ wb.T + b, act)
    yf = y.astype(np.float32)
    d = gy.astype(np.float64)
    if act == capi.AC_MODE_RELU:
        d = np.where(yf > 0, d, 0)
    elif act == capi.AC_MODE_SIGMOID:
        d = (gy * yf * (1 - yf)).astype(np.float64)
    df = d.astype(np.float32)
    db = df.astype(np.float64).sum(0)
    dB = bf16_round(df).astype(np.float64)
    return y, dB.T @ xb, db, dB @ wb, df


@pytest.fixture()
def oracle_bf16(oracle):
    lib = oracle.lib()
    assert lib.lib.ffh_ctx_set_math_mode(lib.ctx, MATH_BF16) == 0
    yield oracle
    assert lib.lib.ffh_ctx_set_math_mode(lib.ctx, 0) == 0


def test_math_mode_rejects_unknown_values(oracle):
    lib = oracle.lib()
    assert lib.lib.ffh_ctx_set_math_mode(lib.ctx, 7) == -1
    assert lib.lib.ffh_ctx_set_math_mode(lib.ctx, 0) == 0


@pytest.mark.parametrize("B,IN,OUT,act", [(96, 256, 128, capi.AC_MODE_RELU), (33, 130, 200, capi.AC_MODE_NONE), (64, 128, 128, capi.AC_MODE_SIGMOID)])
def test_oracle_bf16_mode_matches_numpy_emulation(oracle_bf16, B, IN, OUT, act):
    rng = np.random.default_rng(B + IN)
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    y = oracle_bf16.linear_fwd(x, w, b, act)
    y_e, dw_e, db_e, dx_e, dy_e = emulated(x, w, b, gy, act)
    np.testing.assert_allclose(y, y_e, rtol=2e-6, atol=2e-6)
    dx, dw, db, dy_after = oracle_bf16.linear_bwd(x, y, gy, w, act)
    np.testing.assert_allclose(dy_after, dy_e, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dw, dw_e, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(db, db_e, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(dx, dx_e, rtol=1e-5, atol=2e-5)


def test_oracle_bf16_mode_is_within_the_stated_bound_of_fp32_and_differs(oracle):
    rng = np.random.default_rng(1)
    B, IN, OUT = 64, 512, 256
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    y32 = oracle.linear_fwd(x, w, None)
    lib = oracle.lib()
    lib.lib.ffh_ctx_set_math_mode(lib.ctx, MATH_BF16)
    try:
        y16 = oracle.linear_fwd(x, w, None)
    finally:
        lib.lib.ffh_ctx_set_math_mode(lib.ctx, 0)
    mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
    err = np.abs(y16.astype(np.float64) - y32)
    assert np.all(err <= 2.0 ** -8 * mass + 1e-6)
    assert err.max() > 1e-5                      # the mode really rounds


def test_narrow_layers_stay_fp32_in_bf16_mode(oracle, oracle_bf16):
    """in_dim or out_dim below FFH_BF16_MIN_DIM (128): bit-identical to the fp32 mode (the pair / skinny layers of DLRM)."""
    rng = np.random.default_rng(2)
    for IN, OUT in ((127, 256), (256, 64), (13, 512)):
        x = rng.uniform(-1, 1, (40, IN)).astype(np.float32)
        w = rng.uniform(-1, 1, (OUT, IN)).astype(np.float32)
        y16 = oracle_bf16.linear_fwd(x, w, None)
        lib = oracle.lib()
        lib.lib.ffh_ctx_set_math_mode(lib.ctx, 0)
        y32 = oracle.linear_fwd(x, w, None)
        lib.lib.ffh_ctx_set_math_mode(lib.ctx, MATH_BF16)
        assert y16.tobytes() == y32.tobytes()


def test_driver_flag_reaches_the_kernels_cpu(oracle):
    """--allow-tensor-op-math-conversion through FFConfig -> ffh_ctx_set_math_mode: the whole step changes (bf16 GEMMs
    in the 256-wide layers) but stays within bf16 distance of the fp32 run."""
    from dlrm_flexflow_amd import ffmodel
    args = ["--backend", oracle.ORACLE_LIB, "-b", "64", "--arch-sparse-feature-size", "16", "--arch-embedding-size", "100-200-50",
            "--arch-mlp-bot", "13-256-128-16", "--arch-mlp-top", "64-256-128-1", "--data-size", "64"]
    preds = {}
    for flag in (False, True):
        app = ffmodel.DLRM(args + (["--allow-tensor-op-math-conversion"] if flag else []))
        app.warmup()
        app.train_steps(2, trace=False)
        app.model.sync()
        preds[flag] = app.model.layer_output(app.model.num_layers - 1).get()
        app.close()
    d = np.abs(preds[True] - preds[False]).max()
    assert 1e-7 < d < 2e-2, d


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
def _gpu_helpers():
    import test_gpu_parity as T
    return T


@pytest.fixture()
def hip_bf16(hip, oracle_bf16):
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, MATH_BF16) == 0
    yield hip
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,IN,OUT,act", [
    (2048, 512, 256, capi.AC_MODE_RELU), (2048, 432, 512, capi.AC_MODE_RELU), (4096, 1024, 1024, capi.AC_MODE_RELU),
    (1024, 3456, 1024, capi.AC_MODE_RELU), (512, 479, 1024, capi.AC_MODE_RELU), (512, 857, 1024, capi.AC_MODE_NONE),
    (333, 130, 200, capi.AC_MODE_SIGMOID), (65, 128, 128, capi.AC_MODE_NONE), (1000, 257, 129, capi.AC_MODE_RELU),
    (8192, 256, 128, capi.AC_MODE_RELU), (100, 2000, 1000, capi.AC_MODE_NONE),
    # outputs with >= 256 tiles of 256 x 256: the big-tile kernel (8 waves of 128 x 64), ragged on every edge; the second
    # one also in dX (reduction depth >= 1024)
    (16500, 200, 1000, capi.AC_MODE_RELU), (16500, 1000, 1030, capi.AC_MODE_RELU),
    # ... and the split-K weight gradient on big tiles (>= 32 of them, batch >= 8192)
    (8200, 1800, 1030, capi.AC_MODE_RELU)])
def test_linear_bf16_mode_hip_vs_oracle_same_mode(hip_bf16, oracle_bf16, B, IN, OUT, act):
    T = _gpu_helpers()
    rng = np.random.default_rng(IN * OUT + 1)
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    y = T.gpu_linear_fwd(hip_bf16, x, w, b, act)
    y_exp = oracle_bf16.linear_fwd(x, w, b, act)
    mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
    T.assert_gemm_close(y, y_exp, mass, "y (bf16 mode)")
    dx, dw, db, dy_after = T.gpu_linear_bwd(hip_bf16, x, y_exp, gy, w, act)
    dx_e, dw_e, db_e, dy_e = oracle_bf16.linear_bwd(x, y_exp, gy, w, act)
    np.testing.assert_allclose(dy_after, dy_e, rtol=1e-6, atol=1e-7)
    a = np.abs(dy_e).astype(np.float64)
    T.assert_gemm_close(dw, dw_e, a.T @ np.abs(x).astype(np.float64), "dw (bf16 mode)")
    T.assert_gemm_close(db, db_e, a.sum(0), "db (bf16 mode)")
    T.assert_gemm_close(dx, dx_e, a @ np.abs(w).astype(np.float64), "dx (bf16 mode)")
    # and the mode is really on: the fp32 oracle is further away than the summation-order tolerance somewhere
    lib = oracle_bf16.lib()
    lib.lib.ffh_ctx_set_math_mode(lib.ctx, 0)
    y32 = oracle_bf16.linear_fwd(x, w, b, act)
    lib.lib.ffh_ctx_set_math_mode(lib.ctx, MATH_BF16)
    err32 = np.abs(y.astype(np.float64) - y32)
    assert np.all(err32 <= 2.0 ** -8 * mass + 1e-5)
    if act != capi.AC_MODE_SIGMOID:
        assert err32.max() > 1e-5 * mass.max() / 50


@pytest.mark.gpu
@pytest.mark.parametrize("act", [capi.AC_MODE_RELU, capi.AC_MODE_SIGMOID, capi.AC_MODE_NONE])
def test_linear_bf16_mode_bwd_ex_forms(hip_bf16, oracle_bf16, act):
    """overwrite / accumulate dX, mask-by-x, premasked dy, ONLY_DX + ONLY_DW split, forked dW stream, strided operands."""
    import torch
    T = _gpu_helpers()
    rng = np.random.default_rng(7 + act)
    B, IN, OUT = 700, 384, 256
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    y = oracle_bf16.linear_fwd(x, w, None, act)
    dx0 = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    a = np.abs(gy).astype(np.float64)
    mdx, mdw = a @ np.abs(w).astype(np.float64), a.T @ np.abs(x).astype(np.float64)
    s2 = torch.cuda.Stream()
    for flags in (0, capi.LINEAR_DX_OVERWRITE, capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X, capi.LINEAR_DY_PREMASKED):
        for forked in (False, True):
            dx = T.dev(dx0); dw = torch.zeros(OUT, IN, device=T.DEV); db = torch.zeros(OUT, device=T.DEV); dy = T.dev(gy)
            hip_bf16.call("ffh_linear_bwd_ex", T.dev(x), IN, dx, IN, T.dev(y), OUT, dy, OUT, T.dev(w), dw, db, IN, OUT, B, act, flags,
                          None, s2.cuda_stream if forked else None)
            torch.cuda.synchronize()
            dx_e, dw_e, db_e, dy_e = oracle_bf16.linear_bwd_ex(x, y, gy, w, act, flags, dx0=dx0)
            T.assert_gemm_close(T.host(dx), dx_e, mdx + np.abs(dx0), f"dx flags={flags} forked={forked}")
            T.assert_gemm_close(T.host(dw), dw_e, mdw, f"dw flags={flags}")
            T.assert_gemm_close(T.host(db), db_e, a.sum(0), f"db flags={flags}")
            np.testing.assert_allclose(T.host(dy), dy_e, rtol=1e-6, atol=1e-7)
    # the split pair on two streams equals the one call
    dx = torch.zeros(B, IN, device=T.DEV); dw = torch.zeros(OUT, IN, device=T.DEV); db = torch.zeros(OUT, device=T.DEV); dy = T.dev(gy)
    hip_bf16.call("ffh_linear_bwd_ex", T.dev(x), IN, dx, IN, T.dev(y), OUT, dy, OUT, T.dev(w), dw, db, IN, OUT, B, act, capi.LINEAR_ONLY_DX, None, None)
    torch.cuda.synchronize()
    hip_bf16.call("ffh_linear_bwd_ex", T.dev(x), IN, dx, IN, T.dev(y), OUT, dy, OUT, T.dev(w), dw, db, IN, OUT, B, act, capi.LINEAR_ONLY_DW, None, None)
    torch.cuda.synchronize()
    dx_e, dw_e, db_e, _ = oracle_bf16.linear_bwd_ex(x, y, gy, w, act, 0)
    T.assert_gemm_close(T.host(dx), dx_e, mdx, "dx split")
    T.assert_gemm_close(T.host(dw), dw_e, mdw, "dw split")
    T.assert_gemm_close(T.host(db), db_e, a.sum(0), "db split")


@pytest.mark.gpu
@pytest.mark.parametrize("trace", [False, True])
def test_dlrm_step_bf16_mode_hip_vs_oracle_backend(hip, oracle, trace):
    """The driver flag end to end at the Kaggle widths (432->512, 512->256 layers take bf16 operands): 1 warm-up + 3 steps
    on the GPU vs the same host code on the oracle, both in tensor-op mode; and the fp32 run differs."""
    import dlrm_helpers as H
    from dlrm_flexflow_amd import ffmodel
    args = H.KAGGLE_ARGS(2048) + ["--allow-tensor-op-math-conversion"]
    out = {}
    for name, backend in (("hip", capi.HIP_LIB_PATH), ("cpu", oracle.ORACLE_LIB)):
        app = ffmodel.DLRM(["--backend", backend] + args)
        app.warmup()
        app.train_steps(3, trace=trace and name == "hip")
        app.model.sync()
        m = app.model
        out[name] = {f"{m.layer_name(l)}/{i}": m.parameter(l, i).get_weights() for l in range(m.num_layers) for i in range(m.layer_num_weights(l))}
        out[name]["pred"] = m.layer_output(m.num_layers - 1).get()
        app.close()
    # Chained layers: an fp32 summation-order difference of one ulp in an activation can land on the other side of a bf16
    # rounding boundary of the NEXT layer's operand (a 2^-9 relative step in that one operand).  So: at least 99 % of the
    # elements inside the fp32-mode tolerance, every element inside a bf16-step-sized one.
    for k in out["hip"]:
        g, e = out["hip"][k].astype(np.float64), out["cpu"][k].astype(np.float64)
        tight = np.abs(g - e) <= 2e-5 * np.abs(e) + 2e-6
        assert tight.mean() >= 0.99, (k, float(tight.mean()))
        np.testing.assert_allclose(g, e, rtol=2e-3, atol=2e-4, err_msg=k)
    app = ffmodel.DLRM(["--backend", capi.HIP_LIB_PATH] + H.KAGGLE_ARGS(2048))
    app.warmup(); app.train_steps(3, trace=False); app.model.sync()
    p32 = app.model.layer_output(app.model.num_layers - 1).get()
    app.close()
    d = np.abs(p32 - out["hip"]["pred"]).max()
    assert 1e-7 < d < 2e-2, d


# ---------------------------------------------------------------------------------------------------------------------
# FFH_MATH_FP32_SPLIT_BF16X3: fp32-accurate GEMMs on the bf16 pipe (three bf16 terms per operand, six products).  Held to the
# SAME bound as the exact-fp32 kernels against the fp32 oracle, and its error against float64 must be of the same size.
# ---------------------------------------------------------------------------------------------------------------------
MATH_X3 = 3        # FFH_MATH_FP32_SPLIT_BF16X3_ALL: every wide layer on the split kernels, whatever its size (the mode proper, 2, leaves small GEMMs to the fp32 kernels)


@pytest.fixture()
def hip_x3(hip):
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, MATH_X3) == 0
    yield hip
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,IN,OUT,act", [
    (2048, 512, 256, capi.AC_MODE_RELU), (2048, 432, 512, capi.AC_MODE_RELU), (4096, 1024, 1024, capi.AC_MODE_RELU),
    (1024, 3456, 1024, capi.AC_MODE_RELU), (512, 479, 1024, capi.AC_MODE_RELU), (333, 130, 200, capi.AC_MODE_SIGMOID),
    (65, 128, 128, capi.AC_MODE_NONE), (1000, 257, 129, capi.AC_MODE_RELU), (100, 2000, 1000, capi.AC_MODE_NONE),
    # the 256 x 256 tile: forward, deep-K dX, and the split-K weight gradient with >= 32 tiles; ragged on every edge
    (16500, 200, 1000, capi.AC_MODE_RELU), (16500, 1000, 1030, capi.AC_MODE_RELU), (8200, 1800, 1030, capi.AC_MODE_RELU)])
def test_linear_split_bf16x3_mode_meets_the_fp32_bound(hip_x3, oracle, B, IN, OUT, act):
    T = _gpu_helpers()
    rng = np.random.default_rng(IN * OUT + 5)
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    y = T.gpu_linear_fwd(hip_x3, x, w, b, act)
    y_exp = oracle.linear_fwd(x, w, b, act)                                   # the fp32 oracle, default mode
    mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
    T.assert_gemm_close(y, y_exp, mass, "y (split mode vs fp32 oracle)")
    dx, dw, db, dy_after = T.gpu_linear_bwd(hip_x3, x, y_exp, gy, w, act)
    dx_e, dw_e, db_e, dy_e = oracle.linear_bwd(x, y_exp, gy, w, act)
    a = np.abs(dy_e).astype(np.float64)
    T.assert_gemm_close(dw, dw_e, a.T @ np.abs(x).astype(np.float64), "dw (split mode)")
    T.assert_gemm_close(db, db_e, a.sum(0), "db (split mode)")
    T.assert_gemm_close(dx, dx_e, a @ np.abs(w).astype(np.float64), "dx (split mode)")


@pytest.mark.gpu
def test_split_bf16x3_error_against_float64_is_fp32_sized(hip, oracle):
    """|error vs float64| of the split mode next to the exact-fp32 kernel's on the same operands, wide dynamic range
    included (entries spread over 2^-20 .. 2^20): the split mode's worst error stays within 4x the fp32 kernel's."""
    T = _gpu_helpers()
    rng = np.random.default_rng(11)
    B, IN, OUT = 512, 1024, 256
    for spread in (0, 20):
        x = (rng.uniform(-1, 1, (B, IN)) * 2.0 ** rng.integers(-spread, spread + 1, (B, IN))).astype(np.float32)
        w = (rng.uniform(-1, 1, (OUT, IN)) * 2.0 ** rng.integers(-spread, spread + 1, (OUT, IN))).astype(np.float32)
        exact = x.astype(np.float64) @ w.astype(np.float64).T
        mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
        err = {}
        for mode in (0, MATH_X3):
            assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, mode) == 0
            y = T.gpu_linear_fwd(hip, x, w, None, capi.AC_MODE_NONE)
            err[mode] = float((np.abs(y - exact) / mass).max())
        hip.lib.ffh_ctx_set_math_mode(hip.ctx, 0)
        assert err[MATH_X3] <= 4 * err[0] + 1e-8, (spread, err)
        assert err[MATH_X3] < 2e-6, (spread, err)


@pytest.mark.gpu
def test_dlrm_step_split_bf16x3_mode_matches_the_fp32_oracle_backend(hip, oracle):
    """--fp32-split-bf16x3 end to end at the Kaggle widths: 1 warm-up + 3 steps on the GPU against the oracle backend in its
    DEFAULT fp32 mode, at the tolerance of the fp32 driver tests."""
    import dlrm_helpers as H
    from dlrm_flexflow_amd import ffmodel
    args = H.KAGGLE_ARGS(2048)
    out = {}
    for name, backend, extra in (("hip", capi.HIP_LIB_PATH, ["--fp32-split-bf16x3"]), ("cpu", oracle.ORACLE_LIB, [])):
        app = ffmodel.DLRM(["--backend", backend] + args + extra)
        app.warmup()
        app.train_steps(3, trace=False)
        app.model.sync()
        m = app.model
        out[name] = {f"{m.layer_name(l)}/{i}": m.parameter(l, i).get_weights() for l in range(m.num_layers) for i in range(m.layer_num_weights(l))}
        out[name]["pred"] = m.layer_output(m.num_layers - 1).get()
        app.close()
    for k in out["hip"]:
        np.testing.assert_allclose(out["hip"][k], out["cpu"][k], rtol=2e-5, atol=2e-6, err_msg=k)


# ---------------------------------------------------------------------------------------------------------------------
# bf16 twins (ffh_ctx_bf16_mirror_set, ABI 6): in tensor-op mode a registered fp32 buffer has a bfloat16 twin that its writers
# keep current and the bf16-pipe GEMMs read instead.  The twin of x is the value the kernel rounds x to anyway, so every result
# must be BIT-identical with and without twins (fp32 outputs) -- only the weight gradient's atomic order may differ.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("B,IN,OUT", [(4096, 1024, 1024), (2048, 512, 256), (8192, 3456, 1024), (4160, 3456, 1000), (576, 1024, 136), (16640, 1024, 1096)])
def test_bf16_twins_give_the_same_bits_as_in_kernel_rounding(hip_bf16, B, IN, OUT):
    import torch
    hip = hip_bf16
    dev = "cuda:0"
    rng = np.random.default_rng(B + OUT)
    x = torch.from_numpy(np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-1, 1, OUT).astype(np.float32)).to(dev)
    gy = torch.from_numpy(rng.uniform(-1, 1, (B, OUT)).astype(np.float32)).to(dev)
    route = lambda: hip.lib.ffh_linear_last_route(hip.ctx).decode()

    def run(with_twins):
        y = torch.full((B, OUT), 3.0, device=dev); dx = torch.full((B, IN), 9.0, device=dev)
        dw = torch.zeros(OUT, IN, device=dev); db = torch.zeros(OUT, device=dev); dy = gy.clone()
        tw = {}
        if with_twins:
            for name, t in (("x", x), ("w", w), ("y", y), ("dy", dy), ("dx", dx)):
                tw[name] = torch.zeros(t.shape, dtype=torch.bfloat16, device=dev)
                assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, tw[name].data_ptr()) == 0
            hip.call("ffh_convert_f32_to_bf16", tw["x"], x, x.numel(), None)       # the producers of x, w, dy in a model
            hip.call("ffh_convert_f32_to_bf16", tw["w"], w, w.numel(), None)
            hip.call("ffh_convert_f32_to_bf16", tw["dy"], dy, dy.numel(), None)
        hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_RELU, None)
        r_f = route()
        flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DY_PREMASKED | capi.LINEAR_DX_MASK_BY_X
        hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, flags, None, None)
        r_b = route()
        torch.cuda.synchronize()
        if with_twins:
            for t in (x, w, y, dy, dx):
                assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, None) == 0
        return y, dx, dw, db, tw, r_f, r_b

    y0, dx0, dw0, db0, _, rf0, rb0 = run(False)
    y1, dx1, dw1, db1, tw, rf1, rb1 = run(True)
    # twins serve a GEMM whose reduction depth is a multiple of 64 (forward: IN, dX: OUT, dW: the batch); edge tiles in M / N are fine
    want = (IN % 64 == 0, OUT % 64 == 0, B % 64 == 0)
    tok = {t.split(" gemm")[0].split("linear_bwd ")[-1]: t for t in rb1.split(";")}
    assert "twins" not in rf0 + rb0, (rf0, rb0)
    assert ("twins" in rf1) == want[0] and ("twins" in tok["dx"]) == want[1] and ("twins" in tok["dw"]) == want[2], (rf1, rb1)
    # same operands; the same kernel structure gives the same bits, the LDS-DMA kernel (outputs of >= one 256 x 256 tile per CU)
    # sums in its own order: 1e-5 of the term mass
    ymass = (x.abs().double() @ w.abs().double().T + b.abs().double()).cpu().numpy()
    dxmass = (gy.abs().double() @ w.abs().double()).cpu().numpy()
    if "bf16_dma" in rf1: np.testing.assert_array_less(np.abs((y1 - y0).cpu().numpy()), 1e-5 * ymass + 1e-6)
    else: assert torch.equal(y0, y1)
    if "bf16_dma" in tok["dx"]: np.testing.assert_array_less(np.abs((dx1 - dx0).cpu().numpy()), 1e-5 * dxmass + 1e-6)
    else: assert torch.equal(dx0, dx1)
    assert torch.equal(tw["y"], y1.to(torch.bfloat16)) and torch.equal(tw["dx"], dx1.to(torch.bfloat16))   # nearest-even, as torch rounds
    assert torch.equal(tw["x"], x.to(torch.bfloat16))
    mass = (gy.abs().double().T @ x.abs().double()).cpu().numpy()
    np.testing.assert_array_less(np.abs((dw1 - dw0).cpu().numpy()), 2e-5 * mass + 1e-6)      # split-K atomics: order differs run to run
    np.testing.assert_array_less(np.abs((db1 - db0).cpu().numpy()), 2e-5 * gy.abs().double().sum(0).cpu().numpy() + 1e-6)


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("B,IN,OUT", [(32768, 3456, 1024), (32768, 1024, 1024), (32768, 1024, 512), (16704, 1088, 1344)])
def test_bf16_dma_kernel_layers_of_the_benched_step_vs_emulation(hip_bf16, B, IN, OUT):
    """The LDS-DMA kernel (csrc/linear_bf16_dma.hip) on the layers it serves in the benched step (batch 32768, twins registered as
    the model registers them) and on a ragged shape (edge tiles in rows and columns, partial last round): forward with bias +
    relu, then the model's backward form (premasked dy, dX stored and masked by relu'(x), weight gradient on its own stream)
    against float64 sums over the bf16-rounded operands (the numpy restatement the oracle's tensor-op mode is pinned to in
    test_oracle_bf16_mode_matches_numpy_emulation) at 1e-5 of the term mass; the twins of y and dX are the nearest-even
    roundings of the fp32 results; the route names the kernel for all three GEMMs."""
    import torch
    hip = hip_bf16
    dev = "cuda:0"
    rng = np.random.default_rng(B + IN + OUT)
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    xb, wb = bf16_round(x).astype(np.float64), bf16_round(w).astype(np.float64)
    y_e = np.maximum(xb @ wb.T + b, 0)
    dy_e = np.where(y_e.astype(np.float32) > 0, gy, 0).astype(np.float32)      # premasked by the layer above
    dB = bf16_round(dy_e).astype(np.float64)
    dw_e, db_e = dB.T @ xb, dy_e.astype(np.float64).sum(0)
    dx_e = np.where(x > 0, dB @ wb, 0)
    ax, aw, ad = np.abs(x).astype(np.float32), np.abs(w).astype(np.float32), np.abs(dy_e)
    xd, wd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    y = torch.full((B, OUT), 3.0, device=dev); dx = torch.full((B, IN), 9.0, device=dev)
    dw = torch.zeros(OUT, IN, device=dev); db = torch.zeros(OUT, device=dev); dy = torch.from_numpy(dy_e).to(dev)
    tw = {}
    for name, t in (("x", xd), ("w", wd), ("y", y), ("dy", dy), ("dx", dx)):
        tw[name] = torch.zeros(t.shape, dtype=torch.bfloat16, device=dev)
        assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, tw[name].data_ptr()) == 0
    try:
        for name, t in (("x", xd), ("w", wd), ("dy", dy)):
            hip.call("ffh_convert_f32_to_bf16", tw[name], t, t.numel(), None)
        hip.call("ffh_linear_fwd", xd, IN, y, OUT, wd, bd, IN, OUT, B, capi.AC_MODE_RELU, None)
        r_f = hip.lib.ffh_linear_last_route(hip.ctx).decode()
        flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DY_PREMASKED | capi.LINEAR_DX_MASK_BY_X
        s2 = torch.cuda.Stream()
        hip.call("ffh_linear_bwd_ex", xd, IN, dx, IN, y, OUT, dy, OUT, wd, dw, db, IN, OUT, B, capi.AC_MODE_RELU, flags, None, s2.cuda_stream)
        r_b = hip.lib.ffh_linear_last_route(hip.ctx).decode()
        torch.cuda.synchronize()
    finally:
        for t in (xd, wd, y, dy, dx):
            assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, None) == 0
    print("routes:", r_f, "|", r_b)
    assert "bf16_dma_256x256_twins" in r_f and r_b.count("bf16_dma_256x256_twins") == 2, (r_f, r_b)
    def close(got, exp, mass, what):
        bad = np.abs(got.astype(np.float64) - exp) > 1e-5 * mass + 1e-6
        assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} off, worst {np.abs(got - exp).max():.3e}"
    yh = y.cpu().numpy()
    close(yh, y_e, (ax @ aw.T).astype(np.float64) + np.abs(b), "y")
    close(dx.cpu().numpy(), dx_e, (ad @ aw).astype(np.float64), "dx")
    close(dw.cpu().numpy(), dw_e, (ad.T @ ax).astype(np.float64), "dw")
    close(db.cpu().numpy(), db_e, ad.astype(np.float64).sum(0), "db")
    assert torch.equal(tw["y"], y.to(torch.bfloat16)) and torch.equal(tw["dx"], dx.to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize("IN,OUT,label", [(256, 1, True), (256, 1, False), (512, 4, False)])
def test_one_launch_backward_of_a_skinny_layer_writes_the_twin_of_its_data_gradient(hip_bf16, oracle, IN, OUT, label):
    """The 256 -> 1 layer on top of the Terabyte MLP stays fp32 in tensor-op mode (out < 128), but its input gradient is the
    dy of the 512 -> 256 layer below, which reads bf16: the one-launch backward (with and without the loss step folded in)
    writes the registered twin of dX beside dX -- the nearest-even rounding of exactly the fp32 values it stores, which
    themselves equal the oracle's."""
    import torch
    hip = hip_bf16
    dev = "cuda:0"
    B = 4100
    rng = np.random.default_rng(IN + OUT)
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    y = oracle.linear_fwd(x, w, None, capi.AC_MODE_SIGMOID if label else capi.AC_MODE_NONE)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    lab = rng.integers(0, 2, (B, OUT)).astype(np.float32)
    xd, wd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(y).to(dev)
    dx = torch.full((B, IN), 9.0, device=dev); dw = torch.zeros(OUT, IN, device=dev); db = torch.zeros(OUT, device=dev)
    dy = torch.from_numpy(gy).to(dev)
    tw = torch.zeros(B, IN, dtype=torch.bfloat16, device=dev)
    assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, dx.data_ptr(), dx.numel() * 4, tw.data_ptr()) == 0
    flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X
    try:
        if label:
            perf = torch.zeros(64, dtype=torch.uint8, device=dev)
            hip.call("ffh_linear_bwd_mse", xd, IN, dx, IN, yd, OUT, dy, OUT, wd, dw, db, IN, OUT, B, capi.AC_MODE_SIGMOID, flags,
                     torch.from_numpy(lab).to(dev), 1.0 / B, perf, 0, None)
        else:
            hip.call("ffh_linear_bwd_ex", xd, IN, dx, IN, yd, OUT, dy, OUT, wd, dw, db, IN, OUT, B, capi.AC_MODE_NONE, flags, None, None)
        assert "skinny" in hip.lib.ffh_linear_last_route(hip.ctx).decode()
        torch.cuda.synchronize()
    finally:
        assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, dx.data_ptr(), dx.numel() * 4, None) == 0
    assert torch.equal(tw, dx.to(torch.bfloat16))
    if not label:
        dx_e = oracle.linear_bwd_ex(x, y, gy, w, capi.AC_MODE_NONE, flags, dx0=None)[0]
        mass = np.abs(gy).astype(np.float64) @ np.abs(w).astype(np.float64)
        assert np.all(np.abs(dx.cpu().numpy().astype(np.float64) - dx_e) <= 1e-5 * mass + 1e-6)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_dlrm_step_bf16_mode_twins_on_equals_twins_off(hip):
    """Whole model in tensor-op mode at the Terabyte widths (top 3456-1024-1024-512-256-1: every big layer reads twins, the gather
    writes the twin of the Concat output, the optimizer the weights' twin), batch 4096, rows capped: three steps with the twins
    against the same run with --no-bf16-twins (operands rounded inside the kernels).  Identical arithmetic; the only
    freedom is the atomic order of the weight gradients."""
    import dlrm_helpers as H
    from dlrm_flexflow_amd import ffmodel
    rows = "-".join(str(min(r, 50000)) for r in [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155, 4, 976, 14,
                                                 39979771, 25641295, 39664984, 585935, 12972, 108, 36])
    args = ["--backend", capi.HIP_LIB_PATH, "-b", "4096", "--arch-sparse-feature-size", "128", "--arch-embedding-size", rows, "--arch-mlp-bot", "13-512-256-128",
            "--arch-mlp-top", "3456-1024-1024-512-256-1", "--data-size", "4096", "--allow-tensor-op-math-conversion"]
    out = []
    for off in (False, True):
        app = ffmodel.DLRM(args + (["--no-bf16-twins"] if off else []))
        app.warmup(); app.train_steps(3, trace=False); app.model.sync()
        m = app.model
        o = {f"{m.layer_name(l)}/{i}": m.parameter(l, i).get_weights() for l in range(m.num_layers) for i in range(m.layer_num_weights(l))}
        o["pred"] = m.layer_output(m.num_layers - 1).get()
        out.append(o)
        app.close()
    for k in out[0]:
        g, e = out[0][k].astype(np.float64), out[1][k].astype(np.float64)
        tight = np.abs(g - e) <= 2e-5 * np.abs(e) + 2e-6
        assert tight.mean() >= 0.99, (k, float(tight.mean()))                  # a one-ulp dW difference can cross a bf16 rounding boundary downstream
        np.testing.assert_allclose(g, e, rtol=2e-3, atol=2e-4, err_msg=k)
