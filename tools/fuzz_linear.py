#!/usr/bin/env python3
"""Random-shape check of ffh_linear_fwd / ffh_linear_bwd_ex on the GPU against the CPU oracle (test infrastructure):
shapes drawn so that every kernel family of linear.hip / linear_sk.hip is hit (LDS-DMA single / paired launches, register-staged
tiles, skinny outputs, the persistent stream-K kernels), ragged sizes, strides, all flag combinations; tolerance 1e-5 of the term
mass (FUZZ_TOL overrides).  Usage: tools/fuzz_linear.py [cases] [seed] [math_mode]
math_mode 1 (tensor-op bf16 operands): both sides run in that mode; 2 (fp32-accurate bf16x3 split): the GPU runs in it, the
oracle computes in fp32 -- same tolerance either way."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from dlrm_flexflow_amd import capi
from oracle import oracle
oracle.build()
hip = capi.load_hip(0)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
math_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
TOL = float(os.environ.get("FUZZ_TOL", "1e-5"))     # north_star: 1e-5 of the term mass
assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, math_mode) == 0
assert oracle.lib().lib.ffh_ctx_set_math_mode(oracle.lib().ctx, math_mode) == 0
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def close(got, exp, mass, what):
    tol = TOL * mass + 1e-6
    bad = np.abs(got.astype(np.float64) - exp) > tol
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} off, worst {np.abs(got - exp).max():.3e} vs tol {tol.max():.3e}"
worst = 0
for case in range(ncases):
    kind = rng.integers(0, 5)
    if kind == 0:      # LDS-DMA territory, multiples of 4
        B = int(rng.integers(64, 3000)); IN = 4 * int(rng.integers(16, 200)); OUT = 4 * int(rng.integers(16, 200))
    elif kind == 1:    # anything goes (unaligned -> register-staged kernels)
        B = int(rng.integers(1, 600)); IN = int(rng.integers(1, 300)); OUT = int(rng.integers(5, 300))
    elif kind == 2:    # skinny outputs
        B = int(rng.integers(1, 5000)); OUT = int(rng.integers(1, 17)); IN = 4 * int(rng.integers(1, 256 if OUT <= 4 else 65))
    elif kind == 3:    # wide and deep
        B = int(rng.integers(512, 4097)); IN = 4 * int(rng.integers(64, 300)); OUT = 4 * int(rng.integers(64, 300))
    else:              # whole 128 x 128 x 64 tiles, enough of them: the persistent one-workgroup-per-CU kernels (linear_sk.hip)
        B = 128 * int(rng.choice([32, 64, 96, 128])); IN = 128 * int(rng.integers(1, 9)); OUT = 128 * int(rng.integers(1, 9))
    if math_mode and kind != 2 and rng.integers(0, 4):   # mostly layers the bf16-pipe modes serve (both dims >= 128), any alignment
        IN = max(IN, 128) + int(rng.integers(0, 4)) * (kind == 1); OUT = max(OUT, 128) + int(rng.integers(0, 4)) * (kind == 1)
    act = int(rng.choice([capi.AC_MODE_NONE, capi.AC_MODE_RELU, capi.AC_MODE_SIGMOID]))
    padx, pady = 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3))
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32) if rng.integers(0, 2) else rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    xt = torch.zeros(B, IN + padx, device="cuda"); xt[:, :IN] = dev(x)
    yt = torch.full((B, OUT + pady), 7.0, device="cuda")
    hip.call("ffh_linear_fwd", xt, IN + padx, yt, OUT + pady, dev(w), dev(b), IN, OUT, B, act, None)
    y_e = oracle.linear_fwd(x, w, b, act)
    close(yt[:, :OUT].cpu().numpy(), y_e, np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b), f"case {case} fwd {B}x{IN}->{OUT} act {act}")
    flags = 0
    if rng.integers(0, 2): flags |= capi.LINEAR_DX_OVERWRITE
    if rng.integers(0, 2): flags |= capi.LINEAR_DX_MASK_BY_X
    if rng.integers(0, 2) and act != capi.AC_MODE_SIGMOID: flags |= capi.LINEAR_DY_PREMASKED
    mode = int(rng.integers(0, 3))    # 0 one call, 1 one call with a second stream, 2 split calls
    dx0 = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    dxt = torch.zeros(B, IN + padx, device="cuda"); dxt[:, :IN] = dev(dx0)
    dyt = torch.zeros(B, OUT + pady, device="cuda"); dyt[:, :OUT] = dev(gy)
    yd = torch.zeros(B, OUT + pady, device="cuda"); yd[:, :OUT] = dev(y_e)
    dw, db = torch.zeros(OUT, IN, device="cuda"), torch.zeros(OUT, device="cuda")
    args = (xt, IN + padx, dxt, IN + padx, yd, OUT + pady, dyt, OUT + pady, dev(w), dw, db, IN, OUT, B, act)
    s2 = torch.cuda.Stream()
    if mode == 2:
        hip.call("ffh_linear_bwd_ex", *args, flags | capi.LINEAR_ONLY_DX, None, None)
        hip.call("ffh_linear_bwd_ex", *args, flags | capi.LINEAR_ONLY_DW, None, None)
    else:
        hip.call("ffh_linear_bwd_ex", *args, flags, None, s2.cuda_stream if mode == 1 else None)
    torch.cuda.synchronize()
    dx_e, dw_e, db_e, dy_e = oracle.linear_bwd_ex(x, y_e, gy, w, act, flags | capi.LINEAR_DX_OVERWRITE, dx0=None)
    if not (flags & capi.LINEAR_DX_OVERWRITE): dx_e = dx_e + dx0
    a = np.abs(dy_e).astype(np.float64)
    what = f"case {case} bwd {B}x{IN}->{OUT} act {act} flags {flags} mode {mode}"
    np.testing.assert_allclose(dyt[:, :OUT].cpu().numpy(), dy_e, rtol=1e-6, atol=1e-7, err_msg=what)
    close(dxt[:, :IN].cpu().numpy(), dx_e, a @ np.abs(w).astype(np.float64) + (0 if flags & 1 else np.abs(dx0)), what + " dx")
    close(dw.cpu().numpy(), dw_e, a.T @ np.abs(x).astype(np.float64), what + " dw")
    close(db.cpu().numpy(), db_e, a.sum(0), what + " db")
    assert (dxt[:, IN:].cpu().numpy() == 0).all() and (yt[:, OUT:].cpu().numpy() == 7.0).all(), what + ": wrote into the padding"
print(f"fuzz_linear: {ncases} random cases agree with the oracle" + (f" (math mode {math_mode})" if math_mode else ""))
