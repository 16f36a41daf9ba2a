#!/usr/bin/env python3
"""Random-shape check of ffh_linear_fwd / ffh_linear_bwd_ex on the GPU against the CPU oracle (test infrastructure):
shapes drawn so that every kernel family of linear.hip / linear_sk.hip is hit (LDS-DMA single / paired launches, register-staged
tiles, skinny outputs, the persistent stream-K kernels), ragged sizes, strides, all flag combinations; tolerance 1e-5 of the term
mass (FUZZ_TOL overrides).  Usage: tools/fuzz_linear.py [cases] [seed] [math_mode]
math_mode 1 (tensor-op bf16 operands): both sides run in that mode; 2 / 3 (fp32-accurate bf16x3 split; 3 = on every shape the split
kernels accept, whatever its size -- the form to fuzz: 2 leaves small GEMMs to the fp32 kernels): the GPU runs in it, the oracle computes
in fp32 -- same tolerance either way.  In the split modes half of the cases register THREE-PLANE IMAGES for every operand and result
(ffh_ctx_bf16x3_mirror_set; round 6): the LDS-DMA kernel where the shape allows it (one case in six is drawn big enough), the
split-in-kernel form + a conversion pass elsewhere -- and the images of y and dx must then be the images of exactly what was stored."""
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
def image_of(flat):      # numpy restatement of the I32 image (tests/test_gpu_round6.py): uint16 [groups][3][32]
    def bf16_bits(a):
        u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)
    f32 = lambda b: (b.astype(np.uint32) << 16).view(np.float32)
    a = np.ascontiguousarray(flat, np.float32)
    b1 = bf16_bits(a); r1 = a - f32(b1); b2 = bf16_bits(r1); r2 = r1 - f32(b2); b3 = bf16_bits(r2)
    return np.stack([b1.reshape(-1, 32), b2.reshape(-1, 32), b3.reshape(-1, 32)], axis=1)
worst = 0
n_images = n_dma = 0
for case in range(ncases):
    kind = rng.integers(0, 6 if math_mode >= 2 else 5)
    if kind == 0:      # LDS-DMA territory, multiples of 4
        B = int(rng.integers(64, 3000)); IN = 4 * int(rng.integers(16, 200)); OUT = 4 * int(rng.integers(16, 200))
    elif kind == 1:    # anything goes (unaligned -> register-staged kernels)
        B = int(rng.integers(1, 600)); IN = int(rng.integers(1, 300)); OUT = int(rng.integers(5, 300))
    elif kind == 2:    # skinny outputs
        B = int(rng.integers(1, 5000)); OUT = int(rng.integers(1, 17)); IN = 4 * int(rng.integers(1, 256 if OUT <= 4 else 65))
    elif kind == 3:    # wide and deep
        B = int(rng.integers(512, 4097)); IN = 4 * int(rng.integers(64, 300)); OUT = 4 * int(rng.integers(64, 300))
    elif kind == 4:    # whole 128 x 128 x 64 tiles, enough of them: the persistent one-workgroup-per-CU kernels (linear_sk.hip)
        B = 128 * int(rng.choice([32, 64, 96, 128])); IN = 128 * int(rng.integers(1, 9)); OUT = 128 * int(rng.integers(1, 9))
    else:              # split modes: enough 256 x 256 tiles for the LDS-DMA kernel (>= 3 per 4 CUs) forward and / or data gradient, ragged rows
        B = 12288 + 32 * int(rng.integers(0, 9)); IN = 32 * int(rng.choice([16, 24, 32, 33])); OUT = 32 * int(rng.choice([24, 32, 36]))
    if math_mode and kind != 2 and rng.integers(0, 4):   # mostly layers the bf16-pipe modes serve (both dims >= 128), any alignment
        IN = max(IN, 128) + int(rng.integers(0, 4)) * (kind == 1); OUT = max(OUT, 128) + int(rng.integers(0, 4)) * (kind == 1)
    act = int(rng.choice([capi.AC_MODE_NONE, capi.AC_MODE_RELU, capi.AC_MODE_SIGMOID]))
    padx, pady = 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3))
    images = math_mode >= 2 and bool(rng.integers(0, 2))
    if kind == 5: padx, pady, images = 32 * int(rng.integers(0, 2)), 32 * int(rng.integers(0, 2)), True
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32) if rng.integers(0, 2) else rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    xt = torch.zeros(B, IN + padx, device="cuda"); xt[:, :IN] = dev(x)
    yt = torch.full((B, OUT + pady), 7.0, device="cuda")
    wt = dev(w)
    regs = {}
    def register(name, t, convert):
        if not images or t.data_ptr() % 128 or t.numel() % 32: return
        img = torch.zeros(t.numel() // 32 * 96, dtype=torch.int16, device="cuda")
        assert hip.lib.ffh_ctx_bf16x3_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, img.data_ptr()) == 0
        regs[name] = (t, img)
        if convert: hip.call("ffh_convert_f32_to_bf16x3", t, 1, t.numel(), t.numel(), None)
    register("x", xt, True); register("w", wt, True); register("y", yt, True)      # (y: its padding columns hold 7.0 and must keep their image)
    hip.call("ffh_linear_fwd", xt, IN + padx, yt, OUT + pady, wt, dev(b), IN, OUT, B, act, None)
    r_fwd = hip.lib.ffh_linear_last_route(hip.ctx).decode()
    if "x3_dma" in r_fwd: n_dma += 1
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
    register("yd", yd, True); register("dy", dyt, True); register("dx", dxt, True)
    args = (xt, IN + padx, dxt, IN + padx, yd, OUT + pady, dyt, OUT + pady, wt, dw, db, IN, OUT, B, act)
    s2 = torch.cuda.Stream()
    if mode == 2:
        hip.call("ffh_linear_bwd_ex", *args, flags | capi.LINEAR_ONLY_DX, None, None)
        hip.call("ffh_linear_bwd_ex", *args, flags | capi.LINEAR_ONLY_DW, None, None)
    else:
        hip.call("ffh_linear_bwd_ex", *args, flags, None, s2.cuda_stream if mode == 1 else None)
    r_bwd = hip.lib.ffh_linear_last_route(hip.ctx).decode()
    if "x3_dma" in r_bwd: n_dma += 1
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
    if regs:
        n_images += 1
        # the producers' contract (include/ff_hip.h): a GEMM that ran on the mode's kernels left the image of exactly what it stored
        on_pipe = {"y": any(k in r_fwd for k in ("bf16x3", "x3_dma")), "dx": any(("dx" in t.split("|")[0]) and ("bf16x3" in t or "x3_dma" in t) for t in r_bwd.split(";"))}
        for name in ("y", "dx"):
            if name in regs and on_pipe[name]:
                t, img = regs[name]
                assert np.array_equal(img.cpu().numpy().view(np.uint16).reshape(-1, 3, 32), image_of(t.cpu().numpy().ravel())), what + f": image of {name}"
        for t, _ in regs.values():
            assert hip.lib.ffh_ctx_bf16x3_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, None) == 0
print(f"fuzz_linear: {ncases} random cases agree with the oracle" + (f" (math mode {math_mode})" if math_mode else "") + (f"; {n_images} with images, {n_dma} calls on the LDS-DMA kernel" if math_mode >= 2 else ""))
