#!/usr/bin/env python3
"""Split mode (FFH_MATH_FP32_SPLIT_BF16X3_ALL), one Linear layer through the C-ABI: forward / dX / dW timed with the operands split inside the
kernels, with three-plane images of the operands only, and with images of everything (outputs' images written by the epilogues).
usage: x3_image_probe.py 32768x3456x1024 [more shapes]        (warm clocks: every leg runs 150 ms before it is timed)
A shape with the suffix "n" (32768x3456x1024n) times the data gradient WITHOUT the relu'-by-x mask -- the form of a layer whose input is not a
ReLU output (the first top layer behind the Concat: no second 4-byte read per dX element)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab
hip = _lab.load_hip(0)
dev = "cuda"


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ms = 0.0
    while ms < 150.0:
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for shape in (sys.argv[1:] or ["32768x3456x1024"]):
    nomask = shape.endswith("n")
    B, IN, OUT = (int(v) for v in shape.rstrip("n").split("x"))
    DXF = 4 | 1 | 8 | (0 if nomask else 16)
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 3) == 0
    x = torch.relu(torch.randn(B, IN, device=dev)); w = torch.randn(OUT, IN, device=dev) * 0.05; b = torch.randn(OUT, device=dev)
    y = torch.empty(B, OUT, device=dev); dy = torch.randn(B, OUT, device=dev); dx = torch.zeros(B, IN, device=dev)
    dw = torch.zeros(OUT, IN, device=dev); db = torch.zeros(OUT, device=dev)
    T = {"x": x, "w": w, "y": y, "dy": dy, "dx": dx}
    img = {n: torch.zeros((t.numel() + 31) // 32 * 96, dtype=torch.int16, device=dev) for n, t in T.items()}
    fl = 2.0 * B * IN * OUT
    print(f"layer {IN} -> {OUT} at batch {B}{' (dX without the relu mask)' if nomask else ''}: TFLOP/s are fp32-equivalent (2 * B * in * out), roofline 416.7 = bf16 MFMA peak / 6")
    for label, regs in (("split in the kernels", ()), ("operand images", ("x", "w", "dy")), ("all images", ("x", "w", "dy", "y", "dx"))):
        for n in regs:
            assert hip.lib.ffh_ctx_bf16x3_mirror_set(hip.ctx, T[n].data_ptr(), T[n].numel() * 4, img[n].data_ptr()) == 0
        for n in regs:
            if n in ("x", "w", "dy"): hip.call("ffh_convert_f32_to_bf16x3", T[n], 1, T[n].numel(), T[n].numel(), None)
        tf = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_RELU, None))
        rf = hip.lib.ffh_linear_last_route(hip.ctx).decode()
        tx = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, DXF, None, None))
        rx = hip.lib.ffh_linear_last_route(hip.ctx).decode()
        tw_ = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, 2 | 8, None, None))
        rw = hip.lib.ffh_linear_last_route(hip.ctx).decode()
        tag = lambda r: r.split("|")[1] if "|" in r else r
        print(f"  {label:22s} fwd {tf:8.1f} us {fl/tf/1e6:6.1f} TF ({fl/tf/1e6/416.7:.3f}) | dX {tx:8.1f} us {fl/tx/1e6:6.1f} TF ({fl/tx/1e6/416.7:.3f}) | dW {tw_:8.1f} us {fl/tw_/1e6:6.1f} TF ({fl/tw_/1e6/416.7:.3f})   [{tag(rf)} / {tag(rx)} / {tag(rw)}]", flush=True)
        for n in regs:
            assert hip.lib.ffh_ctx_bf16x3_mirror_set(hip.ctx, T[n].data_ptr(), T[n].numel() * 4, None) == 0
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 0) == 0
    tf = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_RELU, None))
    tx = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, DXF, None, None))
    tw_ = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, 2 | 8, None, None))
    print(f"  {'exact fp32 MFMA':22s} fwd {tf:8.1f} us {fl/tf/1e6:6.1f} TF         | dX {tx:8.1f} us {fl/tx/1e6:6.1f} TF         | dW {tw_:8.1f} us {fl/tw_/1e6:6.1f} TF", flush=True)
    del x, w, y, dy, dx, dw, img
