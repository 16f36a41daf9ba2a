#!/usr/bin/env python3
"""Tensor-op mode, one big layer: forward / dX / dW timed with no twins, operand twins only, output twin only, all twins."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab
hip = _lab.load_hip(0)
assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 1) == 0
B, IN, OUT = (int(v) for v in (sys.argv[1].split("x") if len(sys.argv) > 1 else (32768, 3456, 1024)))
dev = "cuda"
x = torch.relu(torch.randn(B, IN, device=dev)); w = torch.randn(OUT, IN, device=dev) * 0.05; b = torch.randn(OUT, device=dev)
y = torch.empty(B, OUT, device=dev); dy = torch.randn(B, OUT, device=dev); dx = torch.zeros(B, IN, device=dev)
dw = torch.zeros(OUT, IN, device=dev); db = torch.zeros(OUT, device=dev)
tw = {n: torch.zeros(t.shape, dtype=torch.bfloat16, device=dev) for n, t in (("x", x), ("w", w), ("y", y), ("dy", dy), ("dx", dx))}
for n, t in (("x", x), ("w", w), ("dy", dy)):
    hip.call("ffh_convert_f32_to_bf16", tw[n], t, t.numel(), None)
T = {"x": x, "w": w, "y": y, "dy": dy, "dx": dx}
def timeit(fn, iters=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
fl = 2.0 * B * IN * OUT
for label, regs in (("no twins", ()), ("operand twins", ("x", "w", "dy")), ("output twins only", ("y", "dx")), ("all twins", ("x", "w", "dy", "y", "dx"))):
    for n in regs:
        assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, T[n].data_ptr(), T[n].numel() * 4, tw[n].data_ptr()) == 0
    tf = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_RELU, None))
    rf = hip.lib.ffh_linear_last_route(hip.ctx).decode()
    tx = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, 4 | 1 | 8, None, None))
    rx = hip.lib.ffh_linear_last_route(hip.ctx).decode()
    tw_ = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, 2 | 8, None, None))
    rw = hip.lib.ffh_linear_last_route(hip.ctx).decode()
    print(f"{label:20s} fwd {tf:8.1f} us {fl/tf/1e6:7.1f} TF | dX {tx:8.1f} us {fl/tx/1e6:7.1f} TF | dW {tw_:8.1f} us {fl/tw_/1e6:7.1f} TF   [{rf.split('|')[1]} / {rx.split('|')[1]} / {rw.split('|')[1] if '|' in rw else rw}]", flush=True)
    for n in regs:
        assert hip.lib.ffh_ctx_bf16_mirror_set(hip.ctx, T[n].data_ptr(), T[n].numel() * 4, None) == 0
