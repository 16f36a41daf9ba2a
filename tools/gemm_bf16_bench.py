#!/usr/bin/env python3
"""fwd / dX / dW of the big layers in fp32 mode and in tensor-op (bf16 operand) mode, same process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab
hip = _lab.load_hip(0)
def timeit(fn, iters=10):
    for _ in range(20): fn()      # the clock needs tens of milliseconds of load to settle
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(32768, 3456, 1024), (32768, 1024, 1024), (32768, 1024, 512), (32768, 512, 256), (4096, 3456, 1024), (8192, 479, 1024), (2048, 432, 512)]
for B, IN, OUT in shapes:
    x = torch.rand(B, IN, device="cuda"); w = torch.randn(OUT, IN, device="cuda") * 0.05; b = torch.randn(OUT, device="cuda")
    y = torch.empty(B, OUT, device="cuda"); dy = torch.randn(B, OUT, device="cuda"); dx = torch.zeros(B, IN, device="cuda")
    dw = torch.zeros(OUT, IN, device="cuda"); db = torch.zeros(OUT, device="cuda")
    fl = 2.0 * B * IN * OUT
    line = f"{B:6d} {IN:5d} {OUT:5d} "
    for mode in (0, 1, 2):
        hip.lib.ffh_ctx_set_math_mode(hip.ctx, mode)
        tf = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_NONE, None))
        tx = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_NONE, 4 | 1, None, None))
        tw = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_NONE, 2, None, None))
        line += f"| {('fp32', 'bf16', 'bf16x3')[mode]} fwd {tf:7.1f} us {fl/tf/1e6:6.1f} TF  dX {tx:7.1f} us {fl/tx/1e6:6.1f} TF  dW {tw:7.1f} us {fl/tw/1e6:6.1f} TF "
    hip.lib.ffh_ctx_set_math_mode(hip.ctx, 0)
    xb, wb = x.bfloat16(), w.bfloat16()
    t1 = timeit(lambda: torch.mm(xb, wb.t()))
    line += f"|| hipBLASLt bf16 fwd {fl/t1/1e6:6.1f} TF"
    print(line, flush=True)
