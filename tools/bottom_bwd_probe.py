#!/usr/bin/env python3
"""Alone times of the bottom MLP's backward calls at a given batch (the calls the DLRM step makes, with their flags), and the route each takes.
  python tools/bottom_bwd_probe.py [--bf16] [batch ...]      (--bf16: tensor-op math mode, the weight gradient forked to a second stream as in the step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    hip = _lab.load_hip(0)
    bf16 = "--bf16" in sys.argv
    if bf16:
        hip.check(hip.lib.ffh_ctx_set_math_mode(hip.ctx, 1), "math mode")
    s2 = torch.cuda.Stream() if bf16 else None
    batches = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [32768]
    RELU, NONE = capi.AC_MODE_RELU, capi.AC_MODE_NONE
    PRE, ODW, OVR, MBX = capi.LINEAR_DY_PREMASKED, capi.LINEAR_ONLY_DW, capi.LINEAR_DX_OVERWRITE, capi.LINEAR_DX_MASK_BY_X
    for B in batches:
        for name, IN, OUT, act, flags in (("256->128 (relu live, dx masked by x)", 256, 128, RELU, OVR | MBX),
                                          ("512->256 (premasked, dx masked by x)", 512, 256, RELU, PRE | OVR | MBX),
                                          ("13->512 (premasked, dW only)", 13, 512, RELU, PRE | ODW)):
            x = torch.rand(B, IN, device=DEV); y = torch.rand(B, OUT, device=DEV); dy = torch.rand(B, OUT, device=DEV) - 0.5
            w = torch.rand(OUT, IN, device=DEV) - 0.5; dw = torch.zeros(OUT, IN, device=DEV); db = torch.zeros(OUT, device=DEV); dx = torch.zeros(B, IN, device=DEV)
            sdw = int(s2.cuda_stream) if s2 is not None else None
            f = lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, act, flags, None, sdw)
            t = timeit(f)
            route = hip.lib.ffh_linear_last_route(hip.ctx).decode()
            flop = 2.0 * B * IN * OUT * (1 if flags & ODW else 2)
            print(f"B={B:6d} {name:40s} {t:8.1f} us  {flop / t / 1e6:7.1f} TFLOP/s   {route}")


if __name__ == "__main__":
    main()
