#!/usr/bin/env python3
"""Stand-alone timing of ffh_linear_pair_bwd against the two calls it replaces (HIP events on torch's stream).
A ctypes call with 26 arguments costs the host ~6.7 us, so figures near that are the interpreter, not the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab

B, INL, INU, OUTU = 2048, 256, 64, 16
hip = _lab.load_hip(0)
xl = torch.randn(B, INL, device="cuda"); xu = torch.randn(B, INU, device="cuda"); yu = torch.randn(B, OUTU, device="cuda")
gu = torch.randn(B, OUTU, device="cuda"); wu = torch.randn(OUTU, INU, device="cuda"); wl = torch.randn(INU, INL, device="cuda")
dxl = torch.zeros(B, INL, device="cuda"); dyl = torch.zeros(B, INU, device="cuda")
dwu = torch.zeros(OUTU, INU, device="cuda"); dbu = torch.zeros(OUTU, device="cuda")
dwl = torch.zeros(INU, INL, device="cuda"); dbl = torch.zeros(INU, device="cuda")
s = torch.cuda.current_stream().cuda_stream
R, N, P, OW, MX, ODX = capi.AC_MODE_RELU, capi.AC_MODE_NONE, capi.LINEAR_DY_PREMASKED, capi.LINEAR_DX_OVERWRITE, capi.LINEAR_DX_MASK_BY_X, capi.LINEAR_ONLY_DX


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pair():
    hip.call("ffh_linear_pair_bwd", xu, INU, yu, OUTU, gu, OUTU, wu, dwu, dbu, INU, OUTU, R, P, xl, INL, dxl, INL, dyl, INU, wl, INL, R, OW | MX, B, s)


def two():
    hip.call("ffh_linear_bwd_ex", xu, INU, dyl, INU, yu, OUTU, gu, OUTU, wu, dwu, dbu, INU, OUTU, B, R, P | OW | MX, s, None)
    hip.call("ffh_linear_bwd_ex", xl, INL, dxl, INL, xu, INU, dyl, INU, wl, dwl, dbl, INL, INU, B, R, OW | MX | ODX | P, s, None)


print(f"pair launch {timeit(pair):.1f} us   upper backward + lower dX as two calls {timeit(two):.1f} us")
