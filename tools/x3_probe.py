#!/usr/bin/env python3
"""A few forward launches of the 3456->1024 layer at batch 32768 in one math mode (argv[1]: 0 fp32, 1 bf16, 2 split-bf16x3):
the program tools/pmc_x3.sh puts under rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab
hip = _lab.load_hip(0)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, IN, OUT = 32768, 3456, 1024
x = torch.rand(B, IN, device="cuda"); w = torch.randn(OUT, IN, device="cuda") * 0.05; b = torch.randn(OUT, device="cuda")
y = torch.empty(B, OUT, device="cuda")
hip.lib.ffh_ctx_set_math_mode(hip.ctx, mode)
for _ in range(4):
    hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_NONE, None)
torch.cuda.synchronize()
