#!/usr/bin/env python3
"""A few forward launches of the 3456->1024 layer at batch 32768 in one math mode (argv[1]: 0 fp32, 1 bf16, 2 split-bf16x3 from three-plane
images -- the LDS-DMA kernel --, 20 split-bf16x3 with the split inside the kernel): the program tools/pmc_x3.sh puts under rocprofv3 --pmc."""
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
hip.lib.ffh_ctx_set_math_mode(hip.ctx, 2 if mode == 20 else mode)
if mode == 2:
    img = {}
    for n, t in (("x", x), ("w", w), ("y", y)):
        img[n] = torch.zeros((t.numel() + 31) // 32 * 96, dtype=torch.int16, device="cuda")
        assert hip.lib.ffh_ctx_bf16x3_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, img[n].data_ptr()) == 0
    for t in (x, w):
        hip.call("ffh_convert_f32_to_bf16x3", t, 1, t.numel(), t.numel(), None)
for _ in range(4):
    hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_NONE, None)
torch.cuda.synchronize()
