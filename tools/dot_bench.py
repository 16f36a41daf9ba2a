#!/usr/bin/env python3
"""Stand-alone timing of the fused pairwise-dot interaction kernels (HIP events on torch's stream).
  python3 tools/dot_bench.py [B] [c] [d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = int(sys.argv[2]) if len(sys.argv) > 2 else 27
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
hip = _lab.load_hip(0)
P = c * (c - 1) // 2
z = torch.randn(B, c * d, device="cuda")
out = torch.empty(B, d + P, device="cuda")
g = torch.randn(B, d + P, device="cuda")
zg = torch.zeros(B, c * d, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=50):
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


fwd = timeit(lambda: hip.call("ffh_dot_interaction_fwd", z, c * d, out, d + P, B, c, d, s))
bwd = timeit(lambda: hip.call("ffh_dot_interaction_bwd", z, c * d, g, d + P, zg, c * d, B, c, d, 1, s))
fb = 4 * B * (c * d + d + P)
bb = 4 * B * (2 * c * d + d + P)
print(f"B={B} c={c} d={d}: fwd {fwd:.1f} us ({fb / fwd / 1e3:.0f} GB/s), bwd {bwd:.1f} us ({bb / bwd / 1e3:.0f} GB/s)")
