#!/usr/bin/env python3
"""The chain launches of narrow Linear layers (ffh_mlp_chain_fwd / _bwd, csrc/mlp_chain.hip) next to the per-layer calls they
replace, HIP events on one stream:   python tools/chain_bench.py [batch ...]
Chains: the Terabyte / MLPerf bottom MLP 13-512-256-128, the Kaggle bottom MLP 13-512-256-64-16, the Kaggle top MLP 432-512-256(-1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab

DEV = "cuda:0"
RELU, SIG = capi.AC_MODE_RELU, capi.AC_MODE_SIGMOID


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    hip = _lab.load_hip(0)
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 8192, 32768]
    single = "--single" in sys.argv       # every layer of the bottom MLP as a chain of its own (where does the chain's time go?)
    only = [a[7:] for a in sys.argv[1:] if a.startswith("--only=")]
    chains = [("bottom 13-512-256-128", (13, 512, 256, 128), (RELU, RELU, RELU), False),
              ("kaggle bottom 13-512-256-64-16", (13, 512, 256, 64, 16), (RELU,) * 4, False),
              ("kaggle top 432-512-256 (+ dx)", (432, 512, 256), (RELU, RELU), True),
              ("kaggle top fwd 432-512-256-1", (432, 512, 256, 1), (RELU, RELU, SIG), True)]
    if single:
        chains = [("13-512", (13, 512), (RELU,), False), ("512-256", (512, 256), (RELU,), True), ("256-128", (256, 128), (RELU,), True),
                  ("512-256-128", (512, 256, 128), (RELU, RELU), True), ("432-512", (432, 512), (RELU,), True), ("64-16", (64, 16), (RELU,), True)]
    if only:
        chains = [c for c in chains if any(o.replace("_", " ") in c[0] for o in only)]
    s2 = torch.cuda.Stream()
    for B in batches:
        for name, widths, acts, want_dx in chains:
            n = len(widths) - 1
            x = torch.rand(B, widths[0], device=DEV)
            w = [torch.randn(widths[l + 1], widths[l], device=DEV) * (2.0 / widths[l]) ** 0.5 for l in range(n)]
            b = [torch.zeros(widths[l + 1], device=DEV) for l in range(n)]
            y = [torch.empty(B, widths[l + 1], device=DEV) for l in range(n)]
            dy = [torch.rand(B, widths[l + 1], device=DEV) for l in range(n)]
            dw = [torch.zeros_like(t) for t in w]
            db = [torch.zeros_like(t) for t in b]
            dx = torch.zeros(B, widths[0], device=DEV) if want_dx else None
            layers = hip.chain_layers([dict(w=w[l], bias=b[l], y=y[l], dy=dy[l], dw=dw[l], db=db[l], in_dim=widths[l], out_dim=widths[l + 1],
                                            activation=acts[l]) for l in range(n)])
            flops = 2.0 * B * sum(widths[l] * widths[l + 1] for l in range(n))

            def chain_fwd():
                hip.check(hip.lib.ffh_mlp_chain_fwd(hip.ctx, capi.ptr(x), widths[0], layers, n, B, None), "chain fwd")

            def layer_fwd():
                cur, ld = x, widths[0]
                for l in range(n):
                    hip.call("ffh_linear_fwd", cur, ld, y[l], widths[l + 1], w[l], b[l], widths[l], widths[l + 1], B, acts[l], None)
                    cur, ld = y[l], widths[l + 1]

            t_c, t_l = timeit(chain_fwd), timeit(layer_fwd)
            print(f"B={B:6d} {name:34s} forward : chain {t_c:7.1f} us ({flops / t_c * 1e-6:6.1f} TFLOP/s)   per-layer {t_l:7.1f} us")
            if acts[-1] == SIG:
                continue
            flags = capi.LINEAR_DY_PREMASKED | capi.LINEAR_DX_OVERWRITE

            def chain_bwd():
                hip.check(hip.lib.ffh_mlp_chain_bwd(hip.ctx, capi.ptr(x), widths[0], capi.ptr(dx), widths[0], layers, n, B, flags, None), "chain bwd")

            def layer_bwd():
                for l in range(n - 1, -1, -1):
                    xin, ldx = (x, widths[0]) if l == 0 else (y[l - 1], widths[l])
                    dxl = dx if l == 0 else dy[l - 1]
                    f = capi.LINEAR_DY_PREMASKED | capi.LINEAR_DX_OVERWRITE | (capi.LINEAR_DX_MASK_BY_X if l > 0 else 0)
                    hip.call("ffh_linear_bwd_ex", xin, ldx, dxl, ldx, y[l], widths[l + 1], dy[l], widths[l + 1], w[l], dw[l], db[l], widths[l], widths[l + 1], B,
                             acts[l], f, None, s2.cuda_stream)
                torch.cuda.current_stream().wait_stream(s2)

            chain_fwd(); torch.cuda.synchronize()
            t_c, t_l = timeit(chain_bwd), timeit(layer_bwd)
            nb = 2 if want_dx else (2.0 - widths[0] * widths[1] / sum(widths[l] * widths[l + 1] for l in range(n)))
            print(f"B={B:6d} {name:34s} backward: chain {t_c:7.1f} us ({nb * flops / t_c * 1e-6:6.1f} TFLOP/s)   per-layer {t_l:7.1f} us   route {hip.lib.ffh_linear_last_route(hip.ctx).decode()}")


if __name__ == "__main__":
    main()
