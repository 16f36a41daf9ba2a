#!/usr/bin/env python3
"""Times fwd / dW / dX of the DLRM layer shapes with each tile config forced (FFH_GEMM_CFG)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from dlrm_flexflow_amd import capi
    import _lab
    hip = _lab.load_hip(0)
    def timeit(fn, iters=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    shapes = [(2048, 13, 512), (2048, 512, 256), (2048, 256, 64), (2048, 64, 16), (2048, 432, 512), (2048, 256, 1),
              (4096, 3456, 1024), (4096, 1024, 1024), (4096, 1024, 512), (4096, 512, 256), (4096, 256, 1), (4096, 13, 512), (4096, 256, 128)]
    for B, IN, OUT in shapes:
        x = torch.randn(B, IN, device="cuda"); w = torch.randn(OUT, IN, device="cuda") * 0.05; b = torch.randn(OUT, device="cuda")
        y = torch.empty(B, OUT, device="cuda"); dy = torch.randn(B, OUT, device="cuda"); dx = torch.zeros(B, IN, device="cuda")
        dw = torch.zeros(OUT, IN, device="cuda"); db = torch.zeros(OUT, device="cuda")
        tf = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_RELU, None))
        tw = timeit(lambda: hip.call("ffh_linear_bwd", x, IN, None, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, None))
        tb = timeit(lambda: hip.call("ffh_linear_bwd", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, None))
        print(f"{B:5d} {IN:5d} {OUT:5d}  fwd {tf:7.1f}  dW {tw:7.1f}  dW+dX {tb:7.1f}", flush=True)
else:
    for cfg in ("-1", "0", "1", "2"):
        print("FFH_GEMM_CFG =", cfg, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, FFH_GEMM_CFG=cfg))
