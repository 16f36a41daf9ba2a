#!/usr/bin/env python3
"""dW (split-K) timing of the Kaggle layer shapes over tile config x split factor (FFH_GEMM_CFG / FFH_GEMM_SPLIT)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from dlrm_flexflow_amd import capi
    hip = capi.load_hip(0)
    def timeit(fn, iters=100):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    out = []
    for B, IN, OUT in ((2048, 432, 512), (2048, 512, 256)):
        x = torch.randn(B, IN, device="cuda"); w = torch.randn(OUT, IN, device="cuda") * 0.05
        y = torch.rand(B, OUT, device="cuda"); dy = torch.randn(B, OUT, device="cuda"); dx = torch.zeros(B, IN, device="cuda")
        dw = torch.zeros(OUT, IN, device="cuda"); db = torch.zeros(OUT, device="cuda")
        t0 = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, None, IN, y, OUT, dy, OUT, w, dw, None, IN, OUT, B, capi.AC_MODE_NONE, 2, None, None))
        t1 = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, None, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, 2, None, None))
        t2 = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, 1 | 4, None, None))
        out.append(f"{OUT}x{IN}: dW plain {t0:5.1f} relu+bias {t1:5.1f} dX {t2:5.1f}")
    print(" | ".join(out), flush=True)
else:
    for cfg in ("1", "2"):
        for split in ("0", "2", "3", "4", "6", "8", "16"):
            print(f"cfg {cfg} split {split:>2}: ", end="", flush=True)
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, FFH_GEMM_CFG=cfg, FFH_GEMM_SPLIT=split))
