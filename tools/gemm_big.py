#!/usr/bin/env python3
"""Times fwd / dX / dW of the big Terabyte / MLPerf layers separately, per forced tile config (FFH_GEMM_CFG), next to
torch's sgemm (hipBLASLt) as a known-good reference on the same box."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from dlrm_flexflow_amd import capi
    import _lab
    hip = _lab.load_hip(0)
    def timeit(fn, iters=20):
        for _ in range(20): fn()          # the chip's clock needs tens of milliseconds under load to settle
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    shapes = [(32768, 3456, 1024), (32768, 1024, 1024), (32768, 1024, 512), (4096, 3456, 1024), (4096, 1024, 1024), (8192, 479, 1024)]
    if len(sys.argv) > 2: shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
    ref = os.environ.get("FFH_GEMM_CFG", "-1") == "-1"
    for B, IN, OUT in shapes:
        x = torch.randn(B, IN, device="cuda"); w = torch.randn(OUT, IN, device="cuda") * 0.05; b = torch.randn(OUT, device="cuda")
        if os.environ.get("GEMM_BIG_RELU_X"): x = torch.relu(x)      # operands as the step has them: activations behind a ReLU (half zeros)
        y = torch.empty(B, OUT, device="cuda"); dy = torch.randn(B, OUT, device="cuda"); dx = torch.zeros(B, IN, device="cuda")
        dw = torch.zeros(OUT, IN, device="cuda"); db = torch.zeros(OUT, device="cuda")
        fl = 2.0 * B * IN * OUT
        tf = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_NONE, None))
        tx = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_NONE, 4 | 1, None, None))
        tw = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_NONE, 2, None, None))
        line = f"{B:6d} {IN:5d} {OUT:5d}  fwd {tf:8.1f} us {fl/tf/1e6:6.1f} TF | dX {tx:8.1f} us {fl/tx/1e6:6.1f} TF | dW {tw:8.1f} us {fl/tw/1e6:6.1f} TF"
        if ref:
            t1 = timeit(lambda: torch.mm(x, w.t()))
            t2 = timeit(lambda: torch.mm(dy, w))
            t3 = timeit(lambda: torch.mm(dy.t(), x))
            line += f" || hipBLASLt fwd {fl/t1/1e6:6.1f} dX {fl/t2/1e6:6.1f} dW {fl/t3/1e6:6.1f} TF"
        print(line, flush=True)
else:
    for cfg in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("-1", "3", "4", "5")):
        print("FFH_GEMM_CFG =", cfg, flush=True)
        subprocess.run([sys.executable, __file__, "child", *sys.argv[2:]], env=dict(os.environ, FFH_GEMM_CFG=cfg))
