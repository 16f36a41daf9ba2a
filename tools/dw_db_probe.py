#!/usr/bin/env python3
"""The weight-gradient GEMM alone, with and without the bias gradient riding on it (db = NULL), per layer shape; and the small layers of
the bottom MLP alone.  Shapes: BxINxOUT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab
hip = _lab.load_hip(0)
def timeit(fn, iters=20):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 3456, 1024), (4096, 1024, 1024), (4096, 1024, 512), (32768, 3456, 1024), (32768, 1024, 1024)]
for B, IN, OUT in shapes:
    x = torch.relu(torch.randn(B, IN, device="cuda")); w = torch.randn(OUT, IN, device="cuda") * 0.05
    y = torch.relu(torch.randn(B, OUT, device="cuda")); dy = torch.randn(B, OUT, device="cuda"); dx = torch.zeros(B, IN, device="cuda")
    dw = torch.zeros(OUT, IN, device="cuda"); db = torch.zeros(OUT, device="cuda")
    fl = 2.0 * B * IN * OUT
    out = []
    for act, nm in ((capi.AC_MODE_NONE, "none"), (capi.AC_MODE_RELU, "relu")):
        for dbp, dn in ((db, "db"), (None, "no db")):
            for fl_extra, fn in ((0, ""), (capi.LINEAR_DY_PREMASKED, " premasked")):
                if act == capi.AC_MODE_NONE and fl_extra: continue
                t = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, dbp, IN, OUT, B, act, capi.LINEAR_ONLY_DW | fl_extra, None, None))
                route = hip.lib.ffh_linear_last_route(hip.ctx).decode()
                out.append(f"dW act {nm}{fn} {dn}: {t:7.1f} us {fl/t/1e6:6.1f} TF [{route}]")
        t = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, act, capi.LINEAR_ONLY_DX | capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DY_PREMASKED, None, None))
        out.append(f"dX act {nm} premasked overwrite: {t:7.1f} us {fl/t/1e6:6.1f} TF [{hip.lib.ffh_linear_last_route(hip.ctx).decode()}]")
        t = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, db, IN, OUT, B, act, None))
        out.append(f"fwd act {nm}: {t:7.1f} us {fl/t/1e6:6.1f} TF [{hip.lib.ffh_linear_last_route(hip.ctx).decode()}]")
    if os.environ.get("DW_CHECK"):
        dw.zero_(); dy2 = dy.clone()
        hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy2, OUT, w, dw, None, IN, OUT, B, capi.AC_MODE_NONE, capi.LINEAR_ONLY_DW, None, None)
        hip.call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy2, OUT, w, dw, None, IN, OUT, B, capi.AC_MODE_NONE, capi.LINEAR_ONLY_DW, None, None)   # accumulates: 2x
        torch.cuda.synchronize()
        ref = 2.0 * (dy.double().t() @ x.double()); mass = 2.0 * (dy.double().abs().t() @ x.double().abs()) + 1e-30
        out.append(f"check (two accumulating launches, no db) worst |err| / term mass = {((dw.double() - ref).abs() / mass).max().item():.3e}  [{hip.lib.ffh_linear_last_route(hip.ctx).decode()}]")
    print(f"{B} x {IN} -> {OUT}"); [print("   ", o) for o in out]; sys.stdout.flush()
