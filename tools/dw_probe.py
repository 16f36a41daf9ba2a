import sys, os
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
for B, IN, OUT in ((2048, 432, 512), (2048, 512, 256), (4096, 1024, 1024)):
    x = torch.randn(B, IN, device="cuda"); w = torch.randn(OUT, IN, device="cuda") * 0.05
    y = torch.rand(B, OUT, device="cuda"); dy = torch.randn(B, OUT, device="cuda")
    dw = torch.zeros(OUT, IN, device="cuda"); db = torch.zeros(OUT, device="cuda")
    for name, act, dbp in (("none,no-bias", capi.AC_MODE_NONE, None), ("none,bias", capi.AC_MODE_NONE, db), ("relu,bias", capi.AC_MODE_RELU, db)):
        t = timeit(lambda: hip.call("ffh_linear_bwd_ex", x, IN, None, IN, y, OUT, dy, OUT, w, dw, dbp, IN, OUT, B, act, 2, None, None))
        print(f"dW {B}x{OUT}x{IN} {name:14s} {t:7.1f} us  {2.0*B*IN*OUT/t/1e6:6.1f} TF/s", flush=True)
