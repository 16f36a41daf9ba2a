#!/usr/bin/env python3
"""Per-kernel microbenchmarks through the C-ABI (HIP events on the launch stream).

  python tools/microbench.py emb      embedding gather / fused backward at Terabyte + Kaggle shapes
  python tools/microbench.py gemm     Linear fwd / bwd at the DLRM layer shapes
Algorithmic bytes follow SURVEY.md 8(d): fwd B*(L*(8+4D)+4D), fused bwd B*(8L+4D+L*2*4D) per table.
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlrm_flexflow_amd import capi
import _lab

DEV = "cuda:0"
PEAK_HBM = 8.0e12
PEAK_F32 = 157.3e12


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def emb(hip):
    cases = [("terabyte-4tables", 4, 32768, 128, [39884406, 39043, 38532951, 2953546]),
             ("terabyte-small-tables", 6, 32768, 128, [3, 63, 10, 155, 4, 36]),
             ("terabyte-26", 26, 32768, 128, [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546,
                                               403346, 10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984,
                                               585935, 12972, 108, 36]),
             ("kaggle-26", 26, 2048, 16, [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
                                          27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]),
             # what one of 2 / 4 / 8 ranks sees at the Kaggle shape (weak scaling: 2048 samples per rank, tables dealt round-robin)
             ("kaggle-rank-of-2", 13, 4096, 16, [1460, 10131227, 305, 12517, 3, 5683, 3194, 14992, 10, 2173, 7046547, 15, 105]),
             ("kaggle-rank-of-4", 7, 8192, 16, [1460, 305, 3, 3194, 10, 7046547, 105]),
             ("kaggle-rank-of-8", 4, 16384, 16, [1460, 3, 10, 105]),
             ("kaggle-rank2-of-8", 3, 16384, 16, [10131227, 5683, 2173]),
             ("giant-colshard", 1, 32768, 32, [200_000_000])]
    only = [a for a in sys.argv[2:]]
    for name, T, B, D, rows in cases:
        if only and name not in only:
            continue
        W, I = [], []
        for t, R in enumerate(rows):
            w = torch.empty(R, D, device=DEV)
            hip.call("ffh_init_uniform", w, R * D, t, -0.01, 0.01, None)
            i = torch.empty(B, 1, dtype=torch.int64, device=DEV)
            hip.call("ffh_gen_indices", i, B, 100 + t, 0, R, None)
            W.append(w); I.append(i)
        ld = T * D
        Z = torch.empty(B, ld, device=DEV)
        G = torch.empty(B, ld, device=DEV)
        hip.call("ffh_gen_uniform01", G, G.numel(), 5, 0, None)
        ws = torch.empty(hip.lib.ffh_embedding_bwd_workspace_bytes(T, 1, D, B) + 256, dtype=torch.uint8, device=DEV)
        hip.set_workspace(ws, ws.numel())
        fa = hip.emb_tables([(I[t], W[t], Z[:, t * D:], rows[t], ld) for t in range(T)])
        ba = hip.emb_tables([(I[t], W[t], G[:, t * D:], rows[t], ld) for t in range(T)])
        tf = timeit(lambda: hip.check(hip.lib.ffh_embedding_fwd_multi(hip.ctx, fa, T, 1, D, B, capi.AGGR_MODE_SUM, None), "f"))
        tb = timeit(lambda: hip.check(hip.lib.ffh_embedding_bwd_sgd_fused_multi(hip.ctx, ba, T, 1, D, B, capi.AGGR_MODE_SUM, 1e-6, None), "b"))
        # the index-only phase alone (ffh_embedding_bwd_sort_multi): what a training step issues behind the gather, off its critical path;
        # the rest of the fused call is the apply phase (segmented sums + folds + SGD step, one launch above 2048 lookups per table)
        ts = timeit(lambda: hip.check(hip.lib.ffh_embedding_bwd_sort_multi(hip.ctx, ba, T, 1, D, B, None), "s"))
        bf = T * B * (8 + 4 * D + 4 * D)
        bb = T * B * (8 + 4 * D + 2 * 4 * D)
        ta = max(tb - ts, 1e-9)
        print(f"{name:24s} fwd {tf*1e6:9.1f} us  {bf/tf/1e9:8.1f} GB/s ({bf/tf/PEAK_HBM*100:5.1f}% of 8 TB/s) | "
              f"fused bwd+sgd {tb*1e6:9.1f} us  {bb/tb/1e9:8.1f} GB/s ({bb/tb/PEAK_HBM*100:5.1f}%) = sort {ts*1e6:6.1f} us + apply {ta*1e6:6.1f} us", flush=True)
        del W, I, Z, G, ws
        torch.cuda.empty_cache()


def gemm(hip):
    shapes = [(2048, 13, 512), (2048, 512, 256), (2048, 256, 64), (2048, 64, 16), (2048, 432, 512), (2048, 512, 256), (2048, 256, 1),
              (4096, 3456, 1024), (4096, 1024, 1024), (4096, 1024, 512), (4096, 512, 256), (8192, 479, 1024), (8192, 1024, 1024),
              (32768, 1024, 1024), (4096, 4096, 4096)]
    for B, IN, OUT in shapes:
        x = torch.randn(B, IN, device=DEV)
        w = torch.randn(OUT, IN, device=DEV) * 0.05
        b = torch.randn(OUT, device=DEV)
        y = torch.empty(B, OUT, device=DEV)
        dy = torch.randn(B, OUT, device=DEV)
        dx = torch.zeros(B, IN, device=DEV)
        dw = torch.zeros(OUT, IN, device=DEV)
        db = torch.zeros(OUT, device=DEV)
        tf = timeit(lambda: hip.call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, capi.AC_MODE_RELU, None))
        tb = timeit(lambda: hip.call("ffh_linear_bwd", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, capi.AC_MODE_RELU, None))
        tt = timeit(lambda: torch.nn.functional.linear(x, w, b))
        fl = 2.0 * B * IN * OUT
        print(f"linear B={B:6d} in={IN:5d} out={OUT:5d}  fwd {tf*1e6:8.1f} us {fl/tf/1e12:7.2f} TF/s ({fl/tf/PEAK_F32*100:5.1f}%) | "
              f"bwd {tb*1e6:8.1f} us {2*fl/tb/1e12:7.2f} TF/s | torch(hipBLASLt) fwd {tt*1e6:8.1f} us {fl/tt/1e12:7.2f} TF/s", flush=True)


if __name__ == "__main__":
    hip = _lab.load_hip(0)
    print(hip.device_info().name.decode(), hip.device_info().compute_units, "CUs")
    what = sys.argv[1:2] or ["emb", "gemm"]
    if "emb" in what:
        emb(hip)
    if "gemm" in what:
        gemm(hip)
