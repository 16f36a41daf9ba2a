#!/usr/bin/env python3
"""3000 steps of the Kaggle-shape model on the GPU: the loss must keep falling and the weights stay finite."""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dlrm_flexflow_amd import ffmodel
w = bench.workload(sys.argv[1] if len(sys.argv) > 1 else "kaggle", bench.DEFAULT_BATCH[sys.argv[1] if len(sys.argv) > 1 else "kaggle"])
app = ffmodel.DLRM(bench.flags_of(w, ["--device", "0"] + sys.argv[2:]))
app.warmup()
m = app.model
for k in range(6):
    m.reset_metrics()
    app.train_steps(500, trace=False)
    pm = m.perf_metrics()
    print(k, "mse", 2.0 * pm.mse_loss / max(pm.train_all, 1), flush=True)
w0 = m.parameter(0, 0).get_weights()
print("finite", np.isfinite(w0).all(), float(np.abs(w0).max()))
app.close()
