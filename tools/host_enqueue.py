#!/usr/bin/env python3
"""Host wall time to enqueue one eager Kaggle-shape step vs the time the step takes end to end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dlrm_flexflow_amd import ffmodel
w = bench.workload("kaggle", None, 1)
app = ffmodel.DLRM(bench.flags_of(w, ["--device", "0"] + sys.argv[1:]))
app.warmup()
for iters in (20, 100):
    print(f"iters {iters}: enqueue {app.time_kernel(5, iters) * 1e3:.1f} us/step   end-to-end {app.time_kernel(4, iters) * 1e3:.1f} us/step")
app.close()
