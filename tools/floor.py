import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dlrm_flexflow_amd import ffmodel
w = bench.workload("kaggle", None, 1)
app = ffmodel.DLRM(bench.flags_of(w))
app.warmup(); app.train_steps(5, True); app.model.sync()
print("trivial kernel, back-to-back on one stream: %.2f us" % (app.time_kernel(3, 500) * 1e3))
print("embedding fwd:  %.2f us" % (app.time_kernel(0, 500) * 1e3))
print("graph step:     %.2f us" % (app.time_kernel(2, 200) * 1e3))
print("eager step:     %.2f us" % (app.time_kernel(4, 200) * 1e3))
