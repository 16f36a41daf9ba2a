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
import time
app.model.sync()
t0 = time.perf_counter(); app.train_steps(200, False); t1 = time.perf_counter(); app.model.sync(); t2 = time.perf_counter()
print("eager: host enqueue %.1f us/step, until device done %.1f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
t0 = time.perf_counter(); app.train_steps(200, True); t1 = time.perf_counter(); app.model.sync(); t2 = time.perf_counter()
print("graph: host enqueue %.1f us/step, until device done %.1f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
