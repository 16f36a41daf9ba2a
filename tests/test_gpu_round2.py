"""GPU tests (-m gpu) added in round 2: stream-ordering hazards found by review, the configurations that had no
whole-step test (Terabyte / MLPerf widths, the 204.8 GB table), strategy files on the GPU, --profiling."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from dlrm_flexflow_amd import capi, ffmodel
import dlrm_helpers as H

pytestmark = pytest.mark.gpu
HIP = capi.HIP_LIB_PATH


def _tables_and_mlp(app):
    m = app.model
    out = {}
    for li in range(m.num_layers):
        for wi in range(m.layer_num_weights(li)):
            p = m.parameter(li, wi)
            if p.is_local:
                out[f"{m.layer_name(li)}/{wi}"] = p.get_weights()
    return out


@pytest.mark.parametrize("trace", [False, True])
def test_new_batch_every_step_is_ordered_behind_the_table_update(hip, tmp_path, trace):
    """Eager steps with --dataset copy a NEW batch into the id buffers every iteration while the fused table update of
    the step before (side stream) may still be sorting them: the loader orders its copies behind that update.  Shape
    chosen so that the update (4 tables x 16384 lookups, tiled radix path) outlasts the tiny bottom-MLP backward it runs
    beside.  Checked against the same host code on the CPU oracle, step by step: a half-overwritten id buffer or a
    one-step-stale gradient shows up in the LAST step's table delta (different batches touch different rows)."""
    rows = (200000, 50, 1000000, 7)
    B, nb = 16384, 3
    h5, _ = H.make_criteo_like_hdf5(str(tmp_path), n=B * nb, rows=rows)
    args = ["-b", str(B), "--arch-sparse-feature-size", "64", "--arch-embedding-size", "-".join(map(str, rows)),
            "--arch-mlp-bot", "13-16-64", "--arch-mlp-top", "320-32-1", "--dataset", h5]
    res = {}
    for name, backend in (("hip", HIP), ("cpu", H.oracle_backend())):
        app = ffmodel.DLRM(["--backend", backend] + args)
        app.warmup()
        app.train_steps(6, trace=trace and name == "hip")
        app.model.sync()
        before = _tables_and_mlp(app)
        app.train_steps(1, trace=trace and name == "hip")
        app.model.sync()
        after = _tables_and_mlp(app)
        res[name] = (before, after)
        app.close()
    for k in res["hip"][1]:
        g1, c1 = res["hip"][1][k], res["cpu"][1][k]
        np.testing.assert_allclose(g1, c1, rtol=2e-5, atol=2e-6, err_msg=k)
        if k.startswith("Embedding"):
            dg, dc = g1 - res["hip"][0][k], c1 - res["cpu"][0][k]
            scale = np.abs(dc).max()
            assert scale > 0
            # same rows touched by the same amounts (tolerance: fp32 differences of nearly equal numbers)
            assert np.abs(dg - dc).max() <= 2e-3 * scale, (k, float(np.abs(dg - dc).max()), float(scale))


def test_graph_replay_uses_this_steps_gradients_kaggle_shape(hip):
    """hipGraph replay vs eager at the Kaggle shape, where the "gradients ready" event of the side-stream table update
    rides on the first top-MLP layer's backward launch: inside a capture that must be a captured event record (an edge
    of the graph), else the replayed update could read last step's gradients.  Compared on the per-step table delta."""
    args = H.KAGGLE_ARGS(2048)
    res = {}
    for trace in (False, True):
        app = ffmodel.DLRM(["--backend", HIP] + args)
        app.warmup()
        app.train_steps(5, trace=trace)
        app.model.sync()
        before = _tables_and_mlp(app)
        app.train_steps(1, trace=trace)
        app.model.sync()
        after = _tables_and_mlp(app)
        res[trace] = (before, after)
        if trace:
            assert app.model.uses_graph
        app.close()
    n_tab = 0
    for k in res[True][1]:
        np.testing.assert_allclose(res[True][1][k], res[False][1][k], rtol=2e-5, atol=2e-6, err_msg=k)
        if k.startswith("Embedding"):
            dg, de = res[True][1][k] - res[True][0][k], res[False][1][k] - res[False][0][k]
            scale = np.abs(de).max()
            assert np.abs(dg - de).max() <= 5e-3 * scale, (k, float(np.abs(dg - de).max()), float(scale))
            n_tab += 1
    assert n_tab == 26
