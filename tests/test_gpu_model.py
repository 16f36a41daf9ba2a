"""GPU tests (-m gpu) of the whole path: C++ FFModel shim + DLRM driver over the HIP library.
Parity targets: the torch golden model (tests/golden/dlrm_step_torch.npz) and the same host code
running the CPU oracle as its kernel library."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from dlrm_flexflow_amd import capi, ffmodel
import dlrm_helpers as H

pytestmark = pytest.mark.gpu

HIP = capi.HIP_LIB_PATH


@pytest.mark.parametrize("overlap,graph,dense_update", [(True, False, False), (False, False, False), (True, True, False), (False, False, True)])
def test_dlrm_two_steps_match_torch_golden_on_gpu(hip, overlap, graph, dense_update):
    """forward / zero_gradients / backward / update x2 on the MI355X against the torch model of the
    reference topology: 1e-5 relative (fp32 MFMA GEMMs, fused sparse SGD), eager, side-stream
    overlap, hipGraph trace replay, and the reference's dense embedding path."""
    m, h = H.build_golden_dlrm(HIP, enable_graph=graph, overlap=overlap, dense_update=dense_update)
    recs = H.run_steps(m, h, 2, trace=graph)
    H.check_against_golden(recs, h)
    assert m.uses_graph == graph
    pm = m.perf_metrics()
    mse = float(h["g"]["step0/mse_sum"]) + float(h["g"]["step1/mse_sum"])
    assert abs(pm.mse_loss - mse) <= 1e-5 * mse and pm.train_all == 4 * int(h["g"]["B"])
    m.close()


@pytest.mark.parametrize("graph", [False, True])
def test_adam_optimizer_matches_torch_on_gpu(hip, graph):
    """AdamOptimizer (ffh_adam_update over the MLP slab + dense embedding tables) on the GPU against the
    torch model updated with the reference's formula; eager and hipGraph replay."""
    hp = dict(alpha=0.01, beta1=0.9, beta2=0.999, weight_decay=1e-3, epsilon=1e-8)
    m, h = H.build_golden_dlrm(HIP, enable_graph=graph, overlap=False, adam=hp)
    recs = H.run_steps(m, h, 3, trace=graph)
    exp = H.torch_adam_reference(h["g"], 3, **hp)
    for step in range(3):
        for k in recs[step]:
            np.testing.assert_allclose(recs[step][k], exp[step][k], rtol=2e-5, atol=2e-6, err_msg=f"step {step} {k}")
    m.close()


@pytest.mark.timeout(300)
def test_async_launch_threads_on_gpu(hip):
    """--async-launch: weight-gradient GEMMs and the embedding side stream issued by their own host threads."""
    m, h = H.build_golden_dlrm(HIP, overlap=True, extra_argv=["--async-launch"])
    recs = H.run_steps(m, h, 3)[:2]
    H.check_against_golden(recs, h)
    m.close()


def test_trace_replay_equals_eager(hip):
    """begin_trace/end_trace (hipGraph capture + replay) over 4 steps == 4 eager steps, bit for bit
    on the embedding tables and within 1e-6 on the MLP (atomics in the dW split-K)."""
    a, ha = H.build_golden_dlrm(HIP, enable_graph=True)
    b, hb = H.build_golden_dlrm(HIP, enable_graph=False)
    ra, rb = H.run_steps(a, ha, 4, trace=True), H.run_steps(b, hb, 4)
    for k in ra[3]:
        np.testing.assert_allclose(ra[3][k], rb[3][k], rtol=1e-5, atol=1e-6, err_msg=k)
    a.close(); b.close()


DRIVER_C1 = ["-ll:gpu", "1", "-b", "128", "--arch-sparse-feature-size", "16", "--arch-embedding-size", "-".join(["1000"] * 8),
             "--arch-mlp-bot", "13-64-16", "--arch-mlp-top", "144-64-1", "--data-size", "512"]


def _trajectory(backend, args, steps, trace):
    app = ffmodel.DLRM(["--backend", backend] + args)
    app.warmup()
    app.train_steps(steps, trace=trace)
    app.model.sync()
    out = {}
    for li in range(app.model.num_layers):
        for wi in range(app.model.layer_num_weights(li)):
            out[f"{app.model.layer_name(li)}/{wi}"] = app.model.parameter(li, wi).get_weights()
    out["pred"] = app.model.layer_output(app.model.num_layers - 1).get()
    app.close()
    return out


def test_driver_config1_hip_equals_oracle_backend(hip):
    """BASELINE config 1 through the DLRM application object: 1 warm-up + 5 traced steps on the GPU vs
    the same host code on the CPU oracle.  Same seeded weights and batch on both sides."""
    c = _trajectory(H.oracle_backend(), DRIVER_C1, 5, trace=False)
    for trace in (True, False):
        g = _trajectory(HIP, DRIVER_C1, 5, trace=trace)
        assert g.keys() == c.keys()
        for k in g:
            np.testing.assert_allclose(g[k], c[k], rtol=2e-5, atol=2e-6, err_msg=f"{k} trace={trace}")


@pytest.mark.parametrize("trace", [True, False])
def test_driver_kaggle_shape_hip_equals_oracle_backend(hip, trace):
    """BASELINE config 2 (Criteo-Kaggle: 26 tables with the reference script's row counts, D = 16,
    B = 2048, bot 13-512-256-64-16, top 432-512-256-1) -- 1 warm-up + 2 steps, GPU vs oracle."""
    rows = "1396-550-1761917-507795-290-21-11948-608-3-58176-5237-1497287-3127-26-12153-1068715-10-4836-2085-4-1312273-17-15-110946-91-72655"
    args = ["-ll:gpu", "1", "-b", "2048", "--arch-sparse-feature-size", "16", "--arch-embedding-size", rows,
            "--arch-mlp-bot", "13-512-256-64-16", "--arch-mlp-top", "432-512-256-1", "--data-size", "2048"]
    # trace=False: eager launches on three HIP streams (embedding side stream, forked weight-gradient
    # stream) really overlap; trace=True: the same DAG replayed from a hipGraph
    g = _trajectory(HIP, args, 4, trace=trace)
    c = _trajectory(H.oracle_backend(), args, 4, trace=False)
    for k in g:
        np.testing.assert_allclose(g[k], c[k], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("tril,fused", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("trace", [False, True])
def test_dot_interaction_matches_torch_on_gpu(hip, trace, tril, fused):
    """--arch-interaction-op dot / dot-tril on the MI355X: Reshape / Transpose / BatchMatmul / Flat (or Tril) vs torch."""
    import dot_helpers
    out, got, exp = dot_helpers.run_dot_dlrm(HIP, steps=3, trace=trace, tril=tril, fused=fused)
    for g, e in out:
        for k in g:
            np.testing.assert_allclose(g[k], e[k], rtol=2e-5, atol=2e-6, err_msg=k)
    for k in got:
        np.testing.assert_allclose(got[k], exp[k], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("trace", [False, True])
def test_hdf5_dataset_on_gpu(hip, tmp_path, trace):
    """--dataset (HDF5 Criteo layout) feeding the GPU run: the batches on the device are the file's rows, and
    8 steps (one wrap-around; with trace the batch copies stay outside the replayed graph) follow the oracle backend."""
    h5, exp = H.make_criteo_like_hdf5(str(tmp_path))
    res = {}
    for name, backend in (("hip", HIP), ("cpu", H.oracle_backend())):
        app = ffmodel.DLRM(["--backend", backend] + H.HDF5_ARGS + ["--dataset", h5])
        assert app.num_samples == 96
        app.warmup()
        app.train_steps(7, trace=trace and name == "hip")
        app.model.sync()
        if name == "hip":
            assert np.array_equal(app.dense_input().get(), exp["X_int"][16:32])
            assert np.array_equal(app.sparse_input(2).get(np.int64).reshape(-1), exp["X_cat"][16:32, 2])
        res[name] = {f"p{l}": app.model.parameter(l, 0).get_weights() for l in range(app.model.num_layers) if app.model.layer_num_weights(l)}
        res[name]["pred"] = app.model.layer_output(app.model.num_layers - 1).get()
        app.close()
    for k in res["hip"]:
        np.testing.assert_allclose(res["hip"][k], res["cpu"][k], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("batch,bag", [(2048, 1), (512, 3), (8192, 1)])
def test_zipf_duplicate_row_stress_on_gpu(hip, batch, bag):
    """power-law ids (--zipf-alpha 1.05; a few rows take most of the batch) through the gather and the fused
    update -- the single-launch kernel (batch x bag <= 2048) and the tiled path -- against the oracle backend."""
    args = ["-b", str(batch), "--arch-sparse-feature-size", "16", "--arch-embedding-size", "100000-37-5000-2",
            "--arch-mlp-bot", "13-32-16", "--arch-mlp-top", "80-32-1", "--embedding-bag-size", str(bag), "--data-size", str(2 * batch),
            "--zipf-alpha", "1.05"]
    res = {}
    for name, backend in (("hip", HIP), ("cpu", H.oracle_backend())):
        app = ffmodel.DLRM(["--backend", backend] + args)
        app.warmup()
        app.train_steps(3, trace=False)
        app.model.sync()
        res[name] = {f"p{l}": app.model.parameter(l, 0).get_weights() for l in range(app.model.num_layers) if app.model.layer_num_weights(l)}
        res[name]["pred"] = app.model.layer_output(app.model.num_layers - 1).get()
        res[name]["ids"] = app.sparse_input(0).get(np.int64).astype(np.float64)
        app.close()
    assert np.bincount(res["hip"]["ids"].astype(np.int64).reshape(-1)).max() > batch * bag // 20
    for k in res["hip"]:
        np.testing.assert_allclose(res["hip"][k], res["cpu"][k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_dlrm_executable_on_gpu(hip):
    exe = os.path.join(ROOT, "dlrm_flexflow_amd", "host", "dlrm")
    r = subprocess.run([exe] + DRIVER_C1 + ["--epochs", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "THROUGHPUT = " in r.stdout and "[Metrics]" in r.stderr


@pytest.mark.parametrize("how", ["torch", "direct"])
def test_single_rank_nccl_exchange_path_on_gpu(hip, tmp_path, how):
    """--force-exchange with a 1-rank RCCL group: the all-to-all and all-reduce run on the model's HIP streams, served
    by torch.distributed callbacks (device pointers wrapped zero-copy) or by RCCL called from the C++ host layer
    (grouped ncclSend/ncclRecv + ncclAllReduce on a communicator bootstrapped over the torch group); result equals the plain run."""
    worker = os.path.join(ROOT, "tests", "_dist_worker_gpu.py")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500))
    r = subprocess.run(["python", worker, str(tmp_path), how], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(os.path.join(tmp_path, "rank0.npz"))
    # 2 steps x (forward + backward all-to-all); the MLP gradients: with the torch callbacks (they block the host) one all-reduce per step in
    # update(), with RCCL called from C++ one bucket per >= 4 MB of layers from inside backward() -- this small model is one bucket
    assert int(z["alltoall_calls"]) == 4 and int(z["allreduce_calls"]) + int(z["allreduce_bucket_calls"]) == 2
    m, h = H.build_golden_dlrm(HIP, overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    for k, v in ref[1].items():
        np.testing.assert_allclose(z[f"s1/{k}"], v, rtol=1e-5, atol=1e-6, err_msg=k)


def _run_two_ranks_on_one_gpu(tmp_path, *mode, world=2):
    worker = os.path.join(ROOT, "tests", "_dist_worker_gpu.py")
    port = str(29600 + os.getpid() % 300)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen(["python", worker, str(tmp_path), "staged", *mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]


def test_two_ranks_on_one_gpu_equal_single_rank(hip, tmp_path):
    """world_size 2 on the HIP kernels: both ranks share this box's one GPU, the collectives go through the host-staged
    test transport (tests/host_staged_comm.py over gloo; RCCL refuses two ranks on one device).  Table-wise shards, exchange
    buffers, Concat unpack / pack, global-batch fused update and the gradient all-reduce must reproduce the
    single-rank run: tables within 1e-6 (same canonical order), MLP within 1e-5."""
    z = _run_two_ranks_on_one_gpu(tmp_path)
    m, h = H.build_golden_dlrm(HIP, overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B = int(h["g"]["B"])
    seen = set()
    for r in range(2):
        sl = slice(r * B // 2, (r + 1) * B // 2)
        for step in range(2):
            np.testing.assert_allclose(z[r][f"s{step}/pred"], ref[step]["pred"][sl], rtol=1e-5, atol=1e-6)
            for k, v in ref[step].items():
                key = f"s{step}/{k}"
                if k == "pred":
                    continue
                if k.startswith("emb"):
                    t = int(k.split(".")[1])
                    assert (key in z[r].files) == (t % 2 == r)
                    if t % 2 == r:
                        np.testing.assert_allclose(z[r][key], v, rtol=1e-6, atol=1e-7, err_msg=key)
                        seen.add(t)
                else:
                    np.testing.assert_allclose(z[r][key], v, rtol=1e-5, atol=1e-6, err_msg=key)
        assert int(z[r]["alltoall_calls"]) == 4 and int(z[r]["allreduce_calls"]) == 2
    assert seen == set(range(len(h["g"]["rows"])))


def test_two_ranks_on_one_gpu_column_sharded_table(hip, tmp_path):
    """--column-shard-rows on the HIP kernels with two ranks (BASELINE config 5's sharding at toy size): every rank holds all
    rows x D/2 columns of the big table, gathers its slice for the global batch and updates it from the matching gradient
    columns; the other tables stay table-wise; one all-to-all carries both kinds."""
    z = _run_two_ranks_on_one_gpu(tmp_path, "column")
    m, h = H.build_golden_dlrm(HIP, overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B, D = int(h["g"]["B"]), int(h["g"]["D"])
    rows = list(h["g"]["rows"])
    small_owner = {}
    for r in range(2):
        sl = slice(r * B // 2, (r + 1) * B // 2)
        np.testing.assert_allclose(z[r]["s1/pred"], ref[1]["pred"][sl], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z[r]["s1/top.0.weight"], ref[1]["top.0.weight"], rtol=1e-5, atol=1e-6)
        got = z[r]["s1/emb.1.weight"]
        assert got.shape == (rows[1], D // 2)
        np.testing.assert_allclose(got, ref[1]["emb.1.weight"][:, r * D // 2:(r + 1) * D // 2], rtol=1e-6, atol=1e-7)
        for t in range(len(rows)):
            if t != 1 and f"s1/emb.{t}.weight" in z[r].files:
                np.testing.assert_allclose(z[r][f"s1/emb.{t}.weight"], ref[1][f"emb.{t}.weight"], rtol=1e-6, atol=1e-7)
                small_owner[t] = r
    assert sorted(small_owner) == [0, 2, 3]


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_row_sharded_table(hip, tmp_path, world):
    """--row-shard-rows on the HIP kernels (BASELINE configs[4]'s reduce-scatter variant at toy size): every rank holds
    50 / world rows plus the zero row, gathers partial bag sums for the global batch, the reduce-scatter leaves each rank
    its samples; backward all-gathers the gradients and the fused update touches the local rows only."""
    z = _run_two_ranks_on_one_gpu(tmp_path, "row", world=world)
    m, h = H.build_golden_dlrm(HIP, overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B = int(h["g"]["B"])
    rows = list(h["g"]["rows"])
    for r in range(world):
        sl = slice(r * B // world, (r + 1) * B // world)
        for step in range(2):
            np.testing.assert_allclose(z[r][f"s{step}/pred"], ref[step]["pred"][sl], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z[r]["s1/top.0.weight"], ref[1]["top.0.weight"], rtol=1e-5, atol=1e-6)
        r0, r1 = rows[1] * r // world, rows[1] * (r + 1) // world
        got = z[r]["s1/emb.1.weight"]
        assert got.shape[0] == r1 - r0
        np.testing.assert_allclose(got, ref[1]["emb.1.weight"][r0:r1], rtol=1e-6, atol=1e-7)
        assert int(z[r]["reduce_scatter_calls"]) == 2 and int(z[r]["allgather_calls"]) == 2


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_data_parallel_tables(hip, tmp_path, world):
    """--replicate-embedding-rows on the HIP kernels: the three small tables are data-parallel (every rank a copy in the dense
    slab: own-sample gather, ffh_embedding_bwd_dense into the slab gradient, the MLP's all-reduce bucket and SGD launch), the
    50-row table stays table-wise in the all-to-all.  Every copy equals the single-rank table to 1e-5."""
    z = _run_two_ranks_on_one_gpu(tmp_path, "replicated", world=world)
    m, h = H.build_golden_dlrm(HIP, overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B = int(h["g"]["B"])
    owners = []
    for r in range(world):
        sl = slice(r * B // world, (r + 1) * B // world)
        for step in range(2):
            np.testing.assert_allclose(z[r][f"s{step}/pred"], ref[step]["pred"][sl], rtol=1e-5, atol=1e-6)
            for t in (0, 2, 3):
                np.testing.assert_allclose(z[r][f"s{step}/emb.{t}.weight"], ref[step][f"emb.{t}.weight"], rtol=1e-5, atol=1e-6, err_msg=f"rank {r} table {t}")
        np.testing.assert_allclose(z[r]["s1/top.0.weight"], ref[1]["top.0.weight"], rtol=1e-5, atol=1e-6)
        if "s1/emb.1.weight" in z[r].files:
            owners.append(r)
            np.testing.assert_allclose(z[r]["s1/emb.1.weight"], ref[1]["emb.1.weight"], rtol=1e-6, atol=1e-7)
        assert int(z[r]["alltoall_calls"]) == 4 and int(z[r]["allreduce_calls"]) == 2
    assert owners == [1]


def test_single_rank_rccl_reduce_scatter_path_on_gpu(hip, tmp_path):
    """The same row-sharded step with a 1-rank RCCL group served from the C++ host layer: ncclReduceScatter /
    ncclAllGather are really enqueued on the model's side stream (with one rank they move the data unchanged)."""
    worker = os.path.join(ROOT, "tests", "_dist_worker_gpu.py")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29450 + os.getpid() % 400))
    r = subprocess.run(["python", worker, str(tmp_path), "direct", "row"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(os.path.join(tmp_path, "rank0.npz"))
    assert int(z["reduce_scatter_calls"]) == 2 and int(z["allgather_calls"]) == 2 and int(z["alltoall_calls"]) == 4
    m, h = H.build_golden_dlrm(HIP, overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    for k, v in ref[1].items():
        np.testing.assert_allclose(z[f"s1/{k}"], v, rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_on_one_gpu_kaggle_shape(hip, tmp_path, world):
    """The same at the Criteo-Kaggle shape with 2048 samples per rank (2 ranks: 13 tables each, 4096 lookups per table;
    4 ranks: 7 + 7 + 6 + 6 tables, 8192 lookups per table; 8 ranks: 4 + 4 + 3 x 6 tables, 16384 lookups -- uneven all-to-all blocks -> the tiled radix-sort form of the
    fused update, LDS-DMA GEMMs on each rank's slice): after the warm-up + 3 steps every rank's predictions, MLP weights
    and owned tables equal the one-rank run on the whole batch."""
    z = _run_two_ranks_on_one_gpu(tmp_path, "kaggle", world=world)
    app = ffmodel.DLRM(H.KAGGLE_ARGS(2048 * world, HIP))
    app.warmup(); app.train_steps(3, trace=False); app.model.sync()
    m = app.model
    pred = m.layer_output(m.num_layers - 1).get()
    owned = [0] * world
    for r in range(world):
        np.testing.assert_allclose(z[r]["pred"], pred[r * 2048:(r + 1) * 2048], rtol=2e-5, atol=2e-6)
        for l in range(m.num_layers):
            if not m.layer_num_weights(l):
                continue
            key = f"p{l}"
            is_table = m.layer_name(l).startswith("Embedding")
            if is_table and key not in z[r].files:
                continue
            owned[r] += is_table
            w = m.parameter(l, 0).get_weights()
            exp = w if w.size <= 1 << 16 else np.array([w.astype(np.float64).sum(), np.abs(w).astype(np.float64).sum(), float(w[:64].astype(np.float64).sum())])
            np.testing.assert_allclose(z[r][key], exp, rtol=2e-5, atol=2e-6, err_msg=f"rank {r} layer {l}")
    assert owned == [len(range(r, 26, world)) for r in range(world)]
    app.close()


def test_replicated_tables_steady_state_is_ordered_behind_the_slab_update(hip, tmp_path):
    """Round-2 advisor finding (high): data-parallel tables live in the dense parameter slab, which update() writes on the
    compute stream; the next step's gather reads them on the side stream.  With a resident batch (no new-batch event) the
    gather used to be ordered only behind the step's gradients, not behind the optimizer.  Six steady-state steps with the
    overlap on must equal the same run with --no-overlap (everything on one stream): predictions and every replicated
    table bit for bit on the tables' own arithmetic, 1e-6 on the MLP (atomics in split-K dW)."""
    (tmp_path / "ov").mkdir(); (tmp_path / "noov").mkdir()
    a = _run_two_ranks_on_one_gpu(tmp_path / "ov", "kaggle-repl")
    b = _run_two_ranks_on_one_gpu(tmp_path / "noov", "kaggle-repl-noov")
    for r in range(2):
        assert a[r].files == b[r].files
        for k in a[r].files:
            if k.endswith("_calls"):
                continue
            np.testing.assert_allclose(a[r][k], b[r][k], rtol=2e-6, atol=2e-7, err_msg=f"rank {r} {k}")
        assert int(a[r]["allreduce_calls"]) >= 6
