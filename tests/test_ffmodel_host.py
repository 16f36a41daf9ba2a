"""CPU tests of the host layer (C++ FFModel shim + DLRM driver + Python binding + ffcomm).

The operator kernels are supplied by the CPU oracle here -- test infrastructure standing in for
the GPU so that graph construction, aliasing into the concat buffer, the training-step order,
the table-wise sharding and the collectives can be checked without a GPU.  The same host code
runs the HIP library in tests/test_gpu_model.py.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden
from dlrm_flexflow_amd import build, capi, ffmodel
import dlrm_helpers as H


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build_host()


@pytest.mark.parametrize("overlap,dense_update", [(True, False), (False, False), (False, True)])
def test_dlrm_two_steps_match_torch_golden(overlap, dense_update):
    """Whole-model parity: forward / zero_gradients / backward / update x2 against the torch
    model of the reference topology (tests/golden/make_golden.py), 1e-5 relative."""
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=overlap, dense_update=dense_update)
    recs = H.run_steps(m, h, 2)
    H.check_against_golden(recs, h)
    pm = m.perf_metrics()
    B = int(h["g"]["B"])
    assert pm.train_all == 2 * 2 * B          # two steps, the reference's double count for 1 class + accuracy
    mse = float(h["g"]["step0/mse_sum"]) + float(h["g"]["step1/mse_sum"])
    assert abs(pm.mse_loss - mse) <= 1e-5 * mse
    m.close()


@pytest.mark.timeout(180)
def test_launch_worker_threads_on_cpu():
    """The multi-threaded launch path (dW GEMMs and the embedding side stream issued by their own host
    threads, joined through drain() + events) with the CPU oracle as the device: same results as the
    inline path, and no deadlock (hard timeout)."""
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=True, extra_argv=["--async-launch", "--force-async-launch"])
    recs = H.run_steps(m, h, 2)
    H.check_against_golden(recs, h)
    m.close()


@pytest.mark.parametrize("tril,fused", [(False, False), (True, False), (True, True)])
def test_dot_interaction_matches_torch(tril, fused):
    """--arch-interaction-op dot (Reshape / Transpose / BatchMatmul / Flat composition) and dot-tril
    (MLPerf-DLRM's strict lower triangle of Z Z^T, SURVEY 8a-8: torch's Z[:, li, lj] is the only oracle there is):
    predictions and every parameter after warm-up + 2 steps against a torch model, 1e-5."""
    import dot_helpers
    out, got, exp = dot_helpers.run_dot_dlrm(H.oracle_backend(), steps=2, tril=tril, fused=fused)
    for g, e in out:
        for k in g:
            np.testing.assert_allclose(g[k], e[k], rtol=1e-5, atol=1e-6, err_msg=k)
    assert set(got) == set(exp)
    for k in got:
        np.testing.assert_allclose(got[k], exp[k], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("wd", [0.0, 1e-3])
def test_adam_optimizer_matches_torch(wd):
    """AdamOptimizer through the FFModel API (dense embedding gradients, as the reference: every row's
    moments decay every step) against a torch model updated with the reference's formula."""
    hp = dict(alpha=0.01, beta1=0.9, beta2=0.999, weight_decay=wd, epsilon=1e-8)
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False, adam=hp)
    recs = H.run_steps(m, h, 3)
    exp = H.torch_adam_reference(h["g"], 3, **hp)
    for step in range(3):
        assert set(recs[step]) == set(exp[step])
        for k in recs[step]:
            np.testing.assert_allclose(recs[step][k], exp[step][k], rtol=2e-5, atol=2e-6, err_msg=f"step {step} {k}")
    # weights really moved by about alpha per step (Adam's signature), also in rows of tables the batch never touched
    # once they had a gradient -- and untouched-forever rows stay put
    g = h["g"]
    t0 = recs[2]["emb.0.weight"] - g["init/emb.0.weight"]
    touched = np.zeros(t0.shape[0], bool); touched[np.unique(g["sparse0"])] = True
    assert np.abs(t0[touched]).max() > 0.01 and (wd > 0 or not t0[~touched].any())
    m.close()


def test_fused_and_dense_embedding_paths_agree():
    """The fused sparse update and the reference's dense zero/scatter/sweep path give the same
    tables (1e-6: only the summation order inside duplicate rows differs)."""
    a, ha = H.build_golden_dlrm(H.oracle_backend(), overlap=False, dense_update=False)
    b, hb = H.build_golden_dlrm(H.oracle_backend(), overlap=False, dense_update=True)
    ra, rb = H.run_steps(a, ha, 2), H.run_steps(b, hb, 2)
    for k in ra[1]:
        np.testing.assert_allclose(ra[1][k], rb[1][k], rtol=1e-6, atol=1e-7, err_msg=k)
    a.close(); b.close()


def test_gradients_are_exposed_like_the_reference():
    """Tensor::get_grad after backward(): concat slices alias the concat gradient (bit-equal)."""
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    m.forward(); m.zero_gradients(); m.backward(); m.sync()
    n_bot = len(h["g"]["bot"]) - 1
    concat_layer = n_bot + len(h["g"]["rows"])
    dz = m.layer_output(concat_layer).get_grad()
    D = int(h["g"]["D"])
    off = D
    for t in range(len(h["g"]["rows"])):
        ge = m.layer_output(n_bot + t).get_grad()
        assert np.array_equal(ge, dz[:, off:off + D])
        off += D
    m.close()


def test_dlrm_driver_binary_runs_the_tiny_config():
    """BASELINE config 1 (8 tables x 1000 rows, emb_dim 16, batch 128) through the `dlrm`
    executable with the reference's own flags; prints the reference's THROUGHPUT line."""
    exe = os.path.join(ROOT, "dlrm_flexflow_amd", "host", "dlrm")
    args = [exe, "--backend", H.oracle_backend(), "-ll:gpu", "1", "-b", "128", "--arch-sparse-feature-size", "16",
            "--arch-embedding-size", "-".join(["1000"] * 8), "--arch-mlp-bot", "13-64-16", "--arch-mlp-top", "144-64-1",
            "--epochs", "2", "--data-size", "512"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "THROUGHPUT = " in r.stdout and "samples/s" in r.stdout
    assert "parameters.size() = 16" in r.stdout        # 5 Linear x (kernel, bias) + ... as the reference prints
    assert "[Metrics]" in r.stderr and "mean_squared_error" in r.stderr


def test_driver_rejects_what_the_reference_rejects():
    exe = os.path.join(ROOT, "dlrm_flexflow_amd", "host", "dlrm")
    base = [exe, "--backend", H.oracle_backend(), "-b", "8", "--arch-embedding-size", "10-10", "--arch-sparse-feature-size", "4",
            "--arch-mlp-bot", "3-4", "--arch-mlp-top", "12-1"]
    r = subprocess.run(base + ["--arch-interaction-op", "sum"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "'cat', 'dot', 'dot-tril' or 'dot-tril-ops'" in r.stderr
    r = subprocess.run(base + ["--dataset", "x.h5"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "HDF5" in r.stderr
    # -ll:gpu 4 through the library API without a communicator: the model refuses (the `dlrm` binary and run_dlrm.py start
    # their own ranks instead, tests/test_launchers.py)
    code = ("import sys; sys.path.insert(0, %r); from dlrm_flexflow_amd import ffmodel; ffmodel.DLRM(sys.argv[1:])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code] + base[1:] + ["-ll:gpu", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "one process per GPU" in r.stderr
    r = subprocess.run([exe, "--backend", "/nonexistent/libffhip.so", "-b", "8"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "cannot load kernel library" in r.stderr     # no silent fallback


def test_dlrm_app_object_and_data_loader_are_deterministic():
    """Same seed -> same synthetic batch and same trajectory; different seed -> different."""
    args = ["--backend", H.oracle_backend(), "-b", "32", "--arch-sparse-feature-size", "8", "--arch-embedding-size", "50-7-300",
            "--arch-mlp-bot", "13-16-8", "--arch-mlp-top", "32-16-1", "--data-size", "64"]
    outs = []
    for seed in (0, 0, 1):
        app = ffmodel.DLRM(args + ["--seed", str(seed)])
        app.warmup()
        app.train_steps(3, trace=False)
        app.model.sync()
        outs.append((app.sparse_input(1).get(np.int64), app.model.parameter(0, 0).get_weights(), app.model.parameter(3, 0).get_weights()))
        assert app.num_samples == 64 and app.num_tables == 3
        app.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert not np.array_equal(outs[0][0], outs[2][0])
    assert outs[0][0].min() >= 0 and outs[0][0].max() < 7


# ---------------------------------------------------------------------------------------------
# two ranks over gloo: table-wise sharding + all-to-all + bucketed all-reduce
# ---------------------------------------------------------------------------------------------
WORKER = os.path.join(ROOT, "tests", "_dist_worker.py")


def _run_ranks(world, tmp_path, mode):
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return outs


def test_two_rank_gloo_run_equals_single_rank(tmp_path):
    """world_size 2 (gloo, CPU): rank r owns tables r, r+2; each rank runs half the batch through the
    MLPs; the exchange is an all-to-all each way, MLP gradients one all-reduce.  Result must equal
    the single-rank run: embedding tables bit-exact (same canonical order), MLP within 1e-5."""
    _run_ranks(2, tmp_path, "golden")
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B = int(h["g"]["B"])
    seen_tables = set()
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        for step in range(2):
            sl = slice(r * B // 2, (r + 1) * B // 2)
            np.testing.assert_allclose(z[f"s{step}/pred"], ref[step]["pred"][sl], rtol=1e-5, atol=1e-6)
            for k, v in ref[step].items():
                key = f"s{step}/{k}"
                if k == "pred":
                    continue
                if k.startswith("emb"):
                    t = int(k.split(".")[1])
                    if t % 2 == r:
                        assert key in z.files
                        np.testing.assert_allclose(z[key], v, rtol=1e-6, atol=1e-7, err_msg=key)
                        seen_tables.add(t)
                    else:
                        assert key not in z.files              # sole owner: never replicated
                else:
                    np.testing.assert_allclose(z[key], v, rtol=1e-5, atol=1e-6, err_msg=key)
        assert int(z["alltoall_calls"]) == 2 * 2 and int(z["allreduce_calls"]) == 2   # 2 steps x (fwd + bwd), 2 x 1 bucket
    assert seen_tables == set(range(len(h["g"]["rows"])))
    # and against the torch golden itself
    z0 = np.load(os.path.join(tmp_path, "rank0.npz"))
    np.testing.assert_allclose(z0["s1/top.0.weight"], h["g"]["step1/top.0.weight"], rtol=1e-5, atol=1e-6)


def test_two_rank_column_sharded_giant_table(tmp_path):
    """--column-shard-rows 40: the 50-row table is split column-wise (every rank holds all rows x D/2
    columns and gathers its slice for the global batch), the others stay table-wise; one all-to-all
    carries both kinds.  Every rank's slice must equal the same columns of the single-rank table."""
    _run_ranks(2, tmp_path, "column")
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B, D = int(h["g"]["B"]), int(h["g"]["D"])
    rows = list(h["g"]["rows"])
    big = [t for t, r in enumerate(rows) if r >= 40]
    assert big == [1]
    owner_of_small = {}
    k = 0
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        sl = slice(r * B // 2, (r + 1) * B // 2)
        np.testing.assert_allclose(z["s1/pred"], ref[1]["pred"][sl], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["s1/top.0.weight"], ref[1]["top.0.weight"], rtol=1e-5, atol=1e-6)
        got = z["s1/emb.1.weight"]
        assert got.shape == (rows[1], D // 2)
        np.testing.assert_allclose(got, ref[1]["emb.1.weight"][:, r * D // 2:(r + 1) * D // 2], rtol=1e-6, atol=1e-7)
        for t in range(len(rows)):
            if t not in big and f"s1/emb.{t}.weight" in z.files:
                np.testing.assert_allclose(z[f"s1/emb.{t}.weight"], ref[1][f"emb.{t}.weight"], rtol=1e-6, atol=1e-7)
                owner_of_small[t] = r
    assert sorted(owner_of_small) == [0, 2, 3]


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_giant_table(tmp_path, world):
    """--row-shard-rows 40 (BASELINE configs[4]'s reduce-scatter variant): the 50-row table is split row-wise, every rank
    gathers partial bag sums for the global batch (rows held elsewhere read the zero row), a reduce-scatter adds the
    partials and leaves each rank its samples; backward all-gathers the output gradients and each rank updates its rows.
    The other tables stay table-wise in the all-to-all (world 4: rank 1 owns none of them).  Every rank's rows must equal the same rows of the 1-rank table."""
    _run_ranks(world, tmp_path, "row")
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B = int(h["g"]["B"])
    rows = list(h["g"]["rows"])
    assert [t for t, r in enumerate(rows) if r >= 40] == [1]
    small_seen = set()
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        sl = slice(r * B // world, (r + 1) * B // world)
        np.testing.assert_allclose(z["s0/pred"], ref[0]["pred"][sl], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["s1/pred"], ref[1]["pred"][sl], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["s1/top.0.weight"], ref[1]["top.0.weight"], rtol=1e-5, atol=1e-6)
        got = z["s1/emb.1.weight"]
        r0, r1 = rows[1] * r // world, rows[1] * (r + 1) // world      # 50 rows over 4 ranks: 12, 13, 12, 13
        assert got.shape == (r1 - r0, ref[1]["emb.1.weight"].shape[1])
        np.testing.assert_allclose(got, ref[1]["emb.1.weight"][r0:r1], rtol=1e-6, atol=1e-7)
        for t in range(len(rows)):
            if t != 1 and f"s1/emb.{t}.weight" in z.files:
                np.testing.assert_allclose(z[f"s1/emb.{t}.weight"], ref[1][f"emb.{t}.weight"], rtol=1e-6, atol=1e-7)
                small_seen.add(t)
        assert int(z["reduce_scatter_calls"]) == 2 and int(z["allgather_calls"]) == 2 and int(z["alltoall_calls"]) == 4
    assert small_seen == {0, 2, 3}


@pytest.mark.parametrize("world,mode", [(2, "replicated"), (4, "replicated"), (2, "replicated_all")])
def test_data_parallel_replicated_tables(tmp_path, world, mode):
    """--replicate-embedding-rows N: tables with at most N rows are data-parallel -- the reference's default placement for an
    op without a strategy entry [ref: src/runtime/model.cc:500-510]: every rank holds the table, gathers its own samples and
    scatter-adds a dense gradient that rides in the MLP's all-reduce bucket; the 50-row table stays table-wise in the
    all-to-all (mode replicated) or nothing is exchanged at all (replicated_all).  Every copy must equal the 1-rank table."""
    _run_ranks(world, tmp_path, mode)
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    B = int(h["g"]["B"])
    rows = list(h["g"]["rows"])
    repl = [t for t, r in enumerate(rows) if r <= (39 if mode == "replicated" else 1000)]
    assert repl == ([0, 2, 3] if mode == "replicated" else [0, 1, 2, 3])
    owners_of_big = []
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        sl = slice(r * B // world, (r + 1) * B // world)
        for step in range(2):
            np.testing.assert_allclose(z[f"s{step}/pred"], ref[step]["pred"][sl], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(z[f"s{step}/top.0.weight"], ref[step]["top.0.weight"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(z[f"s{step}/bot.0.bias"], ref[step]["bot.0.bias"], rtol=1e-5, atol=1e-6)
            for t in repl:       # a copy on every rank; dense atomics + all-reduce instead of the canonical order: 1e-5, not bit-exact
                np.testing.assert_allclose(z[f"s{step}/emb.{t}.weight"], ref[step][f"emb.{t}.weight"], rtol=1e-5, atol=1e-6, err_msg=f"rank {r} table {t}")
        if mode == "replicated":
            if "s1/emb.1.weight" in z.files:
                owners_of_big.append(r)
                np.testing.assert_allclose(z["s1/emb.1.weight"], ref[1]["emb.1.weight"], rtol=1e-6, atol=1e-7)
            assert int(z["alltoall_calls"]) == 4 and int(z["allreduce_calls"]) == 2
        else:
            assert int(z["alltoall_calls"]) == 0 and int(z["allreduce_calls"]) == 2
    if mode == "replicated":
        assert owners_of_big == [1]


def test_data_parallel_tables_from_a_strategy_file_and_under_adam(tmp_path):
    """A strategy file that splits two tables over the sample dim (the reference's data-parallel config) replicates them;
    --export writes that placement back.  And a purely data-parallel job (every table replicated: the reference's default
    placement) under Adam, whose moments for the tables live in the slab: the one multi-rank case that can leave plain SGD."""
    g = golden("dlrm_step_torch")
    nb = len(g["bot"]) - 1
    entries = [(f"Embedding_{100 + nb + 0}", [1, 2], [0, 1]), (f"Embedding_{100 + nb + 2}", [1, 2], [0, 1])]
    (tmp_path / "strategy.txt").write_text(_strategy_text(entries))
    _run_ranks(2, tmp_path, "strategy")
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        for t in (0, 2):
            np.testing.assert_allclose(z[f"s1/emb.{t}.weight"], ref[1][f"emb.{t}.weight"], rtol=1e-5, atol=1e-6)
        assert ("s1/emb.1.weight" in z.files) == (r == 1) and ("s1/emb.3.weight" in z.files) == (r == 1)   # the others: table t on rank t % 2
        np.testing.assert_allclose(z["s1/top.0.weight"], ref[1]["top.0.weight"], rtol=1e-5, atol=1e-6)
    exp = _parse_strategy(tmp_path / "export.txt")
    assert exp[f"Embedding_{100 + nb + 0}"] == (0, [1, 2], [0, 1]) and exp[f"Embedding_{100 + nb + 1}"] == (0, [1, 1], [1])
    # Adam
    (tmp_path / "adam").mkdir()
    _run_ranks(2, tmp_path / "adam", "replicated_adam")
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False, adam=dict(alpha=0.001))
    ref = H.run_steps(m, h, 2)
    m.close()
    for r in range(2):
        z = np.load(os.path.join(tmp_path / "adam", f"rank{r}.npz"))
        for t in (0, 1, 2, 3):
            np.testing.assert_allclose(z[f"s1/emb.{t}.weight"], ref[1][f"emb.{t}.weight"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(z["s1/top.0.weight"], ref[1]["top.0.weight"], rtol=1e-5, atol=2e-6)


def _strategy_text(entries):
    """The reference's text format [ref: src/runtime/strategy.cc:147-189]: count, then per op name / device type /
    nDims / dims / number of ids / ids, one per line, lists tab-separated."""
    lines = [str(len(entries))]
    for name, dims, ids in entries:
        lines += [name, "0", str(len(dims)), "\t".join(map(str, dims)) + "\t", str(len(ids)), "\t".join(map(str, ids)) + "\t"]
    return "\n".join(lines) + "\n"


def _parse_strategy(path):
    tok = open(path).read().split()
    n, i, out = int(tok[0]), 1, {}
    for _ in range(n):
        name, dev, nd = tok[i], int(tok[i + 1]), int(tok[i + 2]); i += 3
        dims = [int(x) for x in tok[i:i + nd]]; i += nd
        k = int(tok[i]); i += 1
        out[name] = (dev, dims, [int(x) for x in tok[i:i + k]]); i += k
    assert i == len(tok)
    return out


@pytest.mark.parametrize("owners", [(1, 0, 1, 0), (1, 1, 1, 1)])
def test_two_rank_strategy_file_places_the_tables(tmp_path, owners):
    """--import a strategy file in the reference's text format that moves the tables (swapped round-robin;
    all four on rank 1, so rank 0 owns none), leaves one op to the default and names the MLP ops data-parallel.
    Results equal the single-rank run; --export writes the placement in force, which loads back."""
    g = golden("dlrm_step_torch")
    nb = len(g["bot"]) - 1
    entries = [(f"Embedding_{100 + nb + t}", [1, 1], [owners[t]]) for t in range(4)]
    entries += [("Dense_100", [1, 2], [0, 1]), (f"Concat_{100 + nb + 4}", [1, 2], [0, 1]), ("NotInThisModel_7", [1, 1, 1, 4], [0, 1, 2, 3])]
    (tmp_path / "strategy.txt").write_text(_strategy_text(entries))
    _run_ranks(2, tmp_path, "strategy")
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    ref = H.run_steps(m, h, 2)
    n_layers = m.num_layers
    m.close()
    B = int(g["B"])
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        sl = slice(r * B // 2, (r + 1) * B // 2)
        for step in range(2):
            np.testing.assert_allclose(z[f"s{step}/pred"], ref[step]["pred"][sl], rtol=1e-5, atol=1e-6)
            for t in range(4):
                key = f"s{step}/emb.{t}.weight"
                assert (key in z.files) == (owners[t] == r)
                if owners[t] == r:
                    np.testing.assert_allclose(z[key], ref[step][f"emb.{t}.weight"], rtol=1e-6, atol=1e-7, err_msg=key)
            np.testing.assert_allclose(z[f"s{step}/top.0.weight"], ref[step]["top.0.weight"], rtol=1e-5, atol=1e-6)
    exp = _parse_strategy(tmp_path / "export.txt")
    assert len(exp) == n_layers
    for t in range(4):
        assert exp[f"Embedding_{100 + nb + t}"] == (0, [1, 1], [owners[t]])
    assert exp["Dense_100"] == (0, [1, 2], [0, 1]) and exp[f"Concat_{100 + nb + 4}"] == (0, [1, 2], [0, 1])


def test_strategy_file_round_trip_and_rejections(tmp_path):
    """Single rank: export -> import round trip; a strategy this build cannot honour aborts with a message naming the op
    (the reference asserts, include/cuda_helper.h / strategy.cc:91)."""
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False, extra_argv=["--export", str(tmp_path / "a.txt")])
    n_layers = m.num_layers
    a = H.run_steps(m, h, 1)
    m.close()
    exp = _parse_strategy(tmp_path / "a.txt")
    assert len(exp) == n_layers and all(v == (0, [1, 1], [0]) for v in exp.values())
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False, extra_argv=["--import", str(tmp_path / "a.txt"), "--export", str(tmp_path / "b.txt")])
    b = H.run_steps(m, h, 1)
    m.close()
    assert (tmp_path / "a.txt").read_text() == (tmp_path / "b.txt").read_text()
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k])
    bad = {
        "split.txt": _strategy_text([("Embedding_102", [1, 2], [0, 1])]),          # table split over the sample dim
        "device.txt": _strategy_text([("Embedding_102", [1, 1], [3])]),            # device the job does not have
        "model.txt": _strategy_text([("Dense_100", [2, 1], [0, 1])]),              # channel-parallel Linear
        "dims.txt": _strategy_text([("Dense_100", [1, 1, 1], [0])]),               # wrong dimensionality (reference: assert)
        "dup.txt": _strategy_text([("Dense_100", [1, 1], [0]), ("Dense_100", [1, 1], [0])]),
    }
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import dlrm_helpers as H; "
            "H.build_golden_dlrm(H.oracle_backend(), overlap=False, extra_argv=['--import', sys.argv[1]])") % (ROOT, os.path.join(ROOT, "tests"))
    for name, text in bad.items():
        (tmp_path / name).write_text(text)
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path / name)], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "FATAL" in r.stderr, (name, r.stderr[-400:])
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "missing.txt")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "strategy file" in r.stderr


def test_hdf5_criteo_loader(tmp_path):
    """--dataset reads X_int / X_cat / y of the reference's HDF5 layout (written here by the
    preprocess_hdf.py counterpart); batches arrive in file order, whole batches only, wrapping around."""
    h5, exp = H.make_criteo_like_hdf5(str(tmp_path))
    from dlrm_flexflow_amd import hdf5_lite
    for k in exp:                                                   # the file itself holds what the reference script writes
        got = hdf5_lite.read(h5, k)
        assert got.dtype == exp[k].dtype and np.array_equal(got, exp[k])
    app = ffmodel.DLRM(["--backend", H.oracle_backend()] + H.HDF5_ARGS + ["--dataset", h5])
    assert app.num_samples == 96 and app.num_tables == 3            # 100 samples = 6 whole batches of 16
    B = 16
    app.warmup()
    for step in range(8):                                           # batches 0..5, then 0, 1 again
        app.model.sync()
        k = step % 6
        assert np.array_equal(app.dense_input().get(), exp["X_int"][k * B:(k + 1) * B])
        assert np.array_equal(app.model.label_tensor.get().reshape(-1), exp["y"][k * B:(k + 1) * B])
        for t in range(3):
            assert np.array_equal(app.sparse_input(t).get(np.int64).reshape(-1), exp["X_cat"][k * B:(k + 1) * B, t])
        app.train_steps(1, trace=False)
    pm = app.model.perf_metrics()
    assert pm.train_all > 0 and np.isfinite(pm.mse_loss)
    app.close()
    # --data-size caps what is loaded
    app = ffmodel.DLRM(["--backend", H.oracle_backend()] + H.HDF5_ARGS + ["--dataset", h5, "--data-size", "40"])
    assert app.num_samples == 32
    app.close()


def test_hdf5_loader_rejections(tmp_path):
    """Ids outside a table, a dataset of the wrong shape and a missing libhdf5 stop the run with a message
    (the reference asserts on the shapes, examples/cpp/DLRM/dlrm.cc:288-310)."""
    exe = os.path.join(ROOT, "dlrm_flexflow_amd", "host", "dlrm")
    base = [exe, "--backend", H.oracle_backend()] + H.HDF5_ARGS
    h5, _ = H.make_criteo_like_hdf5(str(tmp_path), bad_id=True)
    r = subprocess.run(base + ["--dataset", h5], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "is outside table 1 (7 rows)" in r.stderr
    r = subprocess.run(base[:-4] + ["--arch-mlp-bot", "12-16-8", "--arch-mlp-top", "32-16-1", "--dataset", h5], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "X_int's second dimension" in r.stderr
    r = subprocess.run(base + ["--dataset", h5, "--embedding-bag-size", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "X_cat's second dimension" in r.stderr
    r = subprocess.run(base + ["--dataset", h5], capture_output=True, text=True, timeout=120, env=dict(os.environ, FFH_HDF5_LIB="/nonexistent/libhdf5.so"))
    assert r.returncode != 0 and "needs libhdf5" in r.stderr


def test_two_rank_hdf5_dataset(tmp_path):
    """Two ranks read the same file: each keeps its half of every batch (dense, labels) and all ids of the tables it
    owns; the trajectory equals the single-rank run on the same file."""
    h5, exp = H.make_criteo_like_hdf5(str(tmp_path))
    _run_ranks(2, tmp_path, "hdf5")
    app = ffmodel.DLRM(["--backend", H.oracle_backend()] + H.HDF5_ARGS + ["--dataset", h5])
    app.warmup(); app.train_steps(7, trace=False); app.model.sync()
    ref = {"w_bot": app.model.parameter(0, 0).get_weights(), "w_top": app.model.parameter(app.model.num_layers - 1, 0).get_weights()}
    for t in range(3):
        ref[f"emb{t}"] = app.model.parameter(2 + t, 0).get_weights()
    app.close()
    owned = set()
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert int(z["num_samples"]) == 96
        assert np.array_equal(z["dense"], exp["X_int"][16 + 8 * r:16 + 8 * (r + 1)])      # 8th step = batch 1, this rank's half
        assert np.array_equal(z["label"].reshape(-1), exp["y"][16 + 8 * r:16 + 8 * (r + 1)])
        for t in range(3):
            if f"sparse{t}" in z.files:
                owned.add(t)
                assert np.array_equal(z[f"sparse{t}"].reshape(-1), exp["X_cat"][16:32, t])   # the owner gathers for the global batch
                np.testing.assert_allclose(z[f"emb{t}"], ref[f"emb{t}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["w_bot"], ref["w_bot"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["w_top"], ref["w_top"], rtol=1e-5, atol=1e-6)
    assert owned == {0, 1, 2}


def test_single_rank_forced_exchange_equals_plain_run(tmp_path):
    """--force-exchange: one rank still goes through the all-to-all / all-reduce callbacks (how the
    collectives are exercised on a 1-GPU box); results equal the plain single-rank run."""
    _run_ranks(1, tmp_path, "golden")
    z = np.load(os.path.join(tmp_path, "rank0.npz"))
    assert int(z["alltoall_calls"]) == 4 and int(z["allreduce_calls"]) == 2
    m, h = H.build_golden_dlrm(H.oracle_backend(), overlap=False)
    ref = H.run_steps(m, h, 2)
    m.close()
    for k, v in ref[1].items():
        np.testing.assert_allclose(z[f"s1/{k}"], v, rtol=1e-6, atol=1e-7, err_msg=k)


def test_two_rank_fused_dot_interaction(tmp_path):
    """--arch-interaction-op dot-tril (one launch each way) under 2 ranks equals the single-rank run of the operator
    chain (dot-tril-ops): predictions, MLPs, and every table on its owner."""
    _run_ranks(2, tmp_path, "dot")
    app = ffmodel.DLRM(["--backend", H.oracle_backend()] + H.DOT_ARGS + ["--arch-interaction-op", "dot-tril-ops"])
    app.warmup()
    app.train_steps(3, trace=False)
    m = app.model
    m.sync()
    pred = m.layer_output(m.num_layers - 1).get()
    names = [m.layer_name(l) for l in range(m.num_layers)]
    ref = {names[l]: m.parameter(l, 0).get_weights() for l in range(m.num_layers) if m.layer_num_weights(l)}
    app.close()
    B = pred.shape[0]
    seen = set()
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        np.testing.assert_allclose(z["pred"], pred[r * B // 2:(r + 1) * B // 2], rtol=1e-5, atol=1e-6)
        # layer numbering differs after the interaction (2 ops instead of 6): match Dense / Embedding layers in order
        mine = [k for k in z.files if k.startswith("p") and k[1:].isdigit()]
        fused_names = [n for n in names if n.split("_")[0] in ("Dense", "Embedding")]
        dense_emb = sorted(mine, key=lambda k: int(k[1:]))
        held = [n for n in fused_names if n.startswith("Dense") or (int(n.split("_")[1]) - 102) % 2 == r]
        assert len(dense_emb) == len(held)
        for k, n in zip(dense_emb, held):
            np.testing.assert_allclose(z[k], ref[n], rtol=1e-5, atol=1e-6, err_msg=n)
            seen.add(n)
    assert seen == set(fused_names)


def test_narrow_layer_pair_path_equals_plain_calls():
    """A bottom MLP ending 96 -> 64 -> 16 takes the ffh_linear_pair_bwd route of the host layer (upper backward + lower dX
    in one call, then the lower dW): on the oracle backend that is the same arithmetic as --no-fused-pair, bit for bit."""
    args = ["--backend", H.oracle_backend(), "-b", "48", "--arch-sparse-feature-size", "16", "--arch-embedding-size", "30-11",
            "--arch-mlp-bot", "13-96-64-16", "--arch-mlp-top", "48-24-1", "--data-size", "48", "--no-mlp-chain"]     # (the chain launches of round 5 would take these layers first)
    res = []
    for extra in ([], ["--no-fused-pair"]):
        app = ffmodel.DLRM(args + extra)
        app.warmup()
        app.train_steps(3, trace=False)
        m = app.model
        m.sync()
        res.append({l: m.parameter(l, 0).get_weights() for l in range(m.num_layers) if m.layer_num_weights(l)})
        res[-1]["pred"] = m.layer_output(m.num_layers - 1).get()
        app.close()
    assert set(res[0]) == set(res[1])
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_two_rank_driver_flags(tmp_path):
    """The DLRM application object under 2 ranks with the driver's flags (7 tables over 2 ranks:
    4 + 3, uneven all-to-all splits); loss decreases and both ranks hold identical MLPs."""
    outs = _run_ranks(2, tmp_path, "driver")
    z0 = np.load(os.path.join(tmp_path, "rank0.npz"))
    z1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    assert np.array_equal(z0["w_top"], z1["w_top"]) and np.array_equal(z0["w_bot"], z1["w_bot"])
    assert float(z0["mse_last"]) < float(z0["mse_first"])
    assert "THROUGHPUT" in outs[0]


def test_zipf_index_stream():
    """--zipf-alpha (not a reference flag) draws power-law ids for the duplicate-row stress: in range,
    reproducible from --seed, rank frequencies follow (k+1)^-alpha, and the hot rows are spread over the table."""
    R, B = 5000, 4096
    args = ["--backend", H.oracle_backend(), "-b", str(B), "--arch-sparse-feature-size", "8", "--arch-embedding-size", f"{R}-{R}-3",
            "--arch-mlp-bot", "13-16-8", "--arch-mlp-top", "32-16-1", "--embedding-bag-size", "2", "--data-size", str(4 * B)]
    def ids(extra):
        app = ffmodel.DLRM(args + extra)
        app.warmup()
        app.model.sync()
        out = [app.sparse_input(t).get(np.int64).copy() for t in range(3)]
        app.train_steps(2, trace=False)
        pm = app.model.perf_metrics()
        assert np.isfinite(pm.mse_loss)
        app.close()
        return out
    z = ids(["--zipf-alpha", "1.05"])
    assert all(a.shape == (B, 2) for a in z)
    assert all(a.min() >= 0 for a in z) and z[0].max() < R and z[2].max() < 3
    assert all(np.array_equal(a, b) for a, b in zip(z, ids(["--zipf-alpha", "1.05"])))          # same seed, same stream
    assert not np.array_equal(z[0], ids(["--zipf-alpha", "1.05", "--seed", "3"])[0])
    assert not np.array_equal(z[0], z[1])                                                       # tables draw independently
    counts = np.sort(np.bincount(z[0].reshape(-1), minlength=R))[::-1].astype(np.float64)
    n = counts.sum()
    a1 = 1 - 1.05
    cdf = lambda x: ((x ** a1) - 1) / (((R + 1.0) ** a1) - 1)
    exp_top = cdf(2.0) - cdf(1.0)                                                               # the hottest row's share
    assert abs(counts[0] / n - exp_top) < 4 * np.sqrt(exp_top / n) + 0.01
    exp_top10 = cdf(11.0) - cdf(1.0)
    assert abs(counts[:10].sum() / n - exp_top10) < 0.03
    hot = np.argsort(np.bincount(z[0].reshape(-1), minlength=R))[::-1][:8]
    assert np.ptp(hot) > R // 8                                                                 # not the first rows of the table
    u = ids([])                                                                                 # the reference's uniform draw
    assert np.bincount(u[0].reshape(-1), minlength=R).max() < 12


def test_profiling_flag_prints_per_op_times(oracle):
    """--profiling [ref: src/runtime/model.cc:2358-2362]: every op bracketed by two events and printed in the reference's
    formats [ref: src/ops/linear.cu:541,761; src/ops/concat.cu:297]; the result of the step is unchanged."""
    exe = os.path.join(ROOT, "dlrm_flexflow_amd", "host", "dlrm")
    args = ["--backend", oracle.ORACLE_LIB, "-b", "64", "--arch-sparse-feature-size", "8", "--arch-embedding-size", "100-200-50",
            "--arch-mlp-bot", "13-16-8", "--arch-mlp-top", "32-16-1", "--data-size", "64", "--epochs", "1"]
    r = subprocess.run([exe] + args + ["--profiling"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert out.count("[Linear] forward time = ") == 2 * 4          # warm-up + 1 iteration, four Dense layers
    assert out.count("Linear backward time = ") == 2 * 4
    assert "Dense_100 [Linear] forward time = " in out and "[Concat_105] forward time = " in out
    assert out.count("[Embedding x3] forward time = ") == 2 and out.count("[Embedding x3] backward time = ") == 2
    plain = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert "forward time" not in plain.stdout
    mse = lambda txt: [l for l in txt.splitlines() if "mean_squared_error" in l][-1]
    assert mse(r.stderr) == mse(plain.stderr)


def test_two_rank_hdf5_dataset_with_data_parallel_tables(tmp_path):
    """The file loader next to --replicate-embedding-rows: every rank holds the ids of a replicated table for the whole batch and
    gathers its own half; a new batch every step (the replicated gradient is computed on the compute stream from that step's
    ids).  Trajectory equals the single-rank run; both ranks end with the same copy of the two replicated tables."""
    h5, exp = H.make_criteo_like_hdf5(str(tmp_path))
    _run_ranks(2, tmp_path, "hdf5_replicated")
    app = ffmodel.DLRM(["--backend", H.oracle_backend()] + H.HDF5_ARGS + ["--dataset", h5])
    app.warmup(); app.train_steps(7, trace=False); app.model.sync()
    ref = {"w_bot": app.model.parameter(0, 0).get_weights(), "w_top": app.model.parameter(app.model.num_layers - 1, 0).get_weights()}
    for t in range(3):
        ref[f"emb{t}"] = app.model.parameter(2 + t, 0).get_weights()
    app.close()
    holders = {0: [], 1: [], 2: []}
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert np.array_equal(z["dense"], exp["X_int"][16 + 8 * r:16 + 8 * (r + 1)])
        for t in range(3):
            if f"emb{t}" in z.files:
                holders[t].append(r)
                assert np.array_equal(z[f"sparse{t}"].reshape(-1), exp["X_cat"][16:32, t])       # ids of the whole batch
                np.testing.assert_allclose(z[f"emb{t}"], ref[f"emb{t}"], rtol=1e-5, atol=1e-6, err_msg=f"rank {r} table {t}")
        np.testing.assert_allclose(z["w_top"], ref["w_top"], rtol=1e-5, atol=1e-6)
        assert int(z["alltoall_calls"]) == 2 * 8 and int(z["allreduce_calls"]) == 8     # table 2 still crosses the all-to-all
    assert holders[0] == [0, 1] and holders[1] == [0, 1] and len(holders[2]) == 1
