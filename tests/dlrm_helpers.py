"""Shared by the host-layer tests: builds the small DLRM of tests/golden/dlrm_step_torch.npz through
the Python FFModel API (the reference's ffmodel.dense / embedding / concat surface), injects the
golden weights and inputs, runs training steps and collects predictions + parameters.

`backend` is the path of a library exporting include/ff_hip.h: the HIP library for the GPU
tests, the CPU oracle (test infrastructure) for the host-logic tests that run without a GPU.
"""
import os

import numpy as np

from conftest import golden
from dlrm_flexflow_amd import capi, ffmodel


ADAM_HP = dict(alpha=0.01, beta1=0.9, beta2=0.999, weight_decay=0.0, epsilon=1e-8)
MOM_HP = dict(lr=0.05, momentum=0.9, nesterov=False, weight_decay=0.0)


def oracle_backend():
    from oracle import oracle
    oracle.build()
    return oracle.ORACLE_LIB


def build_golden_dlrm(backend, comm=None, enable_graph=False, overlap=True, dense_update=False, g=None, force_exchange=False,
                      column_shard_rows=0, extra_argv=(), adam=None, row_shard_rows=0, replicate_rows=0, sgd=None):
    """Returns (model, handles) with weights and inputs of the golden fixture loaded.
    With `comm` (world_size > 1) each rank loads its batch slice / its tables."""
    g = g or golden("dlrm_step_torch")
    B, D, L = int(g["B"]), int(g["D"]), int(g["L"])
    rows, bot, top = list(g["rows"]), list(g["bot"]), list(g["top"])
    world = comm.world_size if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    argv = ["-b", str(B)] + (["--force-exchange"] if force_exchange else [])
    if column_shard_rows:
        argv += ["--column-shard-rows", str(column_shard_rows)]
    if row_shard_rows:
        argv += ["--row-shard-rows", str(row_shard_rows)]
    if replicate_rows:
        argv += ["--replicate-embedding-rows", str(replicate_rows)]
    cfg = ffmodel.FFConfig(argv=argv + list(extra_argv), backend=backend, comm=comm)
    cfg.set(enable_graph=enable_graph, overlap_embedding=overlap, dense_embedding_update=dense_update)
    m = ffmodel.FFModel(cfg)
    sparse = [m.create_tensor([B, L], ffmodel.DT_INT64) for _ in rows]
    dense = m.create_tensor([B, bot[0]], ffmodel.DT_FLOAT)
    x = dense
    for i in range(len(bot) - 1):
        x = m.dense(x, bot[i + 1], capi.AC_MODE_RELU)
    ly = [m.embedding(s, r, D, capi.AGGR_MODE_SUM) for s, r in zip(sparse, rows)]
    z = m.concat([x] + ly, 1)
    for i in range(len(top) - 1):
        z = m.dense(z, top[i + 1], capi.AC_MODE_SIGMOID if i == len(top) - 2 else capi.AC_MODE_RELU)
    if adam is not None:
        m.set_adam_optimizer(**adam)
    elif sgd is not None:
        m.set_sgd_optimizer(**sgd)
    else:
        m.set_sgd_optimizer(lr=0.01)
    m.compile()
    m.init_layers()
    # layer order: bottom dense..., embeddings..., concat, top dense...
    nb, nt = len(bot) - 1, len(top) - 1
    layer = 0
    names = {}
    for i in range(nb):
        names[f"bot.{i}"] = layer; layer += 1
    for t in range(len(rows)):
        names[f"emb.{t}"] = layer; layer += 1
    layer += 1
    for i in range(nt):
        names[f"top.{i}"] = layer; layer += 1
    assert layer == m.num_layers
    assert m.layer_name(0) == "Dense_100" and m.layer_name(nb) == f"Embedding_{100 + nb}"   # reference naming
    for k, li in names.items():
        if k.startswith("emb"):
            p = m.parameter(li, 0)
            if p.is_local:
                w0 = g[f"init/{k}.weight"]
                cols = p.dims[1]                      # column-sharded giant table: this rank holds D/G columns of every row
                if p.dims[0] != w0.shape[0]:          # row-sharded: rows [R r / G, R (r + 1) / G)
                    R = w0.shape[0]
                    w0 = np.ascontiguousarray(w0[R * rank // world:R * (rank + 1) // world])
                    assert w0.shape[0] == p.dims[0]
                p.set_weights(w0 if cols == w0.shape[1] else np.ascontiguousarray(w0[:, rank * cols:(rank + 1) * cols]))
        else:
            m.parameter(li, 0).set_weights(g[f"init/{k}.weight"])
            m.parameter(li, 1).set_weights(g[f"init/{k}.bias"])
    Bl = B // world
    sl = slice(rank * Bl, (rank + 1) * Bl)
    dense.set(g["dense"][sl])
    m.label_tensor.set(g["label"][sl])
    for t, s in enumerate(sparse):
        if s.is_local:
            s.set(g[f"sparse{t}"])
    return m, {"names": names, "final": m.num_layers - 1, "slice": sl, "g": g}


def run_steps(m, h, steps=2, trace=False):
    """Returns per-step predictions (this rank's rows) and the parameters after each step."""
    out = []
    for _ in range(steps):
        if trace:
            m.begin_trace(7)
        m.forward(); m.zero_gradients(); m.backward(); m.update()
        if trace:
            m.end_trace(7)
        m.sync()
        rec = {"pred": m.layer_output(h["final"]).get()}
        for k, li in h["names"].items():
            p = m.parameter(li, 0)
            if p.is_local:
                rec[f"{k}.weight"] = p.get_weights()
            if not k.startswith("emb"):
                rec[f"{k}.bias"] = m.parameter(li, 1).get_weights()
        out.append(rec)
    return out


def check_against_golden(recs, h, rtol=1e-5, atol=1e-6):
    g = h["g"]
    for step, rec in enumerate(recs):
        # the forward of step k sees the parameters after step k-1: predictions of the golden step k
        np.testing.assert_allclose(rec["pred"], g[f"step{step}/pred"][h["slice"]], rtol=rtol, atol=atol, err_msg=f"pred step {step}")
        for k, v in rec.items():
            if k == "pred":
                continue
            np.testing.assert_allclose(v, g[f"step{step}/{k}"], rtol=rtol, atol=atol, err_msg=f"{k} step {step}")


def torch_optimizer_reference(g, steps, kind, lazy_tables, **hp):
    """The golden DLRM in torch (autograd for the gradients) with the reference's optimizer statements applied by hand in fp32:
    kind "adam" [ref: src/runtime/optimizer_kernel.cu:206-226, alpha_t optimizer.cc:248-254] or "sgd" (lr, momentum, nesterov,
    weight_decay) [ref: optimizer_kernel.cu:23-41].  lazy_tables: the embedding tables follow the touched-rows rule
    (--sparse-embedding-optimizer: rows a batch does not touch keep weight and state; the semantics of torch.optim.SparseAdam);
    False: the reference's dense sweep over every row.  Returns per-step records like run_steps()."""
    import torch
    rows, bot, top = list(g["rows"]), list(g["bot"]), list(g["top"])
    B = int(g["B"])
    P = {}
    for i in range(len(bot) - 1):
        P[f"bot.{i}.weight"] = g[f"init/bot.{i}.weight"]; P[f"bot.{i}.bias"] = g[f"init/bot.{i}.bias"]
    for i in range(len(top) - 1):
        P[f"top.{i}.weight"] = g[f"init/top.{i}.weight"]; P[f"top.{i}.bias"] = g[f"init/top.{i}.bias"]
    for t in range(len(rows)):
        P[f"emb.{t}.weight"] = g[f"init/emb.{t}.weight"]
    P = {k: torch.tensor(np.array(v, np.float32), requires_grad=True) for k, v in P.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    dense, label = torch.from_numpy(g["dense"]), torch.from_numpy(g["label"])
    sparse = [torch.from_numpy(g[f"sparse{t}"]) for t in range(len(rows))]
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)
    b1t = b2t = 1.0
    out = []
    for _ in range(steps):
        x = dense
        for i in range(len(bot) - 1):
            x = torch.relu(x @ P[f"bot.{i}.weight"].T + P[f"bot.{i}.bias"])
        ly = [P[f"emb.{t}.weight"][s].sum(1) for t, s in enumerate(sparse)]
        z = torch.cat([x] + ly, 1)
        for i in range(len(top) - 1):
            z = z @ P[f"top.{i}.weight"].T + P[f"top.{i}.bias"]
            z = torch.sigmoid(z) if i == len(top) - 2 else torch.relu(z)
        for v in P.values():
            v.grad = None
        (0.5 * ((z - label) ** 2).sum() / B).backward()
        if kind == "adam":
            b1t *= hp["beta1"]; b2t *= hp["beta2"]
            alpha_t = hp["alpha"] * np.sqrt(1 - b2t) / (1 - b1t)
        with torch.no_grad():
            for k, w in P.items():
                sel = slice(None)
                if lazy_tables and k.startswith("emb"):
                    sel = torch.unique(sparse[int(k.split(".")[1])].reshape(-1))
                wd = f32(hp.get("weight_decay", 0.0))
                gt = w.grad[sel] + wd * w[sel]
                if kind == "adam":
                    M[k][sel] = f32(hp["beta1"]) * M[k][sel] + (1 - f32(hp["beta1"])) * gt
                    V[k][sel] = f32(hp["beta2"]) * V[k][sel] + (1 - f32(hp["beta2"])) * gt * gt
                    w[sel] -= f32(alpha_t) * M[k][sel] / (torch.sqrt(V[k][sel]) + f32(hp["epsilon"]))
                else:
                    if hp.get("momentum", 0.0) > 0:
                        V[k][sel] = V[k][sel] * f32(hp["momentum"]) + gt
                        gt = gt + f32(hp["momentum"]) * V[k][sel] if hp.get("nesterov") else V[k][sel]
                    w[sel] -= f32(hp["lr"]) * gt
        rec = {"pred": z.detach().numpy().copy()}
        rec.update({k: v.detach().numpy().copy() for k, v in P.items()})
        out.append(rec)
    return out


def torch_adam_reference(g, steps, alpha=0.001, beta1=0.9, beta2=0.999, weight_decay=0.0, epsilon=1e-8):
    """The golden DLRM in torch (autograd for the gradients) with the reference's Adam applied by hand, statement by
    statement in fp32 [ref: src/runtime/optimizer_kernel.cu:206-226; alpha_t: src/runtime/optimizer.cc:248-254] --
    torch.optim.Adam places epsilon differently, which matters for the small gradients of this model.
    Returns per-step {"pred", "<name>.weight", "<name>.bias"} like run_steps()."""
    import torch
    rows, bot, top = list(g["rows"]), list(g["bot"]), list(g["top"])
    B = int(g["B"])
    P = {}
    for i in range(len(bot) - 1):
        P[f"bot.{i}.weight"] = g[f"init/bot.{i}.weight"]; P[f"bot.{i}.bias"] = g[f"init/bot.{i}.bias"]
    for i in range(len(top) - 1):
        P[f"top.{i}.weight"] = g[f"init/top.{i}.weight"]; P[f"top.{i}.bias"] = g[f"init/top.{i}.bias"]
    for t in range(len(rows)):
        P[f"emb.{t}.weight"] = g[f"init/emb.{t}.weight"]
    P = {k: torch.tensor(np.array(v, np.float32), requires_grad=True) for k, v in P.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    dense, label = torch.from_numpy(g["dense"]), torch.from_numpy(g["label"])
    sparse = [torch.from_numpy(g[f"sparse{t}"]) for t in range(len(rows))]
    b1t = b2t = 1.0
    out = []
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)
    for _ in range(steps):
        x = dense
        for i in range(len(bot) - 1):
            x = torch.relu(x @ P[f"bot.{i}.weight"].T + P[f"bot.{i}.bias"])
        ly = [P[f"emb.{t}.weight"][s].sum(1) for t, s in enumerate(sparse)]
        z = torch.cat([x] + ly, 1)
        for i in range(len(top) - 1):
            z = z @ P[f"top.{i}.weight"].T + P[f"top.{i}.bias"]
            z = torch.sigmoid(z) if i == len(top) - 2 else torch.relu(z)
        for v in P.values():
            v.grad = None
        (0.5 * ((z - label) ** 2).sum() / B).backward()
        b1t *= beta1; b2t *= beta2
        alpha_t = alpha * np.sqrt(1 - b2t) / (1 - b1t)
        with torch.no_grad():
            for k, w in P.items():
                gt = w.grad + f32(weight_decay) * w
                M[k] = f32(beta1) * M[k] + (1 - f32(beta1)) * gt
                V[k] = f32(beta2) * V[k] + (1 - f32(beta2)) * gt * gt
                w -= f32(alpha_t) * M[k] / (torch.sqrt(V[k]) + f32(epsilon))
        rec = {"pred": z.detach().numpy().copy()}
        rec.update({k: v.detach().numpy().copy() for k, v in P.items()})
        out.append(rec)
    return out


def make_criteo_like_hdf5(tmp_path, n=100, rows=(50, 7, 300), dense=13, seed=5, bad_id=False):
    """A Criteo-shaped npz (raw integer counts, categorical ids, 0/1 labels) converted by tools/preprocess_hdf.py --
    the counterpart of the reference's examples/cpp/DLRM/preprocess_hdf.py.  Returns (hdf5 path, the arrays the loader
    must deliver: X_int after log(x+1), X_cat as int64, y as float32)."""
    import importlib.util
    rng = np.random.default_rng(seed)
    raw = {"X_int": rng.integers(0, 1000, (n, dense)).astype(np.int32),
           "X_cat": np.stack([rng.integers(0, r, n) for r in rows], 1).astype(np.int32),
           "y": rng.integers(0, 2, n).astype(np.int32)}
    if bad_id:
        raw["X_cat"][n // 2, 1] = rows[1]
    npz, h5 = os.path.join(tmp_path, "day.npz"), os.path.join(tmp_path, "day.h5")
    np.savez(npz, **raw)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("preprocess_hdf", os.path.join(root, "tools", "preprocess_hdf.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    mod.convert(npz, h5)
    return h5, {"X_int": np.log(raw["X_int"].astype(np.float32) + 1), "X_cat": raw["X_cat"].astype(np.int64), "y": raw["y"].astype(np.float32)}


HDF5_ARGS = ["-b", "16", "--arch-sparse-feature-size", "8", "--arch-embedding-size", "50-7-300", "--arch-mlp-bot", "13-16-8",
             "--arch-mlp-top", "32-16-1"]


# dot interaction through the driver: 5 tables + the bottom output = 6 vectors of 8 -> 8 + 15 = 23 inputs of the top MLP
DOT_ARGS = ["-b", "32", "--arch-sparse-feature-size", "8", "--arch-embedding-size", "50-7-300-11-23", "--arch-mlp-bot", "13-16-8",
            "--arch-mlp-top", "23-16-1", "--data-size", "32", "--embedding-bag-size", "2"]


KAGGLE_ROWS = "1396-550-1761917-507795-290-21-11948-608-3-58176-5237-1497287-3127-26-12153-1068715-10-4836-2085-4-1312273-17-15-110946-91-72655"


def KAGGLE_ARGS(global_batch, backend=None):
    """Driver flags of BASELINE configs[1] (Criteo-Kaggle shape) at the given global batch."""
    a = ["-b", str(global_batch), "--arch-sparse-feature-size", "16", "--arch-embedding-size", KAGGLE_ROWS,
         "--arch-mlp-bot", "13-512-256-64-16", "--arch-mlp-top", "432-512-256-1", "--data-size", str(global_batch)]
    return (["--backend", backend] if backend else []) + a
