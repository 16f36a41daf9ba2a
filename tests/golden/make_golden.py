#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the BUILD container only).

Sources of truth, in the order SURVEY.md section 8c ranks them:
  1. embedding forward  : the reference's own EmbeddingLookup_int64_t_float_float__avx2_fma,
                          compiled from /root/reference/src/ops/embedding.cc by oracle/Makefile
                          into oracle/_ref/libref_embedding.so (needs /root/reference).
  2. Linear / Concat / BatchMatmul / SGD / MSE / one whole DLRM step : PyTorch-CPU + numpy,
                          the oracle the reference's own op tests use
                          ([ref: tests/ops/test_harness.py:201-283,300-405,425-510]); same seeds
                          (np.random.seed(0)) and the reference's small known-answer shapes.
The fixtures are plain arrays (inputs + expected outputs) in .npz files; no reference
source text is stored.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path)} bytes")


def embedding_from_reference():
    assert oracle.ref_available(), "run `make -C oracle ref` with /root/reference present"
    rng = np.random.default_rng(0)
    cases = {}
    k = 0
    # D: the three unrolled AVX2 branches (128, 64, 32), 16 (generic, vector only),
    # 256 (generic), 13 and 20 (generic with scalar tail)
    for D in (128, 64, 32, 16, 256, 13, 20):
        for L in (1, 3):
            R, B = 97, 24
            w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
            w[3, :] = -0.0          # sign of zero: 0 + (-0) = +0
            idx = rng.integers(0, R, (B, L)).astype(np.int64)
            idx[0, 0] = 3
            idx[1, :] = idx[0, :]   # duplicate bags
            out = oracle.ref_embedding_fwd(idx, w)
            cases[f"c{k}_w"] = w
            cases[f"c{k}_idx"] = idx
            cases[f"c{k}_out"] = out
            k += 1
    # ragged bags (lengths differ, incl. empty) -- the reference signature allows it
    D, R, B = 32, 50, 10
    w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
    lengths = np.array([0, 1, 2, 3, 0, 5, 1, 1, 4, 2], np.int32)
    flat = rng.integers(0, R, (int(lengths.sum()), 1)).astype(np.int64)
    # call the compiled function directly for the ragged case
    import ctypes as C
    lib_ = C.CDLL(oracle.REF_LIB)
    fn = getattr(lib_, oracle._REF_LOOKUP)
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_bool, C.c_void_p]
    out = np.empty((B, D), np.float32)
    fl = np.ascontiguousarray(flat.reshape(-1))
    fn(D, B, fl.size, R, w.ctypes.data, fl.ctypes.data, lengths.ctypes.data, None, False, out.ctypes.data)
    cases["ragged_w"], cases["ragged_idx"], cases["ragged_len"], cases["ragged_out"] = w, fl, lengths, out
    out_mean = np.empty((B, D), np.float32)
    fn(D, B, fl.size, R, w.ctypes.data, fl.ctypes.data, lengths.ctypes.data, None, True, out_mean.ctypes.data)
    cases["ragged_out_mean"] = out_mean
    cases["n_cases"] = np.array(k)
    save("embedding_fwd_ref", **cases)


def embedding_bwd_from_reference():
    """embed_backward of the reference's CPU path [ref: src/ops/embedding.cc:344-374], compiled from its source: one index
    per sample, gradients accumulated on top of an existing dense table gradient, heavy duplicates included."""
    assert oracle.ref_available(), "run `make -C oracle ref` with /root/reference present"
    rng = np.random.default_rng(7)
    cases = {}
    k = 0
    for (B, R, D) in ((200, 50, 16), (333, 7, 13), (64, 1000, 128), (1, 3, 4), (4096, 3, 16)):
        idx = rng.integers(0, R, (B, 1)).astype(np.int64)
        g = rng.uniform(-1, 1, (B, D)).astype(np.float32)
        wg0 = rng.uniform(-1, 1, (R, D)).astype(np.float32)
        cases[f"c{k}_idx"], cases[f"c{k}_g"], cases[f"c{k}_wg0"] = idx, g, wg0
        cases[f"c{k}_wg"] = oracle.ref_embedding_bwd(idx, g, wg0)
        k += 1
    cases["n_cases"] = np.array(k)
    save("embedding_bwd_ref", **cases)


def linear_from_torch():
    np.random.seed(0)
    cases = {}
    k = 0
    # reference LinearTest shapes are (2,2,3), (10,2000,1000), (20,5000,5000); the first is
    # stored, small stand-ins replace the big two (they are regenerated from the seed at test time)
    for (B, IN, OUT, act) in ((2, 2, 3, "none"), (10, 40, 24, "none"), (16, 13, 32, "relu"), (9, 33, 1, "sigmoid"), (32, 64, 48, "relu")):
        x = np.random.uniform(-1, 1, (B, IN)).astype(np.float32)
        w = np.random.uniform(-1, 1, (OUT, IN)).astype(np.float32)
        b = np.random.uniform(-1, 1, (OUT,)).astype(np.float32) if k else np.zeros(OUT, np.float32)
        gy = np.random.uniform(-1, 1, (B, OUT)).astype(np.float32)
        lin = torch.nn.Linear(IN, OUT)
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(w))
            lin.bias.copy_(torch.from_numpy(b))
        xt = torch.from_numpy(x).requires_grad_(True)
        opt = torch.optim.SGD(lin.parameters(), lr=0.01)
        y = lin(xt)
        if act == "relu":
            y = torch.relu(y)
        elif act == "sigmoid":
            y = torch.sigmoid(y)
        y.backward(torch.from_numpy(gy))
        dw = lin.weight.grad.numpy().copy()
        db = lin.bias.grad.numpy().copy()
        dx = xt.grad.numpy().copy()
        opt.step()
        cases.update({f"c{k}_x": x, f"c{k}_w": w, f"c{k}_b": b, f"c{k}_gy": gy,
                      f"c{k}_y": y.detach().numpy(), f"c{k}_dw": dw, f"c{k}_db": db, f"c{k}_dx": dx,
                      f"c{k}_w_after": lin.weight.detach().numpy().copy(),
                      f"c{k}_b_after": lin.bias.detach().numpy().copy(),
                      f"c{k}_act": np.array({"none": 10, "relu": 11, "sigmoid": 12}[act])})
        k += 1
    cases["n_cases"] = np.array(k)
    save("linear_torch", **cases)


def concat_from_numpy():
    np.random.seed(0)
    cases = {}
    k = 0
    # reference ConcatTest shapes (batch, i_dim, channels) [ref: tests/ops/test_harness.py:300-405]
    for (B, W, N) in ((2, 6, 4), (2, 6, 2), (4, 6, 2), (4, 10, 10)):
        parts = [np.random.uniform(-1, 1, (B, W)).astype(np.float32) for _ in range(N)]
        cases[f"c{k}_n"] = np.array(N)
        for i, p in enumerate(parts):
            cases[f"c{k}_in{i}"] = p
        cases[f"c{k}_out"] = np.concatenate(parts, axis=1)
        k += 1
    # ragged widths, DLRM-like (bottom output + tables)
    parts = [np.random.uniform(-1, 1, (5, w)).astype(np.float32) for w in (3, 4, 4, 1, 7)]
    cases[f"c{k}_n"] = np.array(len(parts))
    for i, p in enumerate(parts):
        cases[f"c{k}_in{i}"] = p
    cases[f"c{k}_out"] = np.concatenate(parts, axis=1)
    k += 1
    cases["n_cases"] = np.array(k)
    save("concat_numpy", **cases)


def bmm_from_torch():
    np.random.seed(0)
    cases = {}
    k = 0
    # (d,m,n,k) [ref: tests/ops/test_harness.py:425-510]; (145,265,15,64) is regenerated at test time
    for (d, m, n, kk) in ((1, 2, 3, 4), (5, 2, 3, 4), (2, 1, 1, 1), (2, 2, 2, 1), (7, 27, 27, 16)):
        a = np.random.uniform(0, 1, (d, n, kk)).astype(np.float32)
        b = np.random.uniform(0, 1, (d, kk, m)).astype(np.float32)
        go = np.random.uniform(0, 1, (d, n, m)).astype(np.float32)
        at = torch.from_numpy(a).requires_grad_(True)
        bt = torch.from_numpy(b).requires_grad_(True)
        o = torch.matmul(at, bt)
        o.backward(torch.from_numpy(go))
        cases.update({f"c{k}_a": a, f"c{k}_b": b, f"c{k}_go": go, f"c{k}_o": o.detach().numpy(),
                      f"c{k}_ga": at.grad.numpy().copy(), f"c{k}_gb": bt.grad.numpy().copy()})
        k += 1
    cases["n_cases"] = np.array(k)
    save("bmm_torch", **cases)


def sgd_mse_from_torch():
    np.random.seed(0)
    cases = {}
    k = 0
    for (lr, wd, mom, nest) in ((0.01, 0.0, 0.0, False), (0.01, 1e-4, 0.0, False), (0.05, 1e-4, 0.9, False), (0.05, 0.0, 0.9, True)):
        w0 = np.random.uniform(-1, 1, (257,)).astype(np.float32)
        p = torch.nn.Parameter(torch.from_numpy(w0.copy()))
        opt = torch.optim.SGD([p], lr=lr, weight_decay=wd, momentum=mom, nesterov=nest)
        grads = []
        for step in range(3):
            g = np.random.uniform(-1, 1, (257,)).astype(np.float32)
            grads.append(g)
            p.grad = torch.from_numpy(g.copy())
            opt.step()
        cases.update({f"c{k}_w0": w0, f"c{k}_g": np.stack(grads), f"c{k}_w3": p.detach().numpy().copy(),
                      f"c{k}_hp": np.array([lr, wd, mom, float(nest)], np.float64)})
        k += 1
    cases["n_cases"] = np.array(k)
    # MSE "avg reduce" backward: (p - y)/B, no factor 2 [ref: src/loss_functions/loss_functions.cu:65-76,202]
    pz = np.random.uniform(0, 1, (37, 1)).astype(np.float32)
    yz = (np.random.uniform(0, 1, (37, 1)) > 0.5).astype(np.float32)
    pt = torch.from_numpy(pz).requires_grad_(True)
    loss = 0.5 * ((pt - torch.from_numpy(yz)) ** 2).sum() / 37
    loss.backward()
    cases.update({"mse_p": pz, "mse_y": yz, "mse_grad": pt.grad.numpy().copy(),
                  "mse_sum": np.array(((pz.astype(np.float64) - yz) ** 2).sum())})
    save("sgd_mse_torch", **cases)


def adam_from_torch():
    """torch.optim.Adam (L2 weight decay folded into the gradient, like the reference's adam_update
    [ref: src/runtime/optimizer_kernel.cu:206-226]).  torch places epsilon after the bias correction of sqrt(v), the
    reference before it: with |g| ~ 0.1..1 the two differ by ~1e-8 relative, far below the 1e-5 of the comparison."""
    np.random.seed(3)
    cases = {}
    k = 0
    for (alpha, b1, b2, wd, eps) in ((0.001, 0.9, 0.999, 0.0, 1e-8), (0.01, 0.9, 0.999, 1e-4, 1e-8), (0.003, 0.8, 0.99, 0.0, 1e-8)):
        w0 = np.random.uniform(-1, 1, (515,)).astype(np.float32)
        p = torch.nn.Parameter(torch.from_numpy(w0.copy()))
        opt = torch.optim.Adam([p], lr=alpha, betas=(b1, b2), eps=eps, weight_decay=wd)
        grads = []
        for step in range(5):
            g = (np.random.uniform(0.1, 1, (515,)) * np.random.choice([-1.0, 1.0], (515,))).astype(np.float32)
            grads.append(g)
            p.grad = torch.from_numpy(g.copy())
            opt.step()
        cases.update({f"c{k}_w0": w0, f"c{k}_g": np.stack(grads), f"c{k}_w5": p.detach().numpy().copy(),
                      f"c{k}_hp": np.array([alpha, b1, b2, wd, eps], np.float64)})
        k += 1
    cases["n_cases"] = np.array(k)
    save("adam_torch", **cases)


class TorchDLRM(torch.nn.Module):
    """The topology the reference driver builds [ref: examples/cpp/DLRM/dlrm.cc:26-65,97-128]:
    bottom MLP (all ReLU), T EmbeddingBag(sum), concat [x, e_0..e_{T-1}], top MLP (ReLU, last sigmoid)."""

    def __init__(self, bot, top, rows, D):
        super().__init__()
        self.bot = torch.nn.ModuleList([torch.nn.Linear(bot[i], bot[i + 1]) for i in range(len(bot) - 1)])
        self.emb = torch.nn.ModuleList([torch.nn.EmbeddingBag(r, D, mode="sum") for r in rows])
        self.top = torch.nn.ModuleList([torch.nn.Linear(top[i], top[i + 1]) for i in range(len(top) - 1)])

    def forward(self, dense, sparse):
        x = dense
        for l in self.bot:
            x = torch.relu(l(x))
        ly = [e(s) for e, s in zip(self.emb, sparse)]
        z = torch.cat([x] + ly, dim=1)
        for i, l in enumerate(self.top):
            z = l(z)
            z = torch.sigmoid(z) if i == len(self.top) - 1 else torch.relu(z)
        return z


def dlrm_step_from_torch():
    torch.manual_seed(0)
    np.random.seed(0)
    B, D, L = 16, 8, 2
    rows = [7, 50, 3, 20]
    bot = [5, 16, D]
    top = [D + len(rows) * D, 24, 1]
    model = TorchDLRM(bot, top, rows, D)
    with torch.no_grad():
        for e, r in zip(model.emb, rows):
            e.weight.uniform_(-(1.0 / r) ** 0.5, (1.0 / r) ** 0.5)
    dense = np.random.uniform(0, 1, (B, bot[0])).astype(np.float32)
    sparse = [np.random.randint(0, r, (B, L)).astype(np.int64) for r in rows]
    label = (np.random.uniform(0, 1, (B, 1)) > 0.5).astype(np.float32)
    arrs = {"B": np.array(B), "D": np.array(D), "L": np.array(L), "rows": np.array(rows),
            "bot": np.array(bot), "top": np.array(top), "dense": dense, "label": label}
    for t, s in enumerate(sparse):
        arrs[f"sparse{t}"] = s
    for name, p in model.named_parameters():
        arrs["init/" + name] = p.detach().numpy().copy()
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    for step in range(2):
        opt.zero_grad()
        p = model(torch.from_numpy(dense), [torch.from_numpy(s) for s in sparse])
        loss = 0.5 * ((p - torch.from_numpy(label)) ** 2).sum() / B
        loss.backward()
        arrs[f"step{step}/pred"] = p.detach().numpy().copy()
        arrs[f"step{step}/mse_sum"] = np.array(((p.detach().numpy().astype(np.float64) - label) ** 2).sum())
        opt.step()
        for name, q in model.named_parameters():
            arrs[f"step{step}/" + name] = q.detach().numpy().copy()
    save("dlrm_step_torch", **arrs)


if __name__ == "__main__":
    embedding_from_reference()
    embedding_bwd_from_reference()
    linear_from_torch()
    concat_from_numpy()
    bmm_from_torch()
    sgd_mse_from_torch()
    adam_from_torch()
    dlrm_step_from_torch()
