"""DLRM with the "dot" interaction built through the driver flags, checked against a torch model of
the same composition (cat -> reshape -> transpose -> bmm -> flat -> cat with the bottom output)."""
import numpy as np
import torch

from dlrm_flexflow_amd import ffmodel


class TorchDotDLRM(torch.nn.Module):
    def __init__(self, bot, top, rows, D, tril=False):
        super().__init__()
        self.tril = tril
        self.bot = torch.nn.ModuleList([torch.nn.Linear(bot[i], bot[i + 1]) for i in range(len(bot) - 1)])
        self.emb = torch.nn.ModuleList([torch.nn.EmbeddingBag(r, D, mode="sum") for r in rows])
        self.top = torch.nn.ModuleList([torch.nn.Linear(top[i], top[i + 1]) for i in range(len(top) - 1)])

    def forward(self, dense, sparse):
        x = dense
        for l in self.bot:
            x = torch.relu(l(x))
        ly = [e(s) for e, s in zip(self.emb, sparse)]
        cat = torch.cat([x] + ly, dim=1)
        z = cat.reshape(x.shape[0], 1 + len(ly), x.shape[1])
        p = torch.bmm(z, z.transpose(1, 2))
        if self.tril:                                  # facebookresearch/dlrm's interact_features: Z[:, li, lj], i > j
            li, lj = torch.tril_indices(p.shape[1], p.shape[2], offset=-1)
            out = torch.cat([x, p[:, li, lj]], dim=1)
        else:
            out = torch.cat([x, p.flatten(1, 2)], dim=1)
        for i, l in enumerate(self.top):
            out = l(out)
            out = torch.sigmoid(out) if i == len(self.top) - 1 else torch.relu(out)
        return out


def run_dot_dlrm(backend, steps=2, trace=False, B=24, D=8, rows=(11, 40, 5), bot=(6, 16, 8), tril=False, fused=False):
    C = 1 + len(rows)
    top = (D + (C * (C - 1) // 2 if tril else C * C), 20, 1)
    args = ["--backend", backend, "-b", str(B), "--arch-sparse-feature-size", str(D), "--arch-embedding-size",
            "-".join(map(str, rows)), "--arch-mlp-bot", "-".join(map(str, bot)), "--arch-mlp-top", "-".join(map(str, top)),
            "--arch-interaction-op", ("dot-tril" if fused else "dot-tril-ops") if tril else "dot", "--data-size", str(B), "--embedding-bag-size", "2"]
    app = ffmodel.DLRM(args)
    m = app.model
    # read the seeded state out of the shim and mirror it in torch
    nb, T = len(bot) - 1, len(rows)
    names = [m.layer_name(i) for i in range(m.num_layers)]
    dense_layers = [i for i, n in enumerate(names) if n.startswith("Dense")]
    emb_layers = [i for i, n in enumerate(names) if n.startswith("Embedding")]
    assert [n.split("_")[0] for n in names] == ["Dense"] * nb + ["Embedding"] * T + (["Concat", "DotInteraction"] if fused else ["Concat", "Reshape", "Transpose", "BatchMatmul", "Tril" if tril else "Flat", "Concat"]) + ["Dense"] * 2
    tm = TorchDotDLRM(bot, top, rows, D, tril=tril)
    with torch.no_grad():
        for k, li in enumerate(dense_layers):
            lin = tm.bot[k] if k < nb else tm.top[k - nb]
            lin.weight.copy_(torch.from_numpy(m.parameter(li, 0).get_weights()))
            lin.bias.copy_(torch.from_numpy(m.parameter(li, 1).get_weights()))
        for t, li in enumerate(emb_layers):
            tm.emb[t].weight.copy_(torch.from_numpy(m.parameter(li, 0).get_weights()))
    app.warmup()                       # loads the batch and runs one step
    dense = app.dense_input().get()
    sparse = [app.sparse_input(t).get(np.int64) for t in range(T)]
    label = m.label_tensor.get()
    opt = torch.optim.SGD(tm.parameters(), lr=0.01)
    out = []
    for step in range(steps + 1):      # step 0 mirrors the warm-up
        opt.zero_grad()
        p = tm(torch.from_numpy(dense), [torch.from_numpy(s) for s in sparse])
        (0.5 * ((p - torch.from_numpy(label)) ** 2).sum() / B).backward()
        opt.step()
        # the shim's most recent forward (warm-up for step 0, else the previous iteration's train step) saw the
        # same parameters as this torch forward
        m.sync()
        out.append(({"pred": m.layer_output(m.num_layers - 1).get()}, {"pred": p.detach().numpy()}))
        if step < steps:
            app.train_steps(1, trace=trace)
            m.sync()
    final_got = {names[li] + "/w": m.parameter(li, 0).get_weights() for li in dense_layers + emb_layers}
    # the shim ran warm-up + `steps` steps = steps+1 updates; so did torch
    final_exp = {}
    for k, li in enumerate(dense_layers):
        lin = tm.bot[k] if k < nb else tm.top[k - nb]
        final_exp[names[li] + "/w"] = lin.weight.detach().numpy()
    for t, li in enumerate(emb_layers):
        final_exp[names[li] + "/w"] = tm.emb[t].weight.detach().numpy()
    app.close()
    return out, final_got, final_exp
