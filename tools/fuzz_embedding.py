#!/usr/bin/env python3
"""Random check of the batched embedding kernels on the GPU against the CPU oracle (test infrastructure): table counts, row
counts from 1 to a few hundred thousand, bag sizes, row widths (vector and scalar paths), batches on both sides of the
small-batch kernel's limit, SUM / AVG, and id distributions from uniform to "every lookup hits one row".  The gather and the
fused sparse update are compared bit for bit (the update's order is canonical, include/ff_hip.h); columns of the shared
buffers that belong to nobody must stay untouched.  Usage: tools/fuzz_embedding.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlrm_flexflow_amd import capi
from oracle import oracle
oracle.build()
hip = capi.load_hip(0)
workspace = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")      # FFHandler.workSpace analogue
hip.set_workspace(workspace, workspace.numel())
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bits = lambda a, b: np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def draw_ids(R, B, L):
    kind = rng.integers(0, 5)
    if kind == 0 or R == 1: return rng.integers(0, R, (B, L))
    if kind == 1: return np.full((B, L), int(rng.integers(0, R)))                       # one hot row
    if kind == 2: return rng.integers(0, min(R, 3), (B, L))                            # three rows share everything
    if kind == 3:                                                                      # power law
        r = np.minimum((rng.pareto(1.1, (B, L)) * 2).astype(np.int64), R - 1)
        return (r * 2654435761) % R
    return np.minimum(rng.integers(0, R, (B, L)), rng.integers(0, R, (B, L)))          # skewed towards the low rows


for case in range(ncases):
    T = int(rng.integers(1, 13))
    D = int(rng.choice([4, 8, 16, 32, 64, 128, 256, 13, 20, 1, 36]))
    L = int(rng.choice([1, 1, 1, 2, 3, 4]))
    B = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 2049)), int(rng.integers(2049, 9000)), 2048 // L, 2048 // L + 1]))
    aggr = int(rng.choice([capi.AGGR_MODE_SUM, capi.AGGR_MODE_AVG]))
    lr = float(rng.choice([0.01, 0.05, 1.0]))
    lead = int(rng.choice([0, 4, 16])) if D % 4 == 0 else int(rng.integers(0, 5))
    ld = lead + T * D + int(rng.choice([0, 4])) * (D % 4 == 0)
    rows = [int(rng.choice([1, 2, 3, int(rng.integers(4, 300)), int(rng.integers(300, 20000)), int(rng.integers(20000, 300000))])) for _ in range(T)]
    Z = torch.full((B, ld), -5.0, device="cuda")
    G = rng.uniform(-1, 1, (B, ld)).astype(np.float32)
    Gt = dev(G)
    idxs, ws, wt, fe, be = [], [], [], [], []
    for t, R in enumerate(rows):
        w = rng.uniform(-1, 1, (R, D)).astype(np.float32)
        idx = draw_ids(R, B, L)
        idxs.append(idx); ws.append(w); wt.append(dev(w))
        it = dev(idx)
        fe.append((it, wt[-1], Z[:, lead + t * D:], R, ld))
        be.append((it, wt[-1], Gt[:, lead + t * D:], R, ld))
    what = f"case {case}: T {T} D {D} L {L} B {B} aggr {aggr} rows {rows} ld {ld} lead {lead}"
    arr = hip.emb_tables(fe)
    hip.check(hip.lib.ffh_embedding_fwd_multi(hip.ctx, arr, T, L, D, B, aggr, None), "fwd_multi " + what)
    z = Z.cpu().numpy()
    for t in range(T):
        assert bits(z[:, lead + t * D:lead + (t + 1) * D], oracle.embedding_fwd(idxs[t], ws[t], aggr)), what + f": gather of table {t}"
    assert (z[:, :lead] == -5.0).all() and (z[:, lead + T * D:] == -5.0).all(), what + ": gather wrote outside its columns"
    arr = hip.emb_tables(be)
    hip.check(hip.lib.ffh_embedding_bwd_sgd_fused_multi(hip.ctx, arr, T, L, D, B, aggr, lr, None), "fused_multi " + what)
    torch.cuda.synchronize()
    for t in range(T):
        gs = np.ascontiguousarray(G[:, lead + t * D:lead + (t + 1) * D])
        assert bits(wt[t].cpu().numpy(), oracle.embedding_bwd_sgd_fused(idxs[t], gs, ws[t], lr, aggr)), what + f": update of table {t}"
    assert bits(Gt.cpu().numpy(), G), what + ": the update modified the gradient buffer"
print(f"fuzz_embedding: {ncases} random cases agree with the oracle bit for bit")
