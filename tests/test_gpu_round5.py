"""GPU tests (-m gpu) added in round 5.

  * ffh_mlp_chain_fwd / _bwd (ABI 12: a chain of narrow Linear layers as one launch forward, two backward) against the oracle's
    PER-LAYER calls at 1e-5 of the term mass: the DLRM chains (Terabyte bottom MLP, Kaggle bottom / top), ragged batches, 32-row
    blocks, outputs that are column slices of wider buffers, a live / premasked top gradient, stored / accumulated / masked dx.
"""
import numpy as np
import pytest
import torch

from dlrm_flexflow_amd import capi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RELU, NONE, SIG = capi.AC_MODE_RELU, capi.AC_MODE_NONE, capi.AC_MODE_SIGMOID


def _close(got, exp, mass, what, tol=1e-5):
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    bad = err > tol * mass + 1e-6
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} beyond {tol} of the term mass, worst {err.max():.3e} (mass there {mass.flat[err.argmax()]:.3e})"


def _route(hip):
    return hip.lib.ffh_linear_last_route(hip.ctx).decode()


def _chain_params(rng, widths):
    ws, bs = [], []
    for i, o in zip(widths[:-1], widths[1:]):
        ws.append((rng.uniform(-1, 1, (o, i)) * np.sqrt(3.0 / i)).astype(np.float32))
        bs.append(rng.uniform(-0.2, 0.2, o).astype(np.float32))
    return ws, bs


CHAINS = [
    # widths, activations, batch, leading dimension of the last output (0: dense)
    ((13, 512, 256, 128), (RELU, RELU, RELU), 4096, 3456),      # Terabyte bottom MLP into the concat buffer, per-rank batch
    ((13, 512, 256, 128), (RELU, RELU, RELU), 1000, 0),         # ragged batch: the last block holds 8 rows
    ((13, 512, 256, 128), (RELU, RELU, RELU), 8192, 0),         # 32-row blocks
    ((13, 512, 256, 64, 16), (RELU, RELU, RELU, RELU), 2048, 432),   # Kaggle bottom MLP
    ((432, 512, 256, 1), (RELU, RELU, SIG), 2048, 0),           # Kaggle top MLP: a one-column last layer, sigmoid
    ((16, 48, 20), (NONE, RELU), 77, 0),                        # widths that are not multiples of 16 / 64
    ((512, 512, 512), (RELU, NONE), 300, 0),                    # the widest the chain takes
]


@pytest.mark.parametrize("widths,acts,B,ld_last", CHAINS)
def test_mlp_chain_forward_vs_oracle_per_layer(hip, oracle, widths, acts, B, ld_last):
    rng = np.random.default_rng(sum(widths) + B)
    ws, bs = _chain_params(rng, widths)
    x = rng.uniform(0, 1, (B, widths[0])).astype(np.float32)
    n = len(ws)
    ys_e, masses, cur = [], [], x
    for l in range(n):
        masses.append(np.abs(cur).astype(np.float64) @ np.abs(ws[l]).astype(np.float64).T + np.abs(bs[l]))
        cur = oracle.linear_fwd(cur, ws[l], bs[l] if l != 1 else None, acts[l])     # (layer 1 without a bias)
        ys_e.append(cur)
    xd = torch.from_numpy(x).to(DEV)
    wd = [torch.from_numpy(w).to(DEV) for w in ws]
    bd = [torch.from_numpy(b).to(DEV) for b in bs]
    yd = []
    for l in range(n):
        ld = ld_last if (l == n - 1 and ld_last) else widths[l + 1]
        yd.append(torch.full((B, ld), 7.0, device=DEV))
    off = 4 if ld_last else 0          # a column slice that starts 16 bytes into the row
    layers = hip.chain_layers([dict(w=wd[l], bias=bd[l] if l != 1 else None, y=(yd[l][:, off:] if l == n - 1 else yd[l]), ldy=yd[l].shape[1],
                                    in_dim=widths[l], out_dim=widths[l + 1], activation=acts[l]) for l in range(n)])
    for rep in range(2):
        hip.check(hip.lib.ffh_mlp_chain_fwd(hip.ctx, capi.ptr(xd), widths[0], layers, n, B, None), "chain fwd")
    assert "mlp_chain_fwd" in _route(hip), _route(hip)
    torch.cuda.synchronize()
    for l in range(n):
        got = yd[l].cpu().numpy()
        g = got[:, off:off + widths[l + 1]] if l == n - 1 else got
        _close(g, ys_e[l], masses[l], f"chain {widths} y{l}")
        if l == n - 1 and ld_last:       # nothing outside the slice was touched
            assert (got[:, :off] == 7.0).all() and (got[:, off + widths[l + 1]:] == 7.0).all()


BWD_CHAINS = [
    # widths, activations, batch, top dy premasked, dx wanted, dx flags
    ((13, 512, 256, 128), (RELU, RELU, RELU), 4096, False, False, 0),          # Terabyte bottom MLP: live relu' at the top, dx discarded
    ((13, 512, 256, 128), (RELU, RELU, RELU), 1000, True, False, 0),
    ((13, 512, 256, 128), (RELU, RELU, RELU), 8192, False, False, 0),          # 32-row blocks
    ((64, 512, 256, 64, 16), (RELU, NONE, RELU, RELU), 2048, False, True, capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X),
    ((432, 512, 256), (RELU, RELU), 2048, True, True, capi.LINEAR_DX_OVERWRITE),      # Kaggle top MLP below the click layer: dx = the concat gradient
    ((432, 512, 256), (RELU, SIG), 333, False, True, 0),                       # sigmoid at the top, dx accumulated
    ((16, 48, 20), (NONE, RELU), 77, False, True, capi.LINEAR_DX_OVERWRITE),
]


@pytest.mark.parametrize("widths,acts,B,premasked,want_dx,dxflags", BWD_CHAINS)
def test_mlp_chain_backward_vs_oracle_per_layer(hip, oracle, widths, acts, B, premasked, want_dx, dxflags):
    rng = np.random.default_rng(sum(widths) + B + 1)
    ws, bs = _chain_params(rng, widths)
    n = len(ws)
    x = np.maximum(rng.uniform(-1, 1, (B, widths[0])), 0).astype(np.float32)        # a ReLU output (for DX_MASK_BY_X)
    ys, cur = [], x
    for l in range(n):
        cur = oracle.linear_fwd(cur, ws[l], bs[l], acts[l])
        ys.append(cur)
    g_top = (rng.uniform(-1, 1, (B, widths[-1])) / B).astype(np.float32)
    if premasked:
        g_top = np.where(ys[-1] > 0, g_top, 0).astype(np.float32) if acts[-1] == RELU else g_top
    dx0 = rng.uniform(-1, 1, (B, widths[0])).astype(np.float32)
    # ---- expectation: the per-layer calls the header states, on the oracle
    exp_dw, exp_db, exp_dy, mass_dw, mass_db, mass_dx = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
    dy = g_top
    exp_dx = None
    for l in range(n - 1, -1, -1):
        xin = x if l == 0 else ys[l - 1]
        f = 0
        if l == n - 1:
            f |= capi.LINEAR_DY_PREMASKED if premasked else 0
        elif acts[l] == RELU:
            f |= capi.LINEAR_DY_PREMASKED
        if l == 0:
            f |= dxflags
        else:
            f |= capi.LINEAR_DX_OVERWRITE | (capi.LINEAR_DX_MASK_BY_X if acts[l - 1] == RELU else 0)
        start = dx0 if (l == 0 and not (dxflags & capi.LINEAR_DX_OVERWRITE)) else None
        dxl, dwl, dbl, dy_after = oracle.linear_bwd_ex(xin, ys[l], dy, ws[l], acts[l], f, dx0=start)
        exp_dw[l], exp_db[l], exp_dy[l] = dwl, dbl, dy_after
        a = np.abs(dy_after).astype(np.float64)
        mass_dw[l] = a.T @ np.abs(xin).astype(np.float64)
        mass_db[l] = a.sum(0)
        mass_dx[l] = a @ np.abs(ws[l]).astype(np.float64) + (np.abs(dx0) if start is not None else 0)
        if l == 0:
            exp_dx = dxl
        dy = dxl
    # ---- the chain on the GPU
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    xd, wd, yd = dev(x), [dev(w) for w in ws], [dev(y) for y in ys]
    for rep in range(2):
        dyd = [torch.full((B, widths[l + 1]), 5.0, device=DEV) for l in range(n)]
        dyd[-1] = dev(g_top)
        dwd = [torch.zeros(widths[l + 1], widths[l], device=DEV) for l in range(n)]
        dbd = [torch.zeros(widths[l + 1], device=DEV) for l in range(n)]
        dxd = dev(dx0) if want_dx else None
        layers = hip.chain_layers([dict(w=wd[l], y=yd[l], dy=dyd[l], dw=dwd[l], db=dbd[l], in_dim=widths[l], out_dim=widths[l + 1], activation=acts[l])
                                   for l in range(n)])
        flags = (capi.LINEAR_DY_PREMASKED if premasked else 0) | dxflags
        hip.check(hip.lib.ffh_mlp_chain_bwd(hip.ctx, capi.ptr(xd), widths[0], capi.ptr(dxd), widths[0], layers, n, B, flags, None), "chain bwd")
        route = _route(hip)
        torch.cuda.synchronize()
        assert "mlp_chain_dw" in route and "mlp_chain_dx" in route, route
        for l in range(n):
            _close(dwd[l].cpu().numpy(), exp_dw[l], mass_dw[l], f"chain {widths} dw{l} (launch {rep})")
            _close(dbd[l].cpu().numpy(), exp_db[l], mass_db[l], f"chain {widths} db{l}")
            _close(dyd[l].cpu().numpy(), exp_dy[l], (mass_dx[l + 1] if l + 1 < n else np.abs(exp_dy[l]).astype(np.float64)) + 1e-30, f"chain {widths} dy{l}")
        if want_dx:
            _close(dxd.cpu().numpy(), exp_dx, mass_dx[0], f"chain {widths} dx")


def test_mlp_chain_refuses_what_it_does_not_serve(hip):
    """Nothing launched, FFH_ERR_UNSUPPORTED / BAD_ARG: widths beyond 512, widths that do not chain, an inner sigmoid in the backward,
    deterministic mode (the weight gradients meet by atomics)."""
    B = 64
    t = lambda *s: torch.zeros(*s, device=DEV)
    mk = lambda i, o, act=RELU: dict(w=t(o, i), y=t(B, o), dy=t(B, o), dw=t(o, i), db=t(o), in_dim=i, out_dim=o, activation=act)
    x = t(B, 1024)
    assert hip.lib.ffh_mlp_chain_fwd(hip.ctx, capi.ptr(x), 1024, hip.chain_layers([mk(1024, 64), mk(64, 16)]), 2, B, None) == capi.FFH_ERR_BAD_ARG
    assert hip.lib.ffh_mlp_chain_fwd(hip.ctx, capi.ptr(x), 64, hip.chain_layers([mk(64, 32), mk(48, 16)]), 2, B, None) == capi.FFH_ERR_BAD_ARG
    assert hip.lib.ffh_mlp_chain_bwd(hip.ctx, capi.ptr(x), 64, None, 64, hip.chain_layers([mk(64, 32, SIG), mk(32, 16)]), 2, B, 0, None) == capi.FFH_ERR_UNSUPPORTED
    hip.check(hip.lib.ffh_ctx_set_deterministic(hip.ctx, 1), "deterministic")
    try:
        assert hip.lib.ffh_mlp_chain_bwd(hip.ctx, capi.ptr(x), 64, None, 64, hip.chain_layers([mk(64, 32), mk(32, 16)]), 2, B, 0, None) == capi.FFH_ERR_UNSUPPORTED
    finally:
        hip.check(hip.lib.ffh_ctx_set_deterministic(hip.ctx, 0), "deterministic")
