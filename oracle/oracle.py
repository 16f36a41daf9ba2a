"""TEST INFRASTRUCTURE -- Python face of the CPU oracle (oracle/ffh_oracle.c) and of the
reference's own AVX2 embedding lookup (oracle/_ref/libref_embedding.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; nothing under dlrm_flexflow_amd/ does.  All functions take and return numpy
arrays on the host.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "_build", "libffh_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libref_embedding.so")

import sys
sys.path.insert(0, os.path.dirname(_HERE))
from dlrm_flexflow_amd import capi  # noqa: E402  (the ctypes prototypes are shared)


def build(force: bool = False) -> None:
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(ORACLE_LIB) or \
            os.path.getmtime(ORACLE_LIB) < os.path.getmtime(os.path.join(_HERE, "ffh_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> capi.FFHLib:
    global _lib
    if _lib is None:
        build()
        _lib = capi.FFHLib(ORACLE_LIB)
        assert _lib.backend == "oracle-cpu"
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ---------------------------------------------------------------------------
# numpy-level wrappers (one per row of SURVEY.md section 8a)
# ---------------------------------------------------------------------------
def embedding_fwd(idx, w, aggr=capi.AGGR_MODE_SUM):
    idx, w = _i64(idx), _f32(w)
    B, Lb = idx.shape
    R, D = w.shape
    out = np.empty((B, D), np.float32)
    lib().call("ffh_embedding_fwd", idx, out, w, Lb, D, B, R, D, aggr, None)
    return out


def embedding_bwd_dense(idx, g, R, aggr=capi.AGGR_MODE_SUM, wgrad=None):
    idx, g = _i64(idx), _f32(g)
    B, Lb = idx.shape
    D = g.shape[1]
    wg = np.zeros((R, D), np.float32) if wgrad is None else wgrad
    lib().call("ffh_embedding_bwd_dense", idx, g, wg, Lb, D, B, R, D, aggr, None)
    return wg


def embedding_bwd_sgd_fused(idx, g, w, lr, aggr=capi.AGGR_MODE_SUM):
    """Returns the updated copy of w."""
    idx, g = _i64(idx), _f32(g)
    w = _f32(w).copy()
    B, Lb = idx.shape
    R, D = w.shape
    lib().call("ffh_embedding_bwd_sgd_fused", idx, g, w, Lb, D, B, R, D, aggr, float(lr), None)
    return w


def embedding_bwd_opt(idx, g, w, opt: "capi.SparseOpt", s0=None, s1=None, aggr=capi.AGGR_MODE_SUM):
    """ffh_embedding_bwd_opt_fused_multi on one table: returns (w, s0, s1) after the touched-rows optimizer step."""
    idx, g = _i64(idx), _f32(g)
    w = _f32(w).copy()
    s0 = None if s0 is None else _f32(s0).copy()
    s1 = None if s1 is None else _f32(s1).copy()
    B, Lb = idx.shape
    R, D = w.shape
    t = lib().emb_tables([(idx, w, g, R, g.shape[1])])
    st = lib().emb_states([(s0, s1)])
    lib().check(lib().lib.ffh_embedding_bwd_opt_fused_multi(lib().ctx, t, st, 1, Lb, D, B, aggr, C.byref(opt), None), "ffh_embedding_bwd_opt_fused_multi")
    return w, s0, s1


def linear_fwd(x, w, bias, act=capi.AC_MODE_NONE):
    x, w = _f32(x), _f32(w)
    B, IN = x.shape
    OUT = w.shape[0]
    y = np.empty((B, OUT), np.float32)
    b = None if bias is None else _f32(bias)
    lib().call("ffh_linear_fwd", x, IN, y, OUT, w, b, IN, OUT, B, act, None)
    return y


def linear_bwd(x, y, dy, w, act=capi.AC_MODE_NONE, want_dx=True, use_bias=True):
    """Returns (dx, dw, db, dy_after_activation_grad), all accumulated from zero."""
    x, y, w = _f32(x), _f32(y), _f32(w)
    dy = _f32(dy).copy()
    B, IN = x.shape
    OUT = w.shape[0]
    dx = np.zeros((B, IN), np.float32) if want_dx else None
    dw = np.zeros((OUT, IN), np.float32)
    db = np.zeros((OUT,), np.float32) if use_bias else None
    lib().call("ffh_linear_bwd", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, act, None)
    return dx, dw, db, dy


def linear_bwd_ex(x, y, dy, w, act, flags, dx0=None, use_bias=True):
    """ffh_linear_bwd_ex with explicit flags (capi.LINEAR_*); dx starts from dx0 (zeros if None).
    Returns (dx, dw, db, dy_after)."""
    x, y, w = _f32(x), _f32(y), _f32(w)
    dy = _f32(dy).copy()
    B, IN = x.shape
    OUT = w.shape[0]
    dx = np.zeros((B, IN), np.float32) if dx0 is None else _f32(dx0).copy()
    dw = np.zeros((OUT, IN), np.float32)
    db = np.zeros((OUT,), np.float32) if use_bias else None
    lib().call("ffh_linear_bwd_ex", x, IN, dx, IN, y, OUT, dy, OUT, w, dw, db, IN, OUT, B, act, flags, None, None)
    return dx, dw, db, dy


def concat_fwd(parts):
    parts = [_f32(p) for p in parts]
    nb = parts[0].shape[0]
    widths = [p.shape[1] for p in parts]
    out = np.empty((nb, sum(widths)), np.float32)
    lib().concat("ffh_concat_fwd", out, sum(widths), parts, widths, None, nb)
    return out


def concat_bwd(og, widths, accumulate_into=None):
    og = _f32(og)
    nb = og.shape[0]
    grads = accumulate_into or [np.zeros((nb, w), np.float32) for w in widths]
    lib().concat("ffh_concat_bwd", og, og.shape[1], grads, widths, None, nb)
    return grads


def bmm_fwd(a, b, a_seq=-1, b_seq=-1, seq=-1):
    a, b = _f32(a), _f32(b)
    batch, n, k = a.shape
    m = b.shape[2]
    o = np.zeros((batch, n, m), np.float32)
    lib().call("ffh_bmm_fwd", o, a, b, m, n, k, batch, a_seq, b_seq, seq, None)
    return o


def bmm_bwd(og, a, b):
    og, a, b = _f32(og), _f32(a), _f32(b)
    batch, n, k = a.shape
    m = b.shape[2]
    ag = np.zeros_like(a)
    bg = np.zeros_like(b)
    lib().call("ffh_bmm_bwd", og, a, ag, b, bg, m, n, k, batch, None)
    return ag, bg


def mse_bwd(logit, label, scale):
    logit, label = _f32(logit), _f32(label)
    g = np.empty_like(logit)
    lib().call("ffh_mse_bwd", g, logit, label, logit.size, float(scale), None)
    return g


def metrics_update(logits, labels, flags, perf=None):
    logits, labels = _f32(logits), _f32(labels)
    ns, nc = logits.shape
    perf = perf or capi.PerfMetrics()
    lib().check(lib().lib.ffh_metrics_update(lib().ctx, logits.ctypes.data, labels.ctypes.data,
                                             C.addressof(perf), ns, nc, flags, None), "ffh_metrics_update")
    return perf


def sgd_update(w, g, lr, wd=0.0, momentum=0.0, nesterov=False, v=None):
    w = _f32(w).copy()
    g = _f32(g)
    lib().call("ffh_sgd_update", w, g, v, w.size, float(lr), float(wd), float(momentum), int(nesterov), None)
    return w


def sgd_update_zero_grad(w, g, lr, wd=0.0, momentum=0.0, nesterov=False, v=None):
    """ffh_sgd_update_ex with FFH_OPT_ZERO_GRAD; returns (w, g_after)."""
    w, g = _f32(w).copy(), _f32(g).copy()
    lib().call("ffh_sgd_update_ex", w, g, v, w.size, float(lr), float(wd), float(momentum), int(nesterov), capi.OPT_ZERO_GRAD, None)
    return w, g


class AdamState:
    """AdamOptimizer's host-side scalars [ref: src/runtime/optimizer.cc:194-201,248-254] (doubles, as there)."""

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, weight_decay=0.0, epsilon=1e-8):
        self.alpha, self.beta1, self.beta2, self.weight_decay, self.epsilon = alpha, beta1, beta2, weight_decay, epsilon
        self.alpha_t, self.beta1_t, self.beta2_t = alpha, 1.0, 1.0

    def next(self):
        self.beta1_t *= self.beta1
        self.beta2_t *= self.beta2
        self.alpha_t = self.alpha * np.sqrt(1 - self.beta2_t) / (1 - self.beta1_t)


def adam_update(w, g, m, v, st: AdamState, zero_grad=False):
    """One adam_update over (w, m, v) in place of copies; returns (w, m, v[, g_after])."""
    w, g, m, v = _f32(w).copy(), _f32(g).copy(), _f32(m).copy(), _f32(v).copy()
    lib().call("ffh_adam_update", w, g, m, v, w.size, float(st.alpha_t), float(st.beta1), float(st.beta2), float(st.weight_decay),
               float(st.epsilon), capi.OPT_ZERO_GRAD if zero_grad else 0, None)
    return (w, m, v, g) if zero_grad else (w, m, v)


def init_uniform(count, seed, lo, hi):
    p = np.empty(count, np.float32)
    lib().call("ffh_init_uniform", p, count, seed, float(lo), float(hi), None)
    return p


def gen_indices(count, seed, first, R):
    p = np.empty(count, np.int64)
    lib().call("ffh_gen_indices", p, count, seed, first, R, None)
    return p


def embedding_localize_rows(idx, row_begin, rows_local):
    """Row-wise sharded table (this build's extension): global ids -> ids relative to row_begin, rows held elsewhere ->
    rows_local (the zero row behind the local slice)."""
    idx = _i64(idx)
    out = np.empty_like(idx)
    lib().call("ffh_embedding_localize_rows", idx, out, idx.size, row_begin, rows_local, None)
    return out


def tril_fwd(z, out_ld=None, col_off=0):
    """out[:, col_off : col_off + n(n-1)/2] = strict lower triangle of z [B][n][n]; the rest of out stays 777."""
    z = _f32(z)
    B, n, _ = z.shape
    P = n * (n - 1) // 2
    out_ld = out_ld or P
    out = np.full((B, out_ld), 777.0, np.float32)
    lib().call("ffh_tril_fwd", out[:, col_off:], out_ld, z, B, n, None)
    return out


def tril_bwd(g, in_grad):
    """Returns in_grad with g [B][n(n-1)/2] added into its strict lower triangle."""
    g, out = _f32(g), _f32(in_grad).copy()
    B, n, _ = out.shape
    lib().call("ffh_tril_bwd", out, g, g.shape[1], B, n, None)
    return out


def dot_interaction_fwd(z):
    """z [B][c][d] -> [B][d + c(c-1)/2]: row 0 passed through, then <z_i, z_j> for i > j."""
    z = _f32(z)
    B, c, d = z.shape
    out = np.empty((B, d + c * (c - 1) // 2), np.float32)
    lib().call("ffh_dot_interaction_fwd", z, c * d, out, out.shape[1], B, c, d, None)
    return out


def dot_interaction_bwd(z, g, z_grad=None):
    """Gradient wrt z of dot_interaction_fwd given g [B][d + c(c-1)/2]; added to z_grad when given."""
    z, g = _f32(z), _f32(g)
    B, c, d = z.shape
    out = np.zeros_like(z) if z_grad is None else _f32(z_grad).copy()
    lib().call("ffh_dot_interaction_bwd", z, c * d, g, g.shape[1], out, c * d, B, c, d, 1 if z_grad is None else 0, None)
    return out


def gen_uniform01(count, seed, first):
    p = np.empty(count, np.float32)
    lib().call("ffh_gen_uniform01", p, count, seed, first, None)
    return p


def gen_bernoulli(count, seed, first):
    p = np.empty(count, np.float32)
    lib().call("ffh_gen_bernoulli", p, count, seed, first, None)
    return p


# ---------------------------------------------------------------------------
# the reference's own compiled function (oracle/_ref)
# ---------------------------------------------------------------------------
_REF_FWD = "_Z13embed_forwardPKlPKiPfPKfiiii"             # embed_forward(const int64_t*, const int*, float*, const float*, int,int,int,int)
_REF_LOOKUP = "_Z45EmbeddingLookup_int64_t_float_float__avx2_fmaiiiiPKfPKlPKiS0_bPf"


def ref_available() -> bool:
    return os.path.exists(REF_LIB)


def ref_embedding_fwd(idx, w, lengths=None, normalize=False):
    """The reference's EmbeddingLookup_int64_t_float_float__avx2_fma
    [ref: src/ops/embedding.cc:23-319] on host arrays.  idx is the flat index list,
    `lengths` the per-bag lengths (default: every bag has idx.shape[1] entries)."""
    lib_ = C.CDLL(REF_LIB)
    fn = getattr(lib_, _REF_LOOKUP)
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_bool, C.c_void_p]
    idx, w = _i64(idx), _f32(w)
    B = idx.shape[0]
    Lb = idx.shape[1] if idx.ndim == 2 else 1
    R, D = w.shape
    if lengths is None:
        lengths = np.full(B, Lb, np.int32)
    lengths = np.ascontiguousarray(lengths, np.int32)
    flat = idx.reshape(-1)
    out = np.empty((B, D), np.float32)
    fn(D, B, flat.size, R, w.ctypes.data, flat.ctypes.data, lengths.ctypes.data, None, normalize, out.ctypes.data)
    return out


_REF_BWD = "_Z14embed_backwardPKlPKiPKfPfiiii"   # embed_backward(const int64_t* in, const int* lengths, const float* out_grad, float* embed, block, B, index_size, data_size)


def ref_embedding_bwd(idx, g, wgrad):
    """The reference's CPU embed_backward [ref: src/ops/embedding.cc:344-374] (compiled from its source, oracle/Makefile):
    wgrad[idx[b]] += g[b] for b ascending.  It reads ONE index per sample (the reference asserts in_dim == 1 on this
    path, src/ops/embedding.cc:408), so idx is [B] or [B][1].  Returns the updated copy of wgrad."""
    lib_ = C.CDLL(REF_LIB)
    fn = getattr(lib_, _REF_BWD)
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    idx, g = _i64(idx).reshape(-1), _f32(g)
    B, D = g.shape
    assert idx.size == B
    wg = _f32(wgrad).copy()
    lengths = np.ones(B, np.int32)
    fn(idx.ctypes.data, lengths.ctypes.data, g.ctypes.data, wg.ctypes.data, D, B, B, wg.shape[0])
    return wg
