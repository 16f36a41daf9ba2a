"""ctypes binding of the operator C-ABI declared in include/ff_hip.h.

`FFHLib(path)` loads ANY library that exports that ABI and declares every
prototype; `load_hip()` loads the product library (libffhip.so, hand-written
gfx950 HIP) and raises loudly when it has not been built -- there is no
fallback of any kind.  Pointers cross as integers (`tensor.data_ptr()` or
`ndarray.ctypes.data`), sizes as int64, streams as `void*`.

The reference binds its operators to Python through python/flexflow_c.h
(opaque handles + cffi, [ref: python/flexflow_c.h:24-42]); this file is the
equivalent stub for the kernel tier.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "libffhip.so")
HEADER_PATH = os.path.join(REPO_ROOT, "include", "ff_hip.h")

FFH_OK = 0
FFH_ERR_BAD_ARG, FFH_ERR_HIP, FFH_ERR_UNSUPPORTED, FFH_ERR_WORKSPACE, FFH_ERR_NOMEM = -1, -2, -3, -4, -5
AC_MODE_NONE, AC_MODE_RELU, AC_MODE_SIGMOID, AC_MODE_TANH, AC_MODE_GELU = 10, 11, 12, 13, 14
AGGR_MODE_NONE, AGGR_MODE_SUM, AGGR_MODE_AVG = 20, 21, 22
MAX_TABLES = 64
EMB_CHUNK, EMB_CHUNK1 = 32, 1024
OPT_ZERO_GRAD = 1
CONCAT_BWD_OVERWRITE = 1
LINEAR_DX_OVERWRITE, LINEAR_ONLY_DW, LINEAR_ONLY_DX, LINEAR_DY_PREMASKED, LINEAR_DX_MASK_BY_X = 1, 2, 4, 8, 16
METRIC_ACCURACY, METRIC_MSE, METRIC_RMSE, METRIC_MAE = 1, 2, 4, 8

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
U64 = C.c_uint64
SZ = C.c_size_t


class EmbTable(C.Structure):
    """struct ffh_emb_table"""
    _fields_ = [("idx", P), ("weight", P), ("io", P), ("num_entries", L), ("ld", L)]


class EmbState(C.Structure):
    """struct ffh_emb_state: per-table optimizer state of the sparse (touched-rows) optimizers"""
    _fields_ = [("s0", P), ("s1", P)]


class SparseOpt(C.Structure):
    """struct ffh_sparse_opt"""
    _fields_ = [("kind", C.c_int32), ("lr", F), ("weight_decay", F), ("momentum", F), ("nesterov", C.c_int32),
                ("beta1", F), ("beta2", F), ("epsilon", F)]


SPARSE_OPT_SGD, SPARSE_OPT_SGD_MOMENTUM, SPARSE_OPT_ADAM = 0, 1, 2


class PerfMetrics(C.Structure):
    """struct ffh_perf_metrics"""
    _fields_ = [("train_all", C.c_int32), ("train_correct", C.c_int32), ("cce_loss", F),
                ("sparse_cce_loss", F), ("mse_loss", F), ("rmse_loss", F), ("mae_loss", F),
                ("pad_", C.c_int32)]


class ChainLayer(C.Structure):
    """struct ffh_chain_layer"""
    _fields_ = [("w", P), ("bias", P), ("y", P), ("dy", P), ("dw", P), ("db", P), ("ldy", L), ("lddy", L),
                ("ldw", C.c_int32), ("in_dim", C.c_int32), ("out_dim", C.c_int32), ("activation", C.c_int32)]


class DeviceInfo(C.Structure):
    """struct ffh_device_info"""
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("total_mem_bytes", L), ("lds_bytes_per_cu", C.c_int32),
                ("clock_khz", C.c_int32)]


# name -> (restype, argtypes); ctx is always the first argument where present
_SIGS = {
    "ffh_abi_version": (I, []),
    "ffh_backend_name": (C.c_char_p, []),
    "ffh_ctx_create": (I, [C.POINTER(P), I]),
    "ffh_ctx_destroy": (I, [P]),
    "ffh_ctx_default": (I, [C.POINTER(P)]),
    "ffh_linear_last_route": (C.c_char_p, [P]),
    "ffh_embedding_last_route": (C.c_char_p, [P]),
    "ffh_last_error_string": (C.c_char_p, [P]),
    "ffh_device_query": (I, [P, C.POINTER(DeviceInfo)]),
    "ffh_ctx_set_workspace": (I, [P, P, SZ]),
    "ffh_ctx_set_math_mode": (I, [P, I]),
    "ffh_ctx_set_deterministic": (I, [P, I]),
    "ffh_ctx_set_dw_cu_reserve": (I, [P, I]),
    "ffh_ctx_reserve_scratch": (I, [P, P]),
    "ffh_ctx_bf16_mirror_set": (I, [P, P, SZ, P]),
    "ffh_ctx_bf16x3_mirror_set": (I, [P, P, SZ, P]),
    "ffh_convert_f32_to_bf16x3": (I, [P, P, L, L, L, P]),
    "ffh_convert_f32_to_bf16": (I, [P, P, P, L, P]),
    "ffh_malloc": (I, [P, C.POINTER(P), SZ]),
    "ffh_free": (I, [P, P]),
    "ffh_memcpy_h2d": (I, [P, P, P, SZ, P]),
    "ffh_memcpy_d2h": (I, [P, P, P, SZ, P]),
    "ffh_memcpy_d2d": (I, [P, P, P, SZ, P]),
    "ffh_stream_create": (I, [P, C.POINTER(P)]),
    "ffh_stream_create_with_priority": (I, [P, C.POINTER(P), I]),
    "ffh_stream_destroy": (I, [P, P]),
    "ffh_stream_sync": (I, [P, P]),
    "ffh_device_sync": (I, [P]),
    "ffh_event_create": (I, [P, C.POINTER(P)]),
    "ffh_event_create_sync": (I, [P, C.POINTER(P)]),
    "ffh_event_destroy": (I, [P, P]),
    "ffh_event_record": (I, [P, P, P]),
    "ffh_event_sync": (I, [P, P]),
    "ffh_stream_wait_event": (I, [P, P, P]),
    "ffh_event_elapsed_ms": (I, [P, P, P, C.POINTER(F)]),
    "ffh_graph_begin_capture": (I, [P, P]),
    "ffh_graph_end_capture": (I, [P, P, C.POINTER(P)]),
    "ffh_graph_launch": (I, [P, P, P]),
    "ffh_graph_destroy": (I, [P, P]),
    "ffh_fill_f32": (I, [P, P, L, F, P]),
    "ffh_zero": (I, [P, P, SZ, P]),
    "ffh_init_uniform": (I, [P, P, L, U64, F, F, P]),
    "ffh_gen_indices": (I, [P, P, L, U64, L, L, P]),
    "ffh_gen_uniform01": (I, [P, P, L, U64, L, P]),
    "ffh_gen_bernoulli": (I, [P, P, L, U64, L, P]),
    "ffh_embedding_fwd": (I, [P, P, P, P, I, I, L, L, L, I, P]),
    "ffh_embedding_fwd_multi": (I, [P, C.POINTER(EmbTable), I, I, I, L, I, P]),
    "ffh_embedding_bwd_dense": (I, [P, P, P, P, I, I, L, L, L, I, P]),
    "ffh_embedding_bwd_sgd_fused": (I, [P, P, P, P, I, I, L, L, L, I, F, P]),
    "ffh_embedding_bwd_sgd_fused_multi": (I, [P, C.POINTER(EmbTable), I, I, I, L, I, F, P]),
    "ffh_embedding_bwd_sort_multi": (I, [P, C.POINTER(EmbTable), I, I, I, L, P]),
    "ffh_embedding_bwd_sgd_apply_multi": (I, [P, C.POINTER(EmbTable), I, I, I, L, I, F, P]),
    "ffh_embedding_bwd_opt_fused_multi": (I, [P, C.POINTER(EmbTable), C.POINTER(EmbState), I, I, I, L, I, C.POINTER(SparseOpt), P]),
    "ffh_embedding_bwd_opt_apply_multi": (I, [P, C.POINTER(EmbTable), C.POINTER(EmbState), I, I, I, L, I, C.POINTER(SparseOpt), P]),
    "ffh_embedding_bwd_workspace_bytes": (SZ, [I, I, I, L]),
    "ffh_embedding_localize_rows": (I, [P, P, P, L, L, L, P]),
    "ffh_linear_fwd": (I, [P, P, L, P, L, P, P, I, I, L, I, P]),
    "ffh_linear_fast_in_dim": (I, [I, I]),
    "ffh_linear_bwd": (I, [P, P, L, P, L, P, L, P, L, P, P, P, I, I, L, I, P]),
    "ffh_linear_bwd_ex": (I, [P, P, L, P, L, P, L, P, L, P, P, P, I, I, L, I, I, P, P]),
    "ffh_linear_bwd_mse": (I, [P, P, L, P, L, P, L, P, L, P, P, P, I, I, L, I, I, P, F, P, I, P]),
    "ffh_linear_pair_bwd": (I, [P, P, L, P, L, P, L, P, P, P, I, I, I, I, P, L, P, L, P, L, P, I, I, I, L, P]),
    "ffh_linear_pair_fwd": (I, [P, P, L, P, P, I, I, P, L, I, P, P, I, I, P, L, L, P]),
    "ffh_mlp_chain_fwd": (I, [P, P, L, C.POINTER(ChainLayer), I, L, P]),
    "ffh_mlp_chain_bwd": (I, [P, P, L, P, L, C.POINTER(ChainLayer), I, L, I, P]),
    "ffh_second_stream_used": (I, [P, I]),
    "ffh_event_record_with_next_linear_bwd": (I, [P, P]),
    "ffh_linear_bwd_set_dx_scatter": (I, [P, P, I, P]),
    "ffh_linear_dx_scatter_used": (I, [P]),
    "ffh_linear_bwd_set_dx_colsum": (I, [P, P, I]),
    "ffh_linear_dx_colsum_used": (I, [P]),
    "ffh_mse_bwd_metrics": (I, [P, P, P, P, P, L, I, F, I, P]),
    "ffh_concat_fwd": (I, [P, P, L, C.POINTER(P), C.POINTER(L), C.POINTER(L), I, L, P]),
    "ffh_concat_bwd": (I, [P, P, L, C.POINTER(P), C.POINTER(L), C.POINTER(L), I, L, P]),
    "ffh_concat_bwd_ex": (I, [P, P, L, C.POINTER(P), C.POINTER(L), C.POINTER(L), I, L, I, P]),
    "ffh_bmm_fwd": (I, [P, P, P, P, I, I, I, L, I, I, I, P]),
    "ffh_bmm_bwd": (I, [P, P, P, P, P, P, I, I, I, L, P]),
    "ffh_transpose_fwd": (I, [P, P, P, I, C.POINTER(L), C.POINTER(I), P]),
    "ffh_transpose_bwd": (I, [P, P, P, I, C.POINTER(L), C.POINTER(I), P]),
    "ffh_tril_fwd": (I, [P, P, L, P, L, I, P]),
    "ffh_tril_bwd": (I, [P, P, P, L, L, I, P]),
    "ffh_dot_interaction_fwd": (I, [P, P, L, P, L, L, I, I, P]),
    "ffh_dot_interaction_bwd": (I, [P, P, L, P, L, P, L, L, I, I, I, P]),
    "ffh_mse_bwd": (I, [P, P, P, P, L, F, P]),
    "ffh_metrics_update": (I, [P, P, P, P, L, I, I, P]),
    "ffh_sgd_update": (I, [P, P, P, P, L, F, F, F, I, P]),
    "ffh_sgd_update_ex": (I, [P, P, P, P, L, F, F, F, I, I, P]),
    "ffh_adam_update": (I, [P, P, P, P, P, L, F, F, F, F, F, I, P]),
    "ffh_add_scaled": (I, [P, P, P, L, F, P]),
    "ffh_sum_slices_f32": (I, [P, P, P, I, L, L, P]),
}


def header_symbols(header_path: str = HEADER_PATH) -> list[str]:
    """Every symbol of the FFH_API_LIST X-macro in include/ff_hip.h."""
    text = open(header_path).read()
    m = re.search(r"#define FFH_API_LIST\(X\)(.*?)\n\n", text, re.S)
    if not m:
        raise RuntimeError("FFH_API_LIST not found in " + header_path)
    return re.findall(r"X\((\w+)\)", m.group(1))


def header_abi_version(header_path: str = HEADER_PATH) -> int:
    """FFH_ABI_VERSION of include/ff_hip.h."""
    m = re.search(r"#define\s+FFH_ABI_VERSION\s+(\d+)", open(header_path).read())
    if not m:
        raise RuntimeError("FFH_ABI_VERSION not found in " + header_path)
    return int(m.group(1))


class FFHError(RuntimeError):
    pass


def ptr(x) -> int:
    """Device/host address of a torch tensor, numpy array, int or None."""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    raise TypeError(f"cannot take the address of {type(x)}")


class FFHLib:
    """A loaded library exporting include/ff_hip.h, plus one ctx."""

    def __init__(self, path: str, device: int = 0):
        if not os.path.exists(path):
            raise FFHError(f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.lib, name)          # AttributeError => symbol missing => loud
            fn.restype = res
            fn.argtypes = args
        if self.lib.ffh_abi_version() != header_abi_version():
            raise FFHError(f"{path}: ABI version {self.lib.ffh_abi_version()}, include/ff_hip.h says {header_abi_version()} (rebuild)")
        self.backend = self.lib.ffh_backend_name().decode()
        ctx = P()
        rc = self.lib.ffh_ctx_create(C.byref(ctx), device)
        if rc != FFH_OK:
            raise FFHError(f"ffh_ctx_create failed ({rc}) on {path}")
        self.ctx = ctx
        self._ws_keepalive = None
        # tests and tools launch on the null stream: its scratch (stream-K slots ...) is reserved here, by the caller -- the library's
        # compute entry points never allocate
        self.check(self.lib.ffh_ctx_reserve_scratch(self.ctx, None), "ffh_ctx_reserve_scratch")

    # -- plumbing -----------------------------------------------------------
    def check(self, rc: int, what: str = ""):
        if rc != FFH_OK:
            msg = self.lib.ffh_last_error_string(self.ctx)
            raise FFHError(f"{what} failed: rc={rc}: {msg.decode() if msg else ''}")

    def call(self, name: str, *args):
        """Call `name(ctx, *args)`; pointers may be tensors/arrays/ints/None."""
        conv = []
        sig = _SIGS[name][1][1:]
        for a, t in zip(args, sig):
            conv.append(ptr(a) if t is P else a)
        if len(args) != len(sig):
            raise TypeError(f"{name}: expected {len(sig)} args, got {len(args)}")
        self.check(getattr(self.lib, name)(self.ctx, *conv), name)

    def set_workspace(self, buf, nbytes: int):
        self._ws_keepalive = buf
        self.check(self.lib.ffh_ctx_set_workspace(self.ctx, ptr(buf), nbytes), "ffh_ctx_set_workspace")

    def device_info(self) -> DeviceInfo:
        info = DeviceInfo()
        self.check(self.lib.ffh_device_query(self.ctx, C.byref(info)), "ffh_device_query")
        return info

    def close(self):
        if self.ctx:
            self.lib.ffh_ctx_destroy(self.ctx)
            self.ctx = None

    # -- helpers for the array-taking entry points --------------------------
    @staticmethod
    def emb_tables(entries) -> "C.Array":
        """entries: iterable of (idx, weight, io, num_entries, ld)."""
        entries = list(entries)
        arr = (EmbTable * len(entries))()
        for k, (idx, w, io, r, ld) in enumerate(entries):
            arr[k] = EmbTable(ptr(idx), ptr(w), ptr(io), int(r), int(ld))
        return arr

    @staticmethod
    def emb_states(entries) -> "C.Array":
        """entries: iterable of (s0, s1) buffers (None where the optimizer kind has no such state)."""
        entries = list(entries)
        arr = (EmbState * len(entries))()
        for k, (s0, s1) in enumerate(entries):
            arr[k] = EmbState(ptr(s0), ptr(s1))
        return arr

    @staticmethod
    def chain_layers(entries) -> "C.Array":
        """entries: iterable of dicts with w, bias, y, dy, dw, db, ldy, lddy, ldw, in_dim, out_dim, activation (missing: None / 0)."""
        entries = list(entries)
        arr = (ChainLayer * len(entries))()
        for k, e in enumerate(entries):
            arr[k] = ChainLayer(ptr(e.get("w")), ptr(e.get("bias")), ptr(e.get("y")), ptr(e.get("dy")), ptr(e.get("dw")), ptr(e.get("db")),
                                int(e.get("ldy", e["out_dim"])), int(e.get("lddy", e["out_dim"])), int(e.get("ldw", e["in_dim"])),
                                int(e["in_dim"]), int(e["out_dim"]), int(e["activation"]))
        return arr

    def transpose(self, name: str, dst, src, in_dims, perm, stream=None):
        n = len(in_dims)
        da = (L * n)(*[int(v) for v in in_dims])
        pa = (I * n)(*[int(v) for v in perm])
        self.check(getattr(self.lib, name)(self.ctx, ptr(dst), ptr(src), n, da, pa, ptr(stream)), name)

    def concat(self, name: str, big, out_blk: int, parts, in_blk, in_ld, num_blocks: int, stream=None):
        n = len(parts)
        pa = (P * n)(*[ptr(p) for p in parts])
        ba = (L * n)(*[int(v) for v in in_blk])
        la = (L * n)(*[int(v) for v in in_ld]) if in_ld is not None else None
        self.check(getattr(self.lib, name)(self.ctx, ptr(big), out_blk, pa, ba, la, n, num_blocks, ptr(stream)), name)


_hip_singleton: FFHLib | None = None


def load_hip(device: int = 0) -> FFHLib:
    """The product library.  Fails loudly if the HIP extension is missing."""
    global _hip_singleton
    if _hip_singleton is None:
        import torch  # noqa: F401  (first: one HIP runtime per process, torch's libamdhip64.so.7)
        _hip_singleton = FFHLib(HIP_LIB_PATH, device)
        if not _hip_singleton.backend.startswith("hip"):
            raise FFHError(f"{HIP_LIB_PATH} is not the HIP backend ({_hip_singleton.backend})")
    return _hip_singleton
