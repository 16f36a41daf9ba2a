"""Python face of the C++ FFModel shim (dlrm_flexflow_amd/host/libffmodel.so), bound with ctypes
over host/ffmodel_c.h -- the counterpart of the reference's python/flexflow/core/flexflow_cffi.py
over python/flexflow_c.h.  Method names follow the reference's Python API
(`ffmodel.dense / embedding / concat / batch_matmul / compile / forward / backward / update`).

The operator kernels come from the library `FFConfig.backend` names; the default is the HIP
library and a missing build raises -- nothing here falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "host", "libffmodel.so")

# enums [ref: include/ffconst.h:4-57]
DT_FLOAT, DT_INT64 = 40, 43
LOSS_MSE_AVG, LOSS_MSE_SUM = 52, 53
METRICS_ACCURACY, METRICS_MSE = 1001, 1008
COMP_MODE_TRAINING = 70


class _H(C.Structure):
    _fields_ = [("impl", C.c_void_p)]


class PerfMetrics(C.Structure):
    _fields_ = [("train_all", C.c_int), ("train_correct", C.c_int), ("cce_loss", C.c_float),
                ("sparse_cce_loss", C.c_float), ("mse_loss", C.c_float), ("rmse_loss", C.c_float),
                ("mae_loss", C.c_float)]


ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
REDUCE_SCATTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class FFComm(C.Structure):
    """struct ffcomm (host/ffcomm.h)"""
    _fields_ = [("rank", C.c_int), ("world_size", C.c_int), ("user", C.c_void_p),
                ("alltoall_f32", ALLTOALL_FN), ("allreduce_sum_f32", ALLREDUCE_FN), ("barrier", BARRIER_FN), ("nonblocking", C.c_int),
                ("reduce_scatter_sum_f32", REDUCE_SCATTER_FN), ("allgather_f32", ALLGATHER_FN), ("allreduce_bucket_sum_f32", ALLREDUCE_FN), ("bucket_channel_own", C.c_int),
                ("alltoall_bucket_f32", ALLTOALL_FN), ("allgather_bucket_f32", ALLGATHER_FN)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIB_PATH):
        raise capi.FFHError(f"{HOST_LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
    import torch  # noqa: F401  one HIP runtime per process: torch's copy is loaded first
    L = C.CDLL(HOST_LIB_PATH, mode=C.RTLD_GLOBAL)
    H, I, P, B, F, D = _H, C.c_int, C.c_void_p, C.c_bool, C.c_float, C.c_double
    IP = C.POINTER(C.c_int)
    sigs = {
        "flexflow_config_create": (H, []), "flexflow_config_destroy": (None, [H]),
        "flexflow_config_parse_args": (None, [H, C.POINTER(C.c_char_p), I]),
        "flexflow_config_set_comm": (None, [H, C.POINTER(FFComm)]),
        "flexflow_rccl_available": (I, [C.c_char_p]), "flexflow_rccl_get_unique_id": (I, [C.POINTER(C.c_ubyte), C.c_char_p]),
        "flexflow_rccl_comm_create": (I, [C.POINTER(C.c_ubyte), I, I, C.c_char_p, C.POINTER(FFComm)]),
        "flexflow_rccl_comm_destroy": (None, [C.POINTER(FFComm)]), "flexflow_rccl_last_error": (C.c_char_p, []),
        "flexflow_rccl_comm_calls": (None, [C.POINTER(FFComm), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "flexflow_rccl_comm_calls2": (None, [C.POINTER(FFComm), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "flexflow_config_set_batch_size": (None, [H, I]), "flexflow_config_get_batch_size": (I, [H]),
        "flexflow_config_set_backend": (None, [H, C.c_char_p]), "flexflow_config_set_seed": (None, [H, C.c_uint64]),
        "flexflow_config_set_device": (None, [H, I]), "flexflow_config_set_enable_graph": (None, [H, B]),
        "flexflow_config_set_overlap_embedding": (None, [H, B]), "flexflow_config_set_dense_embedding_update": (None, [H, B]),
        "flexflow_model_create": (H, [H]), "flexflow_model_destroy": (None, [H]),
        "flexflow_tensor_create": (H, [H, I, IP, I, B]),
        "flexflow_model_add_dense": (H, [H, H, I, I, B, H, H, C.c_char_p]),
        "flexflow_model_add_embedding": (H, [H, H, I, I, I, H, C.c_char_p]),
        "flexflow_model_add_concat": (H, [H, I, C.POINTER(H), I, C.c_char_p]),
        "flexflow_model_add_batch_matmul": (H, [H, H, H, I, I]),
        "flexflow_model_add_flat": (H, [H, H, C.c_char_p]),
        "flexflow_model_add_tril": (H, [H, H, C.c_char_p]),
        "flexflow_model_add_dot_interaction": (H, [H, H, I, C.c_char_p]),
        "flexflow_model_add_transpose": (H, [H, H, I, IP, C.c_char_p]),
        "flexflow_model_add_reshape": (H, [H, H, I, IP, C.c_char_p]),
        "flexflow_zero_initializer_create": (H, []), "flexflow_uniform_initializer_create": (H, [I, F, F]),
        "flexflow_norm_initializer_create": (H, [I, F, F]), "flexflow_glorot_uniform_initializer_create": (H, [I]),
        "flexflow_sgd_optimizer_create": (H, [H, D, D, B, D]), "flexflow_model_set_sgd_optimizer": (None, [H, H]),
        "flexflow_adam_optimizer_create": (H, [H, D, D, D, D, D]), "flexflow_model_set_adam_optimizer": (None, [H, H]),
        "flexflow_adam_optimizer_set_lr": (None, [H, D]),
        "flexflow_model_compile": (None, [H, I, IP, I, I]),
        "flexflow_model_init_layers": (None, [H]), "flexflow_model_reset_metrics": (None, [H]),
        "flexflow_model_forward": (None, [H, I]), "flexflow_model_zero_gradients": (None, [H]),
        "flexflow_model_backward": (None, [H, I]), "flexflow_model_update": (None, [H]),
        "flexflow_model_begin_trace": (None, [H, I]), "flexflow_model_end_trace": (None, [H, I]),
        "flexflow_model_sync": (None, [H]), "flexflow_model_get_perf_metrics": (None, [H, C.POINTER(PerfMetrics)]),
        "flexflow_model_get_label_tensor": (H, [H]), "flexflow_model_get_num_layers": (I, [H]),
        "flexflow_model_get_layer_name": (C.c_char_p, [H, I]), "flexflow_model_get_layer_num_weights": (I, [H, I]),
        "flexflow_model_get_parameter": (H, [H, I, I]), "flexflow_model_get_layer_output": (H, [H, I]),
        "flexflow_model_get_stream": (P, [H]), "flexflow_model_uses_graph": (I, [H]), "flexflow_model_set_trace_mode": (None, [H, I]), "flexflow_model_trace_replays": (I, [H, I]),
        "flexflow_model_get_counter": (C.c_int64, [H, C.c_char_p]),
        "flexflow_model_get_backend_name": (C.c_char_p, [H]), "flexflow_model_get_backend_path": (C.c_char_p, [H]),
        "flexflow_tensor_get_num_dims": (I, [H]), "flexflow_tensor_get_dims": (None, [H, IP]),
        "flexflow_tensor_get_local_rows": (C.c_int64, [H]), "flexflow_tensor_is_local": (B, [H]),
        "flexflow_tensor_get_device_ptr": (P, [H]), "flexflow_tensor_get_ld": (C.c_int64, [H]),
        "flexflow_tensor_set_float": (None, [H, H, IP, I, P]), "flexflow_tensor_set_int64": (None, [H, H, IP, I, P]),
        "flexflow_tensor_get_float": (None, [H, H, P]), "flexflow_tensor_get_int64": (None, [H, H, P]),
        "flexflow_tensor_get_grad_float": (None, [H, H, P]),
        "flexflow_dlrm_create": (H, [I, C.POINTER(C.c_char_p), C.POINTER(FFComm)]), "flexflow_dlrm_destroy": (None, [H]),
        "flexflow_dlrm_get_model": (H, [H]), "flexflow_dlrm_get_num_samples": (I, [H]), "flexflow_dlrm_get_num_tables": (I, [H]),
        "flexflow_dlrm_get_sparse_input": (H, [H, I]), "flexflow_dlrm_get_dense_input": (H, [H]),
        "flexflow_dlrm_warmup": (None, [H]), "flexflow_dlrm_train_steps": (None, [H, I, B]),
        "flexflow_dlrm_run_epochs": (D, [H]), "flexflow_dlrm_time_kernel": (F, [H, I, I]),
        "flexflow_dlrm_probe_step": (None, [H, I, C.POINTER(F), I]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _argv(args):
    arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
    return arr


class Tensor:
    def __init__(self, handle, model: "FFModel | None"):
        self.h = handle
        self.model = model

    @property
    def dims(self):
        n = lib().flexflow_tensor_get_num_dims(self.h)
        d = (C.c_int * n)()
        lib().flexflow_tensor_get_dims(self.h, d)
        return tuple(d)

    @property
    def local_rows(self) -> int:
        return lib().flexflow_tensor_get_local_rows(self.h)

    @property
    def is_local(self) -> bool:
        return bool(lib().flexflow_tensor_is_local(self.h))

    @property
    def device_ptr(self) -> int:
        """Address of element (0, 0) in the backend's memory (0: not held by this rank); tests / tools only."""
        return lib().flexflow_tensor_get_device_ptr(self.h) or 0

    @property
    def ld(self) -> int:
        return lib().flexflow_tensor_get_ld(self.h)

    def _local_shape(self):
        d = self.dims
        if len(d) == 1:
            return d
        if len(d) == 2:
            return (self.local_rows, d[1])
        return (self.local_rows // int(np.prod(d[1:-1])),) + d[1:]

    def set(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        dims = (C.c_int * a.ndim)(*a.shape)
        if a.dtype == np.float32:
            lib().flexflow_tensor_set_float(self.h, self.model.h, dims, a.ndim, a.ctypes.data)
        elif a.dtype == np.int64:
            lib().flexflow_tensor_set_int64(self.h, self.model.h, dims, a.ndim, a.ctypes.data)
        else:
            raise TypeError(a.dtype)

    def get(self, dtype=np.float32) -> np.ndarray:
        out = np.empty(self._local_shape(), dtype)
        if dtype == np.float32:
            lib().flexflow_tensor_get_float(self.h, self.model.h, out.ctypes.data)
        else:
            lib().flexflow_tensor_get_int64(self.h, self.model.h, out.ctypes.data)
        return out

    def get_grad(self) -> np.ndarray:
        out = np.empty(self._local_shape(), np.float32)
        lib().flexflow_tensor_get_grad_float(self.h, self.model.h, out.ctypes.data)
        return out

    # reference names [ref: python/flexflow/core/flexflow_cffi.py Parameter.set_weights/get_weights]
    set_weights = set
    get_weights = get


class FFConfig:
    def __init__(self, argv=None, backend: str | None = None, comm: "FFComm | None" = None):
        self.h = lib().flexflow_config_create()
        self._comm = comm
        if argv:
            full = ["ffmodel"] + list(argv)
            lib().flexflow_config_parse_args(self.h, _argv(full), len(full))
        if backend:
            lib().flexflow_config_set_backend(self.h, backend.encode())
        if comm is not None:
            lib().flexflow_config_set_comm(self.h, C.byref(comm))

    batch_size = property(lambda s: lib().flexflow_config_get_batch_size(s.h),
                          lambda s, v: lib().flexflow_config_set_batch_size(s.h, v))

    def set(self, seed=None, device=None, enable_graph=None, overlap_embedding=None, dense_embedding_update=None):
        if seed is not None: lib().flexflow_config_set_seed(self.h, seed)
        if device is not None: lib().flexflow_config_set_device(self.h, device)
        if enable_graph is not None: lib().flexflow_config_set_enable_graph(self.h, enable_graph)
        if overlap_embedding is not None: lib().flexflow_config_set_overlap_embedding(self.h, overlap_embedding)
        if dense_embedding_update is not None: lib().flexflow_config_set_dense_embedding_update(self.h, dense_embedding_update)
        return self


_NULL = _H(None)


class FFModel:
    def __init__(self, config: FFConfig | None = None, _handle=None):
        self.config = config
        self.h = _handle if _handle is not None else lib().flexflow_model_create(config.h)
        self._owned = _handle is None

    # -- graph construction (reference names) -------------------------------------------------
    def create_tensor(self, dims, data_type=DT_FLOAT, create_grad=True) -> Tensor:
        d = (C.c_int * len(dims))(*dims)
        return Tensor(lib().flexflow_tensor_create(self.h, len(dims), d, data_type, create_grad), self)

    def dense(self, input: Tensor, out_dim, activation=capi.AC_MODE_NONE, use_bias=True, kernel_initializer=None,
              bias_initializer=None, name=None) -> Tensor:
        return Tensor(lib().flexflow_model_add_dense(self.h, input.h, out_dim, activation, use_bias,
                                                     kernel_initializer or _NULL, bias_initializer or _NULL,
                                                     name.encode() if name else None), self)

    def embedding(self, input: Tensor, num_entries, out_dim, aggr=capi.AGGR_MODE_SUM, kernel_initializer=None, name=None) -> Tensor:
        return Tensor(lib().flexflow_model_add_embedding(self.h, input.h, num_entries, out_dim, aggr,
                                                         kernel_initializer or _NULL, name.encode() if name else None), self)

    def concat(self, tensors, axis, name=None) -> Tensor:
        arr = (_H * len(tensors))(*[t.h for t in tensors])
        return Tensor(lib().flexflow_model_add_concat(self.h, len(tensors), arr, axis, name.encode() if name else None), self)

    def batch_matmul(self, a: Tensor, b: Tensor, a_seq_length_dim=-1, b_seq_length_dim=-1) -> Tensor:
        return Tensor(lib().flexflow_model_add_batch_matmul(self.h, a.h, b.h, a_seq_length_dim, b_seq_length_dim), self)

    def flat(self, input: Tensor, name=None) -> Tensor:
        return Tensor(lib().flexflow_model_add_flat(self.h, input.h, name.encode() if name else None), self)

    def dot_interaction(self, input: Tensor, d: int, name=None) -> Tensor:
        """[batch][c * d] (concat of the bottom-MLP output and the embedding outputs) -> [batch][d + c (c - 1) / 2]:
        row 0 passed through, then the pairwise dot products i > j -- the whole interaction in one launch each way."""
        return Tensor(lib().flexflow_model_add_dot_interaction(self.h, input.h, d, name.encode() if name else None), self)

    def tril(self, input: Tensor, name=None) -> Tensor:
        """Strict lower triangle of [batch][n][n] -> [batch][n (n - 1) / 2] (MLPerf-DLRM's pick of the pairwise dots)."""
        return Tensor(lib().flexflow_model_add_tril(self.h, input.h, name.encode() if name else None), self)

    def transpose(self, input: Tensor, perm, name=None) -> Tensor:
        p = (C.c_int * len(perm))(*perm)
        return Tensor(lib().flexflow_model_add_transpose(self.h, input.h, len(perm), p, name.encode() if name else None), self)

    def reshape(self, input: Tensor, shape, name=None) -> Tensor:
        p = (C.c_int * len(shape))(*shape)
        return Tensor(lib().flexflow_model_add_reshape(self.h, input.h, len(shape), p, name.encode() if name else None), self)

    @staticmethod
    def uniform_initializer(seed, lo, hi): return lib().flexflow_uniform_initializer_create(seed, lo, hi)

    @staticmethod
    def norm_initializer(seed, mean, std): return lib().flexflow_norm_initializer_create(seed, mean, std)

    @staticmethod
    def zero_initializer(): return lib().flexflow_zero_initializer_create()

    def set_sgd_optimizer(self, lr=0.01, momentum=0.0, nesterov=False, weight_decay=0.0):
        self._opt = lib().flexflow_sgd_optimizer_create(self.h, lr, momentum, nesterov, weight_decay)
        lib().flexflow_model_set_sgd_optimizer(self.h, self._opt)

    def set_adam_optimizer(self, alpha=0.001, beta1=0.9, beta2=0.999, weight_decay=0.0, epsilon=1e-8):
        """AdamOptimizer [ref: python/flexflow/core/flexflow_cffi.py AdamOptimizer; include/optimizer.h:62-85]"""
        self._opt = lib().flexflow_adam_optimizer_create(self.h, alpha, beta1, beta2, weight_decay, epsilon)
        lib().flexflow_model_set_adam_optimizer(self.h, self._opt)

    def compile(self, loss_type=LOSS_MSE_AVG, metrics=(METRICS_ACCURACY, METRICS_MSE), comp_mode=COMP_MODE_TRAINING):
        m = (C.c_int * len(metrics))(*metrics)
        lib().flexflow_model_compile(self.h, loss_type, m, len(metrics), comp_mode)

    # -- step ---------------------------------------------------------------------------------
    def init_layers(self): lib().flexflow_model_init_layers(self.h)
    def reset_metrics(self): lib().flexflow_model_reset_metrics(self.h)
    def forward(self, seq_length=-1): lib().flexflow_model_forward(self.h, seq_length)
    def zero_gradients(self): lib().flexflow_model_zero_gradients(self.h)
    def backward(self, seq_length=-1): lib().flexflow_model_backward(self.h, seq_length)
    def update(self): lib().flexflow_model_update(self.h)
    def begin_trace(self, trace_id): lib().flexflow_model_begin_trace(self.h, trace_id)
    def end_trace(self, trace_id): lib().flexflow_model_end_trace(self.h, trace_id)
    def sync(self): lib().flexflow_model_sync(self.h)

    def train_step(self):
        self.forward(); self.zero_gradients(); self.backward(); self.update()

    # -- inspection ---------------------------------------------------------------------------
    @property
    def label_tensor(self) -> Tensor: return Tensor(lib().flexflow_model_get_label_tensor(self.h), self)
    @property
    def num_layers(self) -> int: return lib().flexflow_model_get_num_layers(self.h)
    def layer_name(self, i) -> str: return lib().flexflow_model_get_layer_name(self.h, i).decode()
    def layer_num_weights(self, i) -> int: return lib().flexflow_model_get_layer_num_weights(self.h, i)
    def parameter(self, layer, index) -> Tensor: return Tensor(lib().flexflow_model_get_parameter(self.h, layer, index), self)
    def layer_output(self, layer) -> Tensor: return Tensor(lib().flexflow_model_get_layer_output(self.h, layer), self)
    @property
    def stream(self) -> int: return lib().flexflow_model_get_stream(self.h) or 0
    @property
    def uses_graph(self) -> bool: return bool(lib().flexflow_model_uses_graph(self.h))
    def set_trace_mode(self, mode: int): lib().flexflow_model_set_trace_mode(self.h, int(mode))
    def trace_replays(self, trace_id: int = 111) -> bool: return bool(lib().flexflow_model_trace_replays(self.h, trace_id))
    def counter(self, name: str) -> int: return int(lib().flexflow_model_get_counter(self.h, name.encode()))
    @property
    def backend(self) -> dict:
        """the kernel library this model loaded: {'name': ffh_backend_name(), 'path': file}"""
        return {"name": lib().flexflow_model_get_backend_name(self.h).decode(), "path": lib().flexflow_model_get_backend_path(self.h).decode()}

    def perf_metrics(self) -> PerfMetrics:
        p = PerfMetrics()
        lib().flexflow_model_get_perf_metrics(self.h, C.byref(p))
        return p

    def close(self):
        if self._owned and self.h is not None:
            lib().flexflow_model_destroy(self.h)
            self.h = None


class DLRM:
    """examples/cpp/DLRM as a library object: same flags as the reference driver."""

    def __init__(self, argv, comm: "FFComm | None" = None):
        full = ["dlrm"] + [str(a) for a in argv]
        self._comm = comm
        self.h = lib().flexflow_dlrm_create(len(full), _argv(full), C.byref(comm) if comm is not None else None)
        self.model = FFModel(_handle=lib().flexflow_dlrm_get_model(self.h))

    num_samples = property(lambda s: lib().flexflow_dlrm_get_num_samples(s.h))
    num_tables = property(lambda s: lib().flexflow_dlrm_get_num_tables(s.h))
    def sparse_input(self, t) -> Tensor: return Tensor(lib().flexflow_dlrm_get_sparse_input(self.h, t), self.model)
    def dense_input(self) -> Tensor: return Tensor(lib().flexflow_dlrm_get_dense_input(self.h), self.model)
    def warmup(self): lib().flexflow_dlrm_warmup(self.h)
    def train_steps(self, n, trace=True): lib().flexflow_dlrm_train_steps(self.h, n, trace)
    def run_epochs(self) -> float: return lib().flexflow_dlrm_run_epochs(self.h)
    def time_kernel(self, which, iters) -> float: return lib().flexflow_dlrm_time_kernel(self.h, which, iters)

    PROBE_PAIRS = ("gather", "table_update", "alltoall_fwd", "alltoall_bwd", "allreduce", "join_wait", "allreduce_wait") + tuple(f"bucket{i}" for i in range(8))

    def probe_step(self, iters) -> dict:
        """In-step event intervals (milliseconds, averaged over `iters` real eager steps): the side-stream gather (+ forward exchange)
        and table update (+ backward exchange), each collective alone, and the compute stream's wait for the embedding branch -- the
        exposed part of gather + exchange.  COLLECTIVE: with more than one rank every rank must call it."""
        out = (C.c_float * len(self.PROBE_PAIRS))()
        lib().flexflow_dlrm_probe_step(self.h, iters, out, len(self.PROBE_PAIRS))
        return {k: float(out[i]) for i, k in enumerate(self.PROBE_PAIRS)}

    def close(self):
        if self.h is not None:
            lib().flexflow_dlrm_destroy(self.h)
            self.h = None
