"""Minimal HDF5 reader/writer over libhdf5's C API (ctypes) -- this image has libhdf5 but no h5py.
Enough for the reference's Criteo file (2-D / 1-D float32 and int64 datasets at the root group)
[ref: examples/cpp/DLRM/preprocess_hdf.py:14-24].  The C++ loader (host/hdf5_io.cc) opens the same
library with dlopen; `FFH_HDF5_LIB` overrides the search for both."""
import ctypes as C
import os

import numpy as np

_CANDIDATES = ("libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5_serial.so.103", "libhdf5.so.200", "libhdf5.so.310",
               "/opt/conda/lib/libhdf5.so")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    names = (os.environ["FFH_HDF5_LIB"],) if os.environ.get("FFH_HDF5_LIB") else _CANDIDATES
    err = None
    for n in names:
        try:
            l = C.CDLL(n)
            break
        except OSError as e:
            err = e
    else:
        raise RuntimeError(f"libhdf5 not found (tried {names}): {err}")
    hid = C.c_int64
    sig = {
        "H5open": (C.c_int, []), "H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fopen": (hid, [C.c_char_p, C.c_uint, hid]),
        "H5Fclose": (C.c_int, [hid]), "H5Screate_simple": (hid, [C.c_int, C.c_void_p, C.c_void_p]), "H5Sclose": (C.c_int, [hid]),
        "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, C.c_char_p, hid]),
        "H5Dwrite": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), "H5Dread": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
        "H5Dclose": (C.c_int, [hid]), "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]), "H5Tget_class": (C.c_int, [hid]),
        "H5Tget_size": (C.c_size_t, [hid]), "H5Tclose": (C.c_int, [hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid]), "H5Sget_simple_extent_dims": (C.c_int, [hid, C.c_void_p, C.c_void_p]),
    }
    for k, (res, args) in sig.items():
        f = getattr(l, k)
        f.restype, f.argtypes = res, args
    if l.H5open() < 0:
        raise RuntimeError("H5open failed")
    l._types = {np.dtype(np.float32): hid.in_dll(l, "H5T_NATIVE_FLOAT_g").value, np.dtype(np.int64): hid.in_dll(l, "H5T_NATIVE_LLONG_g").value,
                np.dtype(np.float64): hid.in_dll(l, "H5T_NATIVE_DOUBLE_g").value, np.dtype(np.int32): hid.in_dll(l, "H5T_NATIVE_INT_g").value}
    _lib = l
    return l


def write(path: str, datasets: dict) -> None:
    """Create `path` (truncating) with one dataset per item; dtypes float32 / float64 / int32 / int64."""
    l = lib()
    f = l.H5Fcreate(os.fsencode(path), 2, 0, 0)   # H5F_ACC_TRUNC
    if f < 0:
        raise OSError(f"H5Fcreate({path}) failed")
    try:
        for name, a in datasets.items():
            a = np.ascontiguousarray(a)
            t = l._types[a.dtype]
            dims = (C.c_ulonglong * a.ndim)(*a.shape)
            sp = l.H5Screate_simple(a.ndim, dims, None)
            d = l.H5Dcreate2(f, name.encode(), t, sp, 0, 0, 0)
            if d < 0 or l.H5Dwrite(d, t, 0, 0, 0, a.ctypes.data) < 0:
                raise OSError(f"writing dataset {name} failed")
            l.H5Dclose(d); l.H5Sclose(sp)
    finally:
        l.H5Fclose(f)


def read(path: str, name: str) -> np.ndarray:
    """Whole dataset as float32 (float class) or int64 (integer class)."""
    l = lib()
    f = l.H5Fopen(os.fsencode(path), 0, 0)
    if f < 0:
        raise OSError(f"H5Fopen({path}) failed")
    try:
        d = l.H5Dopen2(f, name.encode(), 0)
        if d < 0:
            raise KeyError(name)
        sp, ty = l.H5Dget_space(d), l.H5Dget_type(d)
        nd = l.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_ulonglong * max(nd, 1))()
        l.H5Sget_simple_extent_dims(sp, dims, None)
        dt = np.dtype(np.float32) if l.H5Tget_class(ty) == 1 else np.dtype(np.int64)
        out = np.empty(tuple(dims[:nd]), dt)
        rc = l.H5Dread(d, l._types[dt], 0, 0, 0, out.ctypes.data)
        l.H5Tclose(ty); l.H5Sclose(sp); l.H5Dclose(d)
        if rc < 0:
            raise OSError(f"reading dataset {name} failed")
        return out
    finally:
        l.H5Fclose(f)
