"""CPU tests of the drop-in boundary: both libraries export exactly the symbols
include/ff_hip.h declares, the ctypes prototypes cover all of them, and the product
loader refuses to run without its HIP library (no fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

from dlrm_flexflow_amd import capi


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_header_list_matches_prototypes():
    syms = capi.header_symbols()
    assert len(syms) == len(set(syms))
    assert set(syms) == set(capi._SIGS), set(syms) ^ set(capi._SIGS)
    # every function declared in the header body is in the X-macro list
    text = open(capi.HEADER_PATH).read()
    body = text.split("#define FFH_API_LIST")[0]
    declared = set(re.findall(r"\b(ffh_[a-z0-9_]+)\s*\(", body)) - {"ffh_ctx", "ffh_stream", "ffh_event", "ffh_graph"}
    assert declared == set(syms), declared ^ set(syms)


def test_hip_library_exports_every_symbol():
    from dlrm_flexflow_amd import build
    path = build.build_hip()
    assert os.path.exists(path)
    exp = _exported(path)
    missing = [s for s in capi.header_symbols() if s not in exp]
    assert not missing, missing
    lib = ctypes.CDLL(path)                      # loads without a GPU; no compute call is made
    lib.ffh_abi_version.restype = ctypes.c_int
    lib.ffh_backend_name.restype = ctypes.c_char_p
    assert lib.ffh_abi_version() == capi.header_abi_version()
    assert lib.ffh_backend_name() == b"hip-gfx950"
    # workspace sizing is host arithmetic: callable without a device
    lib.ffh_embedding_bwd_workspace_bytes.restype = ctypes.c_size_t
    lib.ffh_embedding_bwd_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
    n = lib.ffh_embedding_bwd_workspace_bytes(26, 1, 128, 32768)
    assert 26 * 32768 * 16 <= n < 64 << 20


def test_release_library_reads_no_environment_variable():
    """A/B switches exist only in -DFFH_LAB builds (tools/build_variant.sh): the shipped kernel library and the host shim's way to
    it import no getenv, so its behaviour is a function of the C-ABI arguments and the ctx setters alone."""
    from dlrm_flexflow_amd import build
    und = subprocess.check_output(["nm", "-D", "--undefined-only", build.build_hip()], text=True)
    assert not re.search(r"\b(secure_)?getenv\b", und), [l for l in und.splitlines() if "getenv" in l]


def test_oracle_library_exports_every_symbol(oracle):
    exp = _exported(oracle.ORACLE_LIB)
    missing = [s for s in capi.header_symbols() if s not in exp]
    assert not missing, missing
    assert oracle.lib().backend == "oracle-cpu"


def test_product_loader_fails_loudly_without_library(tmp_path, monkeypatch):
    monkeypatch.setattr(capi, "HIP_LIB_PATH", str(tmp_path / "libffhip.so"))
    monkeypatch.setattr(capi, "_hip_singleton", None)
    with pytest.raises(capi.FFHError):
        capi.load_hip()


def test_product_loader_rejects_the_oracle(oracle, monkeypatch):
    """Pointing the product loader at the CPU oracle must raise, not silently run on the CPU."""
    monkeypatch.setattr(capi, "HIP_LIB_PATH", oracle.ORACLE_LIB)
    monkeypatch.setattr(capi, "_hip_singleton", None)
    with pytest.raises(capi.FFHError):
        capi.load_hip()
    monkeypatch.setattr(capi, "_hip_singleton", None)


def test_oracle_stream_with_priority_is_a_no_op(oracle):
    """ABI 11 on the host twin: streams mean nothing there; the entry exists, checks its pointer and hands back the NULL stream."""
    be = oracle.lib()
    s = ctypes.c_void_p(1)
    assert be.lib.ffh_stream_create_with_priority(be.ctx, ctypes.byref(s), -1) == 0 and not s.value
    assert be.lib.ffh_stream_create_with_priority(be.ctx, None, 0) == -1
