"""Builds the in-tree native artefacts with explicit hipcc / g++ commands (no JIT cache):

  dlrm_flexflow_amd/csrc/libffhip.so   hand-written gfx950 HIP kernels behind include/ff_hip.h
  dlrm_flexflow_amd/host/libffmodel.so C++ FFModel shim (reference operator API) over that C-ABI
  dlrm_flexflow_amd/host/dlrm          the examples/cpp/DLRM driver re-stated on the shim

hipcc cross-compiles gfx950 without a GPU; the .so files travel to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
INCLUDE = os.path.join(ROOT, "include")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
             "-Wall", "-Wno-unused-function", "-I", INCLUDE]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    if r.stdout.strip():
        sys.stderr.write(r.stdout)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    out = os.path.join(CSRC, "libffhip.so")
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    if not force and not _newer(out, srcs + hdrs):
        return out
    objs = [s[:-4] + ".o" for s in srcs]

    def compile_one(pair):
        s, o = pair
        if force or _newer(o, [s] + hdrs):
            _run([HIPCC, *HIP_FLAGS, "-c", s, "-o", o])
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    if verbose:
        print("built", out)
    return out


def build_host(force: bool = False, verbose: bool = False) -> list[str]:
    """C++ FFModel shim + DLRM driver (plain g++; they reach HIP only through the C-ABI)."""
    if not os.path.isdir(HOST):
        return []
    mk = os.path.join(HOST, "Makefile")
    if not os.path.exists(mk):
        return []
    cmd = ["make", "-C", HOST, "-j4"] + (["-B"] if force else [])
    _run(cmd)
    return [os.path.join(HOST, f) for f in ("libffmodel.so", "dlrm") if os.path.exists(os.path.join(HOST, f))]


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
