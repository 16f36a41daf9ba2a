"""Tools-only loader: the product kernel library reads no environment variable; the A/B switches (FFH_GEMM_CFG, FFH_SK_NO_SPLIT ...)
exist in lab builds only --
    tools/build_variant.sh tools/lab/libffhip_lab.so -DFFH_LAB
    FFH_TOOLS_LIB=tools/lab/libffhip_lab.so FFH_GEMM_CFG=3 python tools/gemm_big.py ...
    FFH_SK_NO_SPLIT=1 python bench.py --shim-flags="--backend tools/lab/libffhip_lab.so" ...
Without FFH_TOOLS_LIB the tools measure the product library (every switch at its default)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_flexflow_amd import capi  # noqa: E402


def load_hip(device=0):
    path = os.environ.get("FFH_TOOLS_LIB")
    if not path:
        return capi.load_hip(device)
    import torch  # noqa: F401  (one HIP runtime per process)
    lib = capi.FFHLib(os.path.abspath(path), device)
    assert lib.backend.startswith("hip"), lib.backend
    print(f"[tools] kernel library: {path}", file=sys.stderr)
    return lib
