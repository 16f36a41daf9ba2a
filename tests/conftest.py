import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def hip():
    """The product library on cuda:0.  No fallback: a missing/unsupported GPU build is an error."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test on a box without a GPU"
    from dlrm_flexflow_amd import capi
    return capi.load_hip(0)


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: the exact-mode parity tests that go through ffh_linear_* run a second time with the context in the fp32-accurate
# split mode (FFH_MATH_FP32_SPLIT_BF16X3_ALL: every layer the split kernels can take, whatever its size), AT THE SAME BOUND --
# the tests' own 1e-5 of the term mass; no second tolerance.  What a test asserts about the route an fp32 kernel took is
# skipped in that run (conftest.exact_routes(hip)); everything numeric is not.  Not in the list: tests of one-shot REQUESTS that only the
# persistent fp32 kernels serve (the column map and the column sums of the data-gradient epilogue: include/ff_hip.h says a route may
# decline them, and the split kernels do -- the exchange step in this mode is test_gpu_round3.py::test_exchange_path_in_the_bf16_pipe_math_modes).
SPLIT_MODE_TESTS = {
    "test_gpu_parity.py": {"test_linear_torch_golden", "test_linear_vs_oracle", "test_linear_strided_operands_and_accumulate",
                           "test_linear_bwd_ex_forms_equal_reference_form", "test_linear_lds_dma_kernel_shapes", "test_linear_gelu_forward",
                           "test_linear_bwd_mse_equals_the_two_calls", "test_linear_pair_fwd_equals_the_two_calls",
                           "test_linear_pair_bwd_equals_the_two_calls"},
    "test_gpu_round3.py": {"test_linear_layers_of_the_benched_step_at_b32768_vs_oracle", "test_reference_harness_linear_20_5000_5000"},
    "test_gpu_round4.py": {"test_linear_layers_at_the_per_rank_batch_vs_oracle_with_routes",
                           "test_narrow_layer_backward_with_partial_rows_and_last_arriver"},
    "test_gpu_round5.py": {"test_64_row_tiles_of_the_persistent_gemm_vs_oracle", "test_raw_c_abi_call_with_a_padded_reduction_depth_takes_the_fast_path"},
}
MATH_SPLIT_ALL = 3      # FFH_MATH_FP32_SPLIT_BF16X3_ALL (include/ff_hip.h)


def pytest_generate_tests(metafunc):
    names = SPLIT_MODE_TESTS.get(os.path.basename(str(metafunc.definition.fspath)))
    if names and metafunc.function.__name__ in names:
        metafunc.parametrize("_ffh_math_mode", [0, MATH_SPLIT_ALL], ids=["fp32", "split"], indirect=True)


@pytest.fixture(autouse=True)
def _ffh_math_mode(request):
    mode = getattr(request, "param", 0)
    if not mode:
        yield 0
        return
    lib = request.getfixturevalue("hip")
    assert lib.lib.ffh_ctx_set_math_mode(lib.ctx, mode) == 0
    lib.test_math_mode = mode
    try:
        yield mode
    finally:
        lib.test_math_mode = 0
        assert lib.lib.ffh_ctx_set_math_mode(lib.ctx, 0) == 0


def exact_routes(lib):
    """True when the calls of this test run on the exact-fp32 kernels (route tokens of those kernels can be asserted)."""
    return getattr(lib, "test_math_mode", 0) == 0
