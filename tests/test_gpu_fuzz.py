"""Random-shape sweep of the Linear kernels against the oracle (-m gpu): tools/fuzz_linear.py draws shapes, strides,
activations and ffh_linear_bwd_ex flag / stream combinations so that every kernel family of linear.hip is hit."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_linear_random_shapes_agree_with_oracle(hip, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_linear.py"), "30", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "random cases agree with the oracle" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode", [1, 3])
def test_linear_random_shapes_in_the_bf16_pipe_math_modes(hip, mode):
    """mode 1: tensor-op bf16 operands, oracle in the same mode; mode 3: fp32-accurate bf16x3 split on every shape its kernels accept against the fp32
    oracle (mode 2 would leave these small GEMMs to the fp32 kernels), half of the cases with three-plane images registered for operands and results."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_linear.py"), "25", str(20 + mode), str(mode)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"random cases agree with the oracle (math mode {mode})" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", [31, 32])
def test_embedding_random_tables_agree_with_oracle_bit_for_bit(hip, seed):
    """tools/fuzz_embedding.py: 1-12 tables per call, 1 to 300 000 rows, bags of 1-4, widths 1-256, batches on both sides of the
    one-launch limit, SUM / AVG, id distributions up to "every lookup hits one row"; gather and fused update bit for bit."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_embedding.py"), "40", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "random cases agree with the oracle bit for bit" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
