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
