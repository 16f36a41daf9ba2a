"""AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer over the host layer (libffmodel.so, the dlrm driver, the rank
launcher) driving the CPU oracle as its kernel library: the sanitizer run this pool allows (GPU ASan is not available).
The script builds both with -fsanitize=address,undefined into gpurun_out/san (scratch) and runs the driver through the cat,
dot and dot-tril interactions, --profiling, the tensor-op + deterministic flags, strategy export / import and the launcher's
dry run; any report (leaks included) fails it."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_host_layer_and_oracle_under_asan_ubsan():
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no gcc")
    probe = subprocess.run(["gcc", "-fsanitize=address,undefined", "-x", "c", "-", "-o", "/dev/null"], input="int main(void){return 0;}",
                           capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("this gcc has no sanitizer runtime")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_cpu.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "sanitizer run clean" in r.stdout
    assert "runtime error" not in r.stdout and "ERROR: AddressSanitizer" not in r.stdout
