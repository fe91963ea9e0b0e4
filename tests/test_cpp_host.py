"""The C++ host layer (include/idsp_hip.hpp): it must compile and link against
the C ABI with plain g++ (CPU check), and its reference-style test program
(tests/cpp/test_host.cpp) must pass on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    subprocess.run(["make", "-s", "build/test_host"], cwd=ROOT, check=True)
    return os.path.join(ROOT, "build", "test_host")


def test_cpp_host_layer_compiles_and_links():
    exe = _build()
    assert os.path.exists(exe)


def test_cpp_coefficient_front_end_reference_tests():
    """Filter / pid::Builder / Pid / build_config of idsp_hip.hpp: host code, no GPU needed."""
    subprocess.run(["make", "-s", "build/test_coeff"], cwd=ROOT, check=True)
    r = subprocess.run([os.path.join(ROOT, "build", "test_coeff")], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all coefficient front-end tests passed" in r.stdout


@pytest.mark.gpu
def test_cpp_host_layer_reference_tests(gpu):
    exe = _build()
    r = subprocess.run([exe], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host-layer tests passed" in r.stdout
