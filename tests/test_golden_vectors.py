"""Committed golden vectors (tests/golden/vectors.npz, generated from the spec model by
tests/golden/make_vectors.py): the C oracle reproduces them on the CPU, the HIP path on the GPU."""
import pytest

from tests import _golden_cases
from tests._backends import GpuBackend, OracleBackend


def test_oracle_reproduces_golden_vectors():
    _golden_cases.run(OracleBackend())


@pytest.mark.gpu
def test_hip_reproduces_golden_vectors(gpu):
    _golden_cases.run(GpuBackend())
