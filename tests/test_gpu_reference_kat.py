"""The reference's own known-answer tests (tests/golden/ref_kat.json) replayed
through the HIP path via the C ABI."""
import pytest

from tests import _kat_cases
from tests._backends import GpuBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be(gpu):
    return GpuBackend()


@pytest.mark.parametrize("case", _kat_cases.ALL_CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_hip(be, case):
    case(be)
