"""Pin the CPU oracle: every known-answer test the reference holds for the hot
path (tests/golden/ref_kat.json) must pass on `oracle/idsp_oracle.c`, and the
second restatement (`oracle/spec.py`) must agree with it."""
import numpy as np
import pytest

from tests import _kat_cases
from tests._backends import OracleBackend


@pytest.fixture(scope="module")
def be():
    return OracleBackend()


@pytest.mark.parametrize("case", _kat_cases.ALL_CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_oracle(be, case):
    case(be)


def test_cossin_table_matches_survey_and_generator(be):
    from oracle import spec
    from tools.gen_cossin_table import table

    t = be.o.cossin_table()
    s = _kat_cases.KAT["survey_checksums"]
    assert t[:4] == s["lut_first4"] and t[127] == s["lut_127"]
    assert t == spec.cossin_table() == table()
