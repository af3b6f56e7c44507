"""Pins the oracle (oracle/adamml_oracle.py) against golden vectors captured from the
real reference by tools/gen_golden.py (CPU fp32).  Runs without GPU and without
/root/reference."""
import pytest

from tests.golden_cases import CASES
from tests.oracle_harness import oracle_case, load_golden, compare_records


@pytest.mark.parametrize("name", list(CASES.keys()))
def test_oracle_matches_reference_golden(name):
    got = oracle_case(CASES[name])
    compare_records(got, load_golden(name))
