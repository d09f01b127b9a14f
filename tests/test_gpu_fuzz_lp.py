"""A slice of tools/fuzz_lp.py in the suite: the per-row API (Shared / Group /
Mixture of every model) under random hyper-parameters and random operation
sequences, against the oracle (bit-exact)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [0, 500])
def test_per_row_api_fuzz(first):
    import fuzz_lp
    failures = [err for err in (fuzz_lp.trial(seed)
                                for seed in range(first, first + 200)) if err]
    assert not failures, failures


def test_growing_dpd_shared_fuzz():
    """tools/fuzz_lp.py's second kind of trial: values appear in and vanish
    from a DirichletProcessDiscrete Shared (dpd.hpp:66-83) under a live device
    mixture"""
    import fuzz_lp
    failures = [err for err in (fuzz_lp.trial_growing_dpd(seed)
                                for seed in range(4000, 4040)) if err]
    assert not failures, failures
