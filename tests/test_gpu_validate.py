"""Mixture::validate (mixture.hpp:152-163,440-444) on the C ABI:
dist_gibbs_validate recounts sizes and integer statistics from the rows on the
device; dist_mixture_validate recomputes the value scorer's cache."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,mode", [("dd", 2), ("dd", 0), ("gp_nich", 0),
                                         ("dd_bb_gp", 0), ("bnb", 2),
                                         ("dpd", 2)])
def test_validate_accepts_every_state_a_sweep_leaves(config, mode):
    from distributions_amd import engine
    n, k = 6000, 40
    osh, gsh, vals, assign = workloads.make(config, n, k, seed=3)
    gpu = engine.Gibbs(2.0, 0.3, gsh)
    gpu.set_option("value_sorted", mode)
    gpu.load_rows(vals, assign, k, 2)
    report = gpu.validate()
    assert report["code"] == 0 and report["rows_assigned"] == n
    for sweep, batch in enumerate((1500, 777, n, 64)):
        gpu.sweep(0, n, batch, 99, draw_base=sweep * n)
        report = gpu.validate()      # closes the open run, then recounts
        assert report["code"] == 0 and report["rows_assigned"] == n
    state = gpu.sweep_sequential(100, 400, 12345)
    assert state and gpu.validate()["code"] == 0


def test_validate_reports_the_first_inconsistency():
    """statistics replaced by ones the rows do not back (import_stats_dev is
    the multi-GPU exchange's entry point: a wrong image is exactly what a
    broken exchange would leave)"""
    import torch
    from distributions_amd import engine
    n, k = 3000, 12
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=16, seed=1)
    gpu = engine.Gibbs(1.0, 0.0, gsh)
    gpu.load_rows(vals, assign, k, 1)
    assert gpu.validate()["code"] == 0
    words = gpu.core.stat_words()
    image = torch.zeros(words, dtype=torch.int32, device="cuda")
    gpu.core.export_stats_dev(image.data_ptr())
    K = len(gpu)
    good = image.clone()
    # [counts K | i0 K | i1 K | cnt K*dim]: one cell of group 5, value 3
    bad = good.clone()
    bad[3 * K + 5 * 16 + 3] += 1
    gpu.core.import_stats_dev(bad.data_ptr())
    report = gpu.validate(raise_on_failure=False)
    assert (report["code"], report["feature"], report["group"],
            report["detail"]) == (6, 0, 5, 3)
    assert report["found"] == report["expected"] + 1
    with pytest.raises(RuntimeError, match="validate: categorical count"):
        gpu.validate()
    # a group size
    bad = good.clone()
    bad[7] -= 2
    gpu.core.import_stats_dev(bad.data_ptr())
    report = gpu.validate(raise_on_failure=False)
    assert report["code"] in (3, 7) and report["group"] in (7, -1)
    gpu.core.import_stats_dev(good.data_ptr())
    assert gpu.validate()["code"] == 0


def test_mixture_validate():
    from distributions_amd.lp.models import dd, gp, nich
    for module in (dd, gp, nich):
        EXAMPLE = module.EXAMPLES[0]
        shared = module.Shared.from_dict(EXAMPLE['shared'])
        mixture = module.Mixture()
        for value in EXAMPLE['values']:
            mixture.append(module.Group.from_values(shared, [value]))
        mixture.init(shared)
        mixture._core.validate()
        for g, value in enumerate(EXAMPLE['values']):
            mixture.add_value(shared, (g + 1) % len(mixture), value)
            mixture._core.validate()
        mixture.remove_group(shared, 0)
        mixture.add_group(shared)
        mixture._core.validate()


def test_validate_during_sequential_initialisation():
    """rows without a group yet (load_rows_unassigned, then init_sequential
    over a prefix: examples/mixture/main.py:227-232) are not the recount's"""
    import oracle_lib as ol
    from distributions_amd import engine
    n = 2000
    osh, gsh, vals, assign = workloads.make("gp_nich", n, 1, seed=2)
    gpu = engine.Gibbs(1.0, 0.1, gsh)
    gpu.load_rows_unassigned(vals, empty_groups=1)
    report = gpu.validate()
    assert report["code"] == 0 and report["rows_assigned"] == 0
    state = ol.oracle().orc_rng_seed(5)
    state = gpu.init_sequential(0, 700, state)
    report = gpu.validate()
    assert report["code"] == 0 and report["rows_assigned"] == 700
    gpu.init_sequential(700, n, state)
    report = gpu.validate()
    assert report["code"] == 0 and report["rows_assigned"] == n
