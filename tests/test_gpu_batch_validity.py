"""tests/test_batch_validity.py on the engine itself, at ten times the rows:
the exact sequential chain (sweep_sequential) against batch chains of
B = 4096, N/10, N/3 and N rows on the planted mixture, N = 100 000.  The
joint score comes from the product path (SlaveMixture.score_data over the
engine's groups + py_score_counts)."""
import numpy as np
import pytest

import workloads
from test_batch_validity import ALPHA, D, sweeps_to_reach

pytestmark = pytest.mark.gpu


def joint_score(gpu, gsh):
    from distributions_amd import _core
    total = 0.0
    for f, sh in enumerate(gsh):
        m = _core.SlaveMixture(sh)
        for g in range(len(gpu)):
            m.append(gpu.get_group(f, g))
        m.init()
        total += m.score_data()
    return total + _core.py_score_counts(ALPHA, D, gpu.counts())


def run_chain(gsh, vals, start, k0, batch, sweeps, seed=7):
    from distributions_amd import _core, engine
    n = len(start)
    gpu = engine.Gibbs(ALPHA, D, gsh)
    gpu.load_rows(vals, start, k0, 1)
    st = _core.rng_seed(seed)
    traj = [joint_score(gpu, gsh) / n]
    for s in range(sweeps):
        if batch == 0:
            st = gpu.sweep_sequential(0, n, st)
        else:
            gpu.sweep(0, n, batch, seed, draw_base=s * n)
        traj.append(joint_score(gpu, gsh) / n)
    return np.array(traj), gpu.assignments().copy()


def test_batch_chains_reach_the_sequential_plateau_on_the_engine():
    n, k, sweeps = 100_000, 64, 12
    truth, _, gsh, vals = workloads.planted(n, k)
    start = (np.arange(n) % k).astype(np.uint32)
    seq, seq_assign = run_chain(gsh, vals, start, k, 0, sweeps)
    ari_seq = workloads.adjusted_rand_index(truth, seq_assign)
    assert ari_seq > 0.8
    gain = seq[-1] - seq[0]
    assert gain > 5.0
    level = seq[0] + 0.9 * gain
    s_seq = sweeps_to_reach(seq, level)
    report = ["sequential: ARI %.3f, %d sweeps to 90%%: %s" % (
        ari_seq, s_seq, np.round(seq, 2))]
    # (B, sweeps, allowed lag in sweeps): B/N = 0.04, 0.1, 0.33, 1
    for batch, n_sweeps, lag in [(4096, sweeps, 1), (10_000, sweeps, 2),
                                 (32_768, sweeps + 4, 4), (n, sweeps + 12, 9)]:
        traj, assign = run_chain(gsh, vals, start, k, batch, n_sweeps)
        ari = workloads.adjusted_rand_index(truth, assign)
        report.append("B=%d: ARI %.3f, %d sweeps to 90%%: %s" % (
            batch, ari, sweeps_to_reach(traj, level), np.round(traj, 2)))
        assert traj[-3:].mean() > seq[-3:].mean() - 0.05 * gain, report
        assert traj[-3:].mean() < seq[-3:].mean() + 0.10 * gain, report
        assert ari > ari_seq - 0.1, report
        assert sweeps_to_reach(traj, level) <= s_seq + lag, report
    print("\n".join(report))
