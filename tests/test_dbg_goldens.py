"""The oracle against the REFERENCE'S OWN Python flavour.

random.hpp includes Eigen, so the reference's C++ scorers do not compile in
this image and oracle.c's model arithmetic cannot be compared with compiled
reference code.  The reference also ships every model as plain Python
(distributions/dbg/models/*.py, dbg/clustering.py) and tests its C++ flavour
against that one at TOL = 1e-3 (distributions/tests/test_model_flavors.py:61-116).
tests/golden/make_dbg_goldens.py ran those modules where they lie and dumped
their answers; here the oracle takes the lp flavour's seat in that test:

    group statistics after every add / remove   == dbg's Group.dump()
    Group.score_value (scalar, Scorer)           ~  dbg's, TOL
    Mixture.score_value_group / score_value      ~  dbg's, TOL (the vectorised
                                                    cache the row update reads)
    Group.score_data / Mixture.score_data        ~  dbg's, TOL
    LowEntropy score_add_value / score_remove_value / score_counts /
    log_partition_function                       ~  dbg's, TOL

tests/test_gpu_dbg_goldens.py checks the HIP library against the same files.
"""
import ctypes

import numpy as np
import pytest

import dbg_fixtures as fx
import oracle_lib as ol

EPS_LOG = float(np.log1p(2.0 ** -14))   # FastLog truncates the mantissa to 14
                                         # bits (special.hpp:57-67): |error|
                                         # of one fast_log <= log(1 + 2^-14)


def table_allowance(name, scen, words, grid):
    """Where the reference's C++ flavour ITSELF leaves its Python flavour by
    more than TOL, by construction: its lookup tables, which the oracle holds
    bit for bit (tests/test_oracle_golden.py against the compiled
    special.cc).  Two places, both outside what test_model_flavors visits
    (it scores once, after a handful of values):
      nich  fast_lgamma_nu's cubic per two octaves (special.hpp:239-273) is
            off by up to 0.013 (at nu = 1): allow exactly its measured error
            at this group's nu' = nu + count; and the row term is
            (-nu'/2 - 1/2) * fast_log(...) + 1/2 fast_log(...) (nich.hpp:239-
            250, nich.cc:60-66): allow (nu'/2 + 1) * EPS_LOG;
      gp    score = -lgamma(a) + a (fast_log(b) - fast_log(1+b)) + ... - x
            fast_log(1+b) (gp.hpp:198-217, gp.cc:57-66) multiplies truncated
            logarithms by a = alpha + sum, thousands here: allow
            (2a + x) * EPS_LOG.
    Everything else gets no allowance."""
    from scipy.special import gammaln
    raw = scen["shared"]
    i32 = words.view(np.int32)
    if name == "nich":
        nu = float(np.float32(raw["nu"])) + float(i32[0])
        exact = gammaln(0.5 * nu + 0.5) - gammaln(0.5 * nu)
        err = abs(float(ol.oracle().orc_fast_lgamma_nu(nu)) - exact)
        return np.full(len(grid), err + (0.5 * nu + 1.0) * EPS_LOG)
    if name == "gp":
        a = raw["alpha"] + float(i32[1])
        return np.array([(2.0 * a + float(x)) * EPS_LOG for x in grid])
    return np.zeros(len(grid))


KIND = {"dd": ol.DD, "bb": ol.BB, "gp": ol.GP, "nich": ol.NICH,
        "dpd": ol.DPD, "bnb": ol.BNB}


def oracle_shared(name, scen):
    raw = scen["shared"]
    if name == "dpd":
        keys, index, betas = fx.dpd_dense(raw)
        sh = ol.make_shared(ol.DPD, alpha=raw["alpha"], betas=betas,
                            beta0=scen["beta0"])
        return sh, (lambda v: fx.OTHER if v == fx.OTHER else index[int(v)])
    kind = KIND[name]
    sh = ol.make_shared(kind, **raw)
    return sh, (lambda v: int(ol.value_words(kind, [v])[0]))


def check_group_dump(name, scen, words, dump, msg):
    """the statistics image against dbg's Group.dump()"""
    i32 = words.view(np.int32)
    f32 = words.view(np.float32)
    if name == "dd":
        assert i32[1:].tolist() == dump["counts"], msg
        assert i32[0] == sum(dump["counts"]), msg
    elif name == "dpd":
        keys, index, _ = fx.dpd_dense(scen["shared"])
        want = np.zeros(len(keys), np.int64)
        for k, c in dump["counts"].items():
            want[index[int(k)]] = c
        assert i32[1:].tolist() == want.tolist(), msg
        assert i32[0] == want.sum(), msg
    elif name == "bb":
        assert (i32[0], i32[1]) == (dump["heads"], dump["tails"]), msg
    elif name == "bnb":
        assert (i32[0], i32[1]) == (dump["count"], dump["sum"]), msg
    elif name == "gp":
        assert (i32[0], i32[1]) == (dump["count"], dump["sum"]), msg
        fx.assert_close(f32[2], dump["log_prod"], msg + " log_prod")
    elif name == "nich":
        assert i32[0] == dump["count"], msg
        fx.assert_close(f32[1], dump["mean"], msg + " mean")
        # count_times_variance is a difference of large terms after removes:
        # compare on the scale of the sum of squares it came from
        fx.assert_close(f32[2], dump["count_times_variance"], msg + " ctv")


@pytest.mark.parametrize("name,index", fx.scenario_ids())
def test_oracle_models_follow_the_dbg_flavour(name, index):
    L = ol.oracle()
    scen = fx.models()[name]["scenarios"][index]
    sh, word = oracle_shared(name, scen)
    m = ol.OracleMixture(1.0, 0.0, [sh])
    L.orc_mix_slave_append_empty(m.h, 0)
    L.orc_mix_slave_init(m.h, 0)
    grid = scen["grid"]
    for t, step in enumerate(scen["steps"]):
        msg = "%s[%d] step %d (%s %r)" % (name, index, t, step["op"],
                                          step["value"])
        if step["op"] == "add":
            L.orc_mix_slave_add_value(m.h, 0, 0, word(step["value"]))
        elif step["op"] == "remove":
            L.orc_mix_slave_remove_value(m.h, 0, 0, word(step["value"]))
        words = m.get_group(0, 0)
        if "group" in step:
            check_group_dump(name, scen, words, step["group"], msg)
        allow = table_allowance(name, scen, words, grid)
        scalar = [L.orc_group_score_value(ctypes.byref(sh), words, word(v))
                  for v in grid]
        fx.assert_close(scalar, step["score_value"],
                        msg + " Group.score_value", allow)
        cached = [L.orc_mix_slave_score_value_group(m.h, 0, 0, word(v))
                  for v in grid]
        fx.assert_close(cached, step["score_value"],
                        msg + " Mixture.score_value_group", allow)
        accum = []
        for v in grid:
            acc = np.zeros(1, np.float32)
            L.orc_mix_slave_score_value(m.h, 0, word(v), acc)
            accum.append(acc[0])
        fx.assert_close(accum, step["score_value"],
                        msg + " Mixture.score_value", allow)
        fx.assert_close(L.orc_group_score_data(ctypes.byref(sh), words),
                        step["score_data"], msg + " Group.score_data")
        fx.assert_close(L.orc_mix_slave_score_data(m.h, 0),
                        step["score_data"], msg + " Mixture.score_data")


def _le_sigs(L):
    i = ctypes.c_int
    L.orc_le_score_add_value.restype = ctypes.c_float
    L.orc_le_score_add_value.argtypes = [i, i, i, i, i]
    L.orc_le_score_remove_value.restype = ctypes.c_float
    L.orc_le_score_remove_value.argtypes = [i, i, i, i, i]
    L.orc_le_score_counts.restype = ctypes.c_float
    L.orc_le_score_counts.argtypes = [i, ctypes.c_void_p, ctypes.c_size_t]
    L.orc_le_log_partition_function.restype = ctypes.c_float
    L.orc_le_log_partition_function.argtypes = [i]


@pytest.mark.parametrize("case", range(6))
def test_oracle_low_entropy_follows_the_dbg_flavour(case):
    """Clustering::LowEntropy (clustering.hpp:245-331, clustering.cc:185-283)
    against dbg/clustering.py:148-300"""
    L = ol.oracle()
    _le_sigs(L)
    c = fx.low_entropy()[case]
    N = c["dataset_size"]
    # clustering.hpp:281-288 multiplies fast_log(bigger / group_size) by
    # group_size up to 10 000: the C++ flavour leaves the Python one by up to
    # group_size * EPS_LOG there (0.39 at 9 999) -- allowed, nothing else is
    for size, nonempty, sample, empties, want in c["score_add_value"]:
        got = L.orc_le_score_add_value(N, size, nonempty, sample, empties)
        allow = size * EPS_LOG if size <= 10000 else 0.0
        fx.assert_close(got, want, "score_add_value(%d,%d,%d,%d) N=%d" % (
            size, nonempty, sample, empties, N), allow)
    for size, nonempty, sample, empties, want in c["score_remove_value"]:
        got = L.orc_le_score_remove_value(N, size, nonempty, sample, empties)
        allow = (size - 1) * EPS_LOG if size - 1 <= 10000 else 0.0
        fx.assert_close(got, want, "score_remove_value(%d,%d) N=%d" % (
            size, sample, N), allow)
    for counts, want in c["score_counts"]:
        arr = np.ascontiguousarray(counts, np.int32)
        got = L.orc_le_score_counts(N, arr.ctypes.data, arr.size)
        fx.assert_close(got, want, "score_counts(%r) N=%d" % (counts, N))
    for n, want in c["log_partition_function"]:
        fx.assert_close(L.orc_le_log_partition_function(n), want,
                        "log_partition_function(%d)" % n)
