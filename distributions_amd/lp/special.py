"""Mirror of distributions/lp/special.pyx: the table-driven special functions
of include/distributions/special.hpp, evaluated on the GPU."""
import numpy as np

from .. import _core


def _scalar_or_array(fn, x):
    a = np.atleast_1d(np.asarray(x))
    out = fn(a)
    return float(out[0]) if np.ndim(x) == 0 else out


def fast_log(x):
    return _scalar_or_array(_core.vector_log, x)


def fast_exp(x):
    return _scalar_or_array(_core.vector_exp, x)


def fast_lgamma(x):
    return _scalar_or_array(_core.vector_lgamma, x)


def fast_lgamma_nu(x):
    return _scalar_or_array(_core.vector_lgamma_nu, x)


def fast_log_factorial(n):
    return _scalar_or_array(_core.vector_log_factorial, n)
