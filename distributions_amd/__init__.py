"""distributions_amd -- MI355X (gfx950) implementation of the collapsed-Gibbs
mixture hot path of forcedotcom/distributions, behind the reference's
Shared / Group / Mixture interface.

    distributions_amd.lp       mirror of distributions.lp (models, clustering,
                               mixture, random, special)
    distributions_amd.engine   the batched row engine and its multi-GPU driver

All arithmetic runs in libdistributions_hip.so; importing this package fails
if the library or the binding is missing (there is no CPU fallback).
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
if not _os.path.exists(_os.path.join(_here, "libdistributions_hip.so")):
    raise ImportError(
        "distributions_amd/libdistributions_hip.so is missing: run "
        "`python __graft_entry__.py` (hipcc, gfx950) first")

# PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.
# Two HIP runtimes in one process cannot both own the GPU, so whichever is
# loaded first must serve both: load torch's first when torch is installed
# (libdistributions_hip then binds to it by SONAME); without torch the
# system runtime under /opt/rocm is used.
try:  # noqa: E402
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover
    _torch = None

from . import _core  # noqa: E402,F401

__version__ = "0.1.0"
