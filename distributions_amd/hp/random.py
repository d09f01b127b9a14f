"""distributions/hp/random.pyx:52-53: `seed(s)` seeds THE process-global
engine that the lp layer draws from (rng.py:37-47, global_rng.pyx:32-33)."""
from ..lp.random import get_rng, seed  # noqa: F401
