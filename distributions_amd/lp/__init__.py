"""Mirror of the reference's `distributions.lp` package (Cython wrappers of the
C++ library, distributions/lp/) over libdistributions_hip:

    lp.models.{dd,bb,gp,nich,dpd}   Shared, Group, Mixture, EXAMPLES, NAME, Value
    lp.clustering                   PitmanYor (+ .Mixture), count_assignments
    lp.mixture                      MixtureIdTracker
    lp.random, lp.special           entropy, discrete sampling, special functions
"""
