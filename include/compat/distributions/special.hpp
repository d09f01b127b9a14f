// Forwarding header: a program written against the reference's
// <distributions/...> headers for the mixture row-update path picks up
// include/distributions_hip.hpp instead when include/compat comes first on
// its include path.
#pragma once
#ifndef DISTRIBUTIONS_HIP_AS_DISTRIBUTIONS
#define DISTRIBUTIONS_HIP_AS_DISTRIBUTIONS 1
#endif
#include "../../distributions_hip.hpp"
