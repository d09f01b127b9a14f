// HIP kernels of libdistributions_hip (gfx950).  Included once, by
// dist_hip.hip.  Layout and roofline notes per kernel are in DESIGN.md.
#pragma once

#include <type_traits>

#include "models.h"

namespace dist {

constexpr int kBlock = 256;
constexpr int kMaxF = DIST_MAX_FEATURES;

}  // namespace dist

// (split by path; the order matters: later parts use the earlier ones)
#include "kernels_api.h"
#include "kernels_rows.h"
#include "kernels_vs.h"
#include "kernels_apply.h"
