// Non-GEMM kernels (normalisation, scans, resampling, element-wise glue).
#pragma once
#include "common.h"

namespace rvcx {
}  // namespace rvcx
