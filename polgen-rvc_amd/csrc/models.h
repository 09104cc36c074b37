// Model containers (weights packed in the slab) and their forward passes.
#pragma once
#include "ctx.h"
#include "layers.h"

namespace rvcx {

struct SynthModel { int dummy = 0; };
struct RmvpeModel { int dummy = 0; };
struct HubertModel { int dummy = 0; };
struct IndexData { int dummy = 0; };

}  // namespace rvcx
