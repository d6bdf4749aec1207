// context_internal.h — what group.cpp needs from a context beyond the C ABI
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/urt.h"

namespace urtd {
hipStream_t context_stream(urt_context* ctx);     // the stream the context currently issues its work on
int context_device(urt_context* ctx);
int context_pending_frames(urt_context* ctx);      // dispatches deferred by frame batching and not yet submitted
}  // namespace urtd
