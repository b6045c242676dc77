// engine_internal.h -- what multi.cpp (the multi-device front end) needs from an engine beyond the public C-ABI.
// Not installed, not exported to callers of include/hafgrasp.h.
#pragma once

#include "../../include/hafgrasp.h"
#include <hip/hip_runtime.h>

namespace haf {

// device copy of the roll records of the engine's last haf_score_rolls call: [n_clouds * roll_count] x 16 bytes, the
// layout of haf_roll_record (static_assert in engine.cpp)
const void *engine_records_dev(const haf_engine *e);
hipStream_t engine_stream(const haf_engine *e);
const haf_config *engine_config(const haf_engine *e);
void engine_set_error(haf_engine *e, const char *msg);

}  // namespace haf
