// capi_common.h -- shared state of the C-ABI translation units (error text, device tables).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/gomoku_hip.h"
#include "pattern_tables.h"

namespace gmk {

void set_error(const char* fmt, ...);

struct DeviceState {
    bool ready = false;
    int device = -1;
    int cu_count = 0;
    uint32_t* d_trans = nullptr;        // n_states*4 words, device form (DeviceTables::dev_trans)
    uint32_t* d_records = nullptr;      // 4 words per emission record (DeviceTables::dev_records)
    int n_states = 0, n_patterns = 0, emit_words = 0, n_records = 0;
};
DeviceState& device_state();

#define GMK_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            gmk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GMK_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

}  // namespace gmk
