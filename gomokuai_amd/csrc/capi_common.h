// capi_common.h -- shared state of the C-ABI translation units (error text, device tables).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/gomoku_hip.h"
#include "pattern_tables.h"

namespace gmk {

void set_error(const char* fmt, ...);

struct DeviceState {
    bool ready = false;
    int device = -1;
    int cu_count = 0;
    uint32_t* d_trans = nullptr;        // n_states*4 words, device form (DeviceTables::dev_trans)
    uint32_t* d_records = nullptr;      // 4 words per emission record (DeviceTables::dev_records), then kPrefixWords words of DeviceTables::dev_prefix4
    int n_states = 0, n_patterns = 0, emit_words = 0, n_records = 0;
};
DeviceState& device_state();

// Profiling switches (phase masks, in-kernel timers, launch-shape overrides) exist in the -DGMK_PROFILE build only:
// `python -m gomokuai_amd.build --profile` makes libgomoku_hip_prof.so, which tools/*.sh load with GMK_HIP_LIB=prof.
// The production library never reads the environment, so a stray variable cannot change a result.
#ifdef GMK_PROFILE
inline const char* profile_env(const char* name) { return std::getenv(name); }
constexpr bool kProfileBuild = true;
#else
inline const char* profile_env(const char*) { return nullptr; }
constexpr bool kProfileBuild = false;
#endif

#define GMK_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            gmk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GMK_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

}  // namespace gmk
