// placeholder: filled in below
#include "capi_common.h"
extern "C" int gmk_eval_batch(const uint16_t*, int, int32_t*, int32_t*, uint32_t*, int32_t*, void*) { gmk::set_error("not built yet"); return GMK_ERR_STATE; }
extern "C" int gmk_eval_batch_host(const uint16_t*, int, int32_t*, int32_t*, uint32_t*, int32_t*) { gmk::set_error("not built yet"); return GMK_ERR_STATE; }
extern "C" int gmk_eval_launch_info(int, int*, int*, int*) { return GMK_ERR_STATE; }
