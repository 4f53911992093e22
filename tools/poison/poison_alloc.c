/* LD_PRELOAD shim for diagnostic runs: every hipMalloc'ed block is filled with 0xCD, so a kernel that reads device memory it (or
 * the host) never wrote sees garbage instead of the zeros a fresh allocation usually happens to hold.
 * Build + use (on the GPU box):  tools/poison/run.sh python -m pytest tests -m gpu -x -q */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <stddef.h>
#include <hip/hip_runtime_api.h>

typedef hipError_t (*malloc_fn)(void**, size_t);

hipError_t hipMalloc(void** ptr, size_t size) {
    static malloc_fn real = NULL;
    if (!real) real = (malloc_fn)dlsym(RTLD_NEXT, "hipMalloc");
    hipError_t e = real(ptr, size);
    if (e == hipSuccess && size) (void)hipMemset(*ptr, 0xCD, size);
    return e;
}
