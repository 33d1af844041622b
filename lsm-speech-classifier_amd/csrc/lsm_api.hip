// Library-level entry points of liblsm_hip.so: version, thread-local error text, device query.
#include "lsm_common.h"

#include <cstdarg>

static thread_local char g_err[512] = "";

void lsm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define LSM_API extern "C" __attribute__((visibility("default")))

LSM_API int lsm_version(void) { return 101; }   // 0.1.1: + lsm_raster_pack_bits / lsm_raster_unpack_bits

LSM_API const char *lsm_last_error(void) { return g_err; }

// Number of visible HIP devices, or a negative error code.  The product path calls this first so
// that a box without a GPU (or without this library) fails loudly instead of falling back.
LSM_API int lsm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        lsm_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return LSM_ERR_HIP;
    }
    return n;
}
