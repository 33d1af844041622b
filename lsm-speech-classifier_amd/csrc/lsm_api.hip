// Library-level entry points of liblsm_hip.so: version, thread-local error text, device query.
#include "lsm_common.h"

#include <cstdarg>
#include <mutex>
#include <set>
#include <utility>

static thread_local char g_err[512] = "";

void lsm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void lsm_allow_big_lds(const void *kernel_fn)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({kernel_fn, dev}).second &&
        hipFuncSetAttribute(kernel_fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        (void)hipGetLastError();        // the launch that follows reports the failure
}

#define LSM_API extern "C" __attribute__((visibility("default")))

// major*10000 + minor*100 + patch; in step with the package's __version__ and _lib.ABI_VERSION (which refuses another number)
LSM_API int lsm_version(void) { return 400; }   // 0.4.0

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
