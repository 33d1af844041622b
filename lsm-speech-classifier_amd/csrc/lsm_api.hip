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

// major*10000 + minor*100 + patch of the package's __version__: build.py passes it (and the identity of the sources the
// library is built from) on this file's command line, so there is ONE place where the number is written; _lib.load()
// refuses a library whose version or build id is not the tree's.
#ifndef LSM_VERSION_NUMBER
#error "build through lsm-speech-classifier_amd/build.py: it defines LSM_VERSION_NUMBER and LSM_BUILD_ID_STRING"
#endif
LSM_API int lsm_version(void) { return LSM_VERSION_NUMBER; }
// "LSM_BUILD_ID=<24 hex digits>": hash of every source and header, the flags and the version (build.source_id)
LSM_API const char *lsm_build_id(void) { return LSM_BUILD_ID_STRING; }

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
