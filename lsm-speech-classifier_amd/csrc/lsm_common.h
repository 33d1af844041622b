// Shared declarations for the HIP translation units of liblsm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define LSM_WAVE 64

// Error codes of the C ABI (include/lsm_hip.h).
#define LSM_OK 0
#define LSM_ERR_ARG -1
#define LSM_ERR_HIP -2
#define LSM_ERR_NOMEM -3
#define LSM_ERR_UNSUPPORTED -4

void lsm_set_error(const char *fmt, ...);

// Raise a kernel's dynamic-LDS limit to the CU's 160 KB, once per (function, device) for the life of the
// process: launch functions then make no runtime call but the launch itself (hipGraph-capture safe, and two
// threads launching the same kernel with different LDS sizes cannot lower each other's limit).
void lsm_allow_big_lds(const void *kernel_fn);

#define LSM_CHECK_HIP(expr)                                                         \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            lsm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                          __FILE__, __LINE__);                                      \
            return LSM_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)

#define LSM_REQUIRE(cond, ...)                                                      \
    do {                                                                            \
        if (!(cond)) {                                                              \
            lsm_set_error(__VA_ARGS__);                                             \
            return LSM_ERR_ARG;                                                     \
        }                                                                           \
    } while (0)
