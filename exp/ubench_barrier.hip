// Cost of __syncthreads() per iteration at 1..16 waves per workgroup, alone and with an LDS write+read
// between barriers (the minimum any multi-wave LIF step must do).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float buf[2048];
    buf[threadIdx.x] = threadIdx.x;
    buf[threadIdx.x + 1024] = 0.f;
    __syncthreads();
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) {
            buf[(i & 1) * 1024 + threadIdx.x] = acc + i;
        }
        __syncthreads();
        if (MODE == 1) {
            acc += buf[(i & 1) * 1024 + ((threadIdx.x + 64) % blockDim.x)];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    float *o; unsigned long long *c;
    CHECK(hipMalloc(&o, 256 * 1024 * 4)); CHECK(hipMalloc(&c, 8));
    const int iters = 2000;
    for (int waves : {1, 2, 4, 8, 16}) for (int mode = 0; mode < 2; ++mode) {
        unsigned long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), 0, 0, o, c, iters);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(waves * 64), 0, 0, o, c, iters);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost));
        printf("waves %2d %-28s %7.1f cycles per iteration\n", waves, mode ? "write + barrier + read" : "barrier only", (double)h / iters);
    }
    return 0;
}
