// Microbenchmarks: cost of ds_add_f32 vs ds_write/ds_read RMW, for one wave and for 8 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(const uint32_t *idx, float *out, int iters, unsigned long long *cyc)
{
    __shared__ float acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float *my = acc + wave * 256;
    uint32_t t[8];
    for (int q = 0; q < 8; ++q) t[q] = idx[q * 64 + lane] & 255;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) __hip_atomic_fetch_add(my + t[q], 1.0f + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) { float x = my[t[q]]; my[t[q]] = x + (1.0f + q); }
            if (MODE == 2) my[t[q]] = 1.0f + q + it;
            if (MODE == 3) { float x = my[t[q]]; asm volatile("" :: "v"(x)); }
            if (MODE == 4) atomicAdd(reinterpret_cast<unsigned int *>(my) + t[q], 1u);
            if (MODE == 5) atomicOr(reinterpret_cast<unsigned int *>(my) + t[q], 1u << q);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x];
}

int main()
{
    uint32_t h[512];
    // distinct targets within each instruction (a permutation of 0..63 shifted per q), like one CSC row
    for (int q = 0; q < 8; ++q) for (int l = 0; l < 64; ++l) h[q * 64 + l] = (l * 37 + q * 64) & 255;
    uint32_t *d; float *o; unsigned long long *c;
    CHECK(hipMalloc(&d, sizeof(h))); CHECK(hipMalloc(&o, 1024 * 512 * 4)); CHECK(hipMalloc(&c, 8));
    CHECK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
    const char *names[6] = {"ds_add_f32", "read+add+write", "ds_write", "ds_read", "ds_add_u32", "ds_or_b32"};
    for (int threads : {64, 512}) for (int blocks : {1, 256}) {
        for (int mode = 0; mode < 6; ++mode) {
            const int iters = 1000;
            unsigned long long hc = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost));
            printf("threads %4d blocks %3d %-16s %8.1f memtime-ticks per wave-instruction\n", threads, blocks, names[mode], (double)hc / (iters * 8));
        }
    }
    return 0;
}
