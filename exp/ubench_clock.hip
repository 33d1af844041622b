// Calibrate s_memtime vs s_memrealtime vs a dependent v_fma chain; also L2-hit load latency and LDS read latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k(float *out, const uint32_t *chase, unsigned long long *res, int iters, int spin)
{
    __shared__ uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (i * 17 + 1) & 1023;
    __syncthreads();
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    // optional warm spin to let clocks ramp
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, y, 1e-7f);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, y, 1e-7f);
    asm volatile("" :: "v"(x));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    // dependent LDS chase
    uint32_t p = threadIdx.x & 1023;
    for (int i = 0; i < iters; ++i) p = lds[p];
    asm volatile("" :: "v"(p));
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    // dependent global chase (L2-resident 1 MB table)
    uint32_t g = threadIdx.x;
    for (int i = 0; i < iters; ++i) g = chase[g];
    asm volatile("" :: "v"(g));
    unsigned long long t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { res[0] = t1 - t0; res[1] = r1 - r0; res[2] = t2 - t1; res[3] = t3 - t2; res[4] = r3 - r0; res[5] = t3 - t0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + p + g;
}

int main()
{
    const int n = 1 << 18;   // 1 MB table
    uint32_t *h = (uint32_t *)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = (uint32_t)(((unsigned long long)i * 40503u + 12345u) % n);
    uint32_t *d; float *o; unsigned long long *c;
    CHECK(hipMalloc(&d, n * 4)); CHECK(hipMalloc(&o, 4096 * 64 * 4)); CHECK(hipMalloc(&c, 64));
    CHECK(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice));
    for (int blocks : {1, 256, 2048}) for (int spin : {0, 2000000}) {
        unsigned long long r[6];
        const int iters = 2000;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, o, d, c, iters, spin); CHECK(hipDeviceSynchronize()); }
        CHECK(hipMemcpy(r, c, 48, hipMemcpyDeviceToHost));
        printf("blocks %4d spin %7d: fma %.2f ticks (%.2f ns) each; LDS dep read %.1f ticks; L2 dep load %.1f ticks; memtime/realtime ratio %.3f -> clock %.0f MHz if realtime=100MHz\n",
               blocks, spin, (double)r[0] / iters, (double)r[1] / iters * 10.0, (double)r[2] / iters, (double)r[3] / iters,
               (double)r[5] / (double)r[4], (double)r[5] / (double)r[4] * 100.0);
    }
    return 0;
}
