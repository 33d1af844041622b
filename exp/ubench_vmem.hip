// Microbenchmark: global_load vs buffer_load (MUBUF, offen + scalar offset) on gfx950, row gathers as the
// reservoir kernels issue them: a wave fetches its 256 B (dword per lane) or 1 KB (4 dwords per lane) slice of a
// pseudo-random row of an L2-sized table.  K independent loads per batch, then one wait; and a dependent chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int MODE, int K>
__global__ void k(const uint32_t *tab, uint32_t rows_mask, uint32_t row_bytes, uint32_t *out, int iters,
                  unsigned long long *cyc)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *base = reinterpret_cast<const char *>(tab);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(tab), 0, (int)((rows_mask + 1u) * row_bytes), 0x00020000);
    constexpr bool WIDE = (MODE & 2) != 0;
    constexpr bool BUF = (MODE & 1) != 0;
    constexpr bool CHAIN = (MODE & 4) != 0;
    const uint32_t lane_off = (uint32_t)(wave * 64 + lane) * (WIDE ? 16u : 4u);
    uint32_t j = __builtin_amdgcn_readfirstlane(blockIdx.x * 7919u + wave * 131u);
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        uint32_t v[K];
        u4 v4[K];
#pragma unroll
        for (int q = 0; q < K; ++q) {
            j = __builtin_amdgcn_readfirstlane(j * 1103515245u + 12345u);
            const uint32_t so = ((j >> 8) & rows_mask) * row_bytes;
            if (WIDE) {
                if (BUF) v4[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, (int)so, 0);
                else v4[q] = *reinterpret_cast<const u4 *>(base + so + lane_off);
            } else {
                if (BUF) v[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)lane_off, (int)so, 0);
                else v[q] = *reinterpret_cast<const uint32_t *>(base + so + lane_off);
            }
            if (CHAIN) {          // the next row index depends on the loaded data (table holds zeros)
                const uint32_t x = WIDE ? v4[q].x : v[q];
                j += __builtin_amdgcn_readfirstlane(x);
            }
        }
#pragma unroll
        for (int q = 0; q < K; ++q) acc += WIDE ? (v4[q].x + v4[q].y + v4[q].z + v4[q].w) : v[q];
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int K>
static int run(const char *name, const uint32_t *tab, uint32_t rows, uint32_t row_bytes, uint32_t *o,
               unsigned long long *c, int threads, int blocks)
{
    const int iters = 2000;
    unsigned long long hc = 0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<MODE, K>), dim3(blocks), dim3(threads), 0, 0, tab, rows - 1, row_bytes, o, iters, c);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHECK(hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost));
    printf("threads %4d blocks %4d K %2d %-28s %8.1f ticks per load (wave 0), %8.3f ms, %7.1f GB/s\n", threads, blocks, K, name,
           (double)hc / (iters * K), ms,
           (double)blocks * (threads / 64) * iters * K * 64.0 * ((MODE & 2) ? 16 : 4) / (ms * 1e6));
    return 0;
}

int main()
{
    const uint32_t rows = 1024, row_bytes = 16384;          // 16 MB: L2 + Infinity Cache resident
    uint32_t *tab, *o; unsigned long long *c;
    CHECK(hipMalloc(&tab, (size_t)rows * row_bytes)); CHECK(hipMalloc(&o, 4096 * 1024 * 4)); CHECK(hipMalloc(&c, 8));
    CHECK(hipMemset(tab, 0, (size_t)rows * row_bytes));
    for (int threads : {64, 512}) for (int blocks : {1, 256, 1024}) {
        if (run<0, 8>("global dword x8 indep", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
        if (run<1, 8>("buffer dword x8 indep", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
        if (run<2, 8>("global dwordx4 x8 indep", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
        if (run<3, 8>("buffer dwordx4 x8 indep", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
        if (run<4, 4>("global dword chain", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
        if (run<5, 4>("buffer dword chain", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
        if (run<6, 4>("global dwordx4 chain", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
        if (run<7, 4>("buffer dwordx4 chain", tab, rows, row_bytes, o, c, threads, blocks)) return 1;
    }
    return 0;
}
