// f64 VALU issue-rate probe for gfx950: how many cycles does a wave64 v_mul_f64 / v_add_f64 /
// v_fma_f64 occupy a SIMD, alone and with the chip saturated?  (Feeds the gammatone roofline.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void probe(double *out, double c, double d, int iters)
{
    double a[CHAINS];
#pragma unroll
    for (int q = 0; q < CHAINS; ++q) a[q] = 1.0 + 1e-9 * (threadIdx.x + q);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int q = 0; q < CHAINS; ++q) {
                if (OP == 0) a[q] = a[q] * c;
                else if (OP == 1) a[q] = a[q] + c;
                else if (OP == 2) a[q] = __builtin_fma(a[q], c, d);
                else { a[q] = a[q] * c; a[q] = a[q] + d; }        // mul then dependent add
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < CHAINS; ++q) s += a[q];
    if (s == 12345.678) out[0] = s;
}

template <int OP, int CHAINS>
static void run(const char *name, int blocks, int iters, double *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<OP, CHAINS><<<blocks, 256>>>(out, 1.0000001, 1e-9, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP, CHAINS><<<blocks, 256>>>(out, 1.0000001, 1e-9, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)iters * 8 * CHAINS * (OP == 3 ? 2 : 1);
    const double waves_per_simd = blocks * 4.0 / 1024.0;           // 256 CUs x 4 SIMDs
    const double instr_per_simd = per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd);
    printf("%-8s chains %d blocks %5d (%.2f waves/SIMD)  %.3f ms  -> %.2f cycles@2.4GHz per wave-instr per SIMD\n",
           name, CHAINS, blocks, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
    double *out; hipMalloc(&out, 64);
    const int it = 20000;
    for (int blocks : {128, 256, 512, 1024, 2048}) {
        run<0, 8>("mul", blocks, it, out);
        run<1, 8>("add", blocks, it, out);
        run<2, 8>("fma", blocks, it, out);
        run<0, 1>("mul", blocks, it, out);
        run<1, 1>("add", blocks, it, out);
        run<3, 1>("mul+add", blocks, it, out);
    }
    return 0;
}
