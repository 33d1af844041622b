// Round 5: is ds_add_f32 (LDS float atomic add, no return) the same function as v_add_f32 on gfx950?  Random operand pairs over
// the whole exponent range (denormals, zeros of both signs, infinities and NaNs counted apart), bits compared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

__global__ void k(const float *a, const float *b, float *valu, float *lds, int n) {
    __shared__ float s[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    float r = x + y;
    asm volatile("" : "+v"(r));
    valu[i] = r;
    s[threadIdx.x] = x;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)&s[threadIdx.x];
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_f32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(y) : "memory");
    lds[i] = s[threadIdx.x];
}

int main() {
    const int n = 1 << 24;
    std::mt19937_64 g(1234);
    std::vector<float> a(n), b(n);
    for (int i = 0; i < n; ++i) {
        uint32_t u = (uint32_t)g(), v = (uint32_t)g();
        const int mode = i & 7;
        if (mode == 1) v = (v & 0x807FFFFFu) | (u & 0x7F800000u);                   // same exponent: cancellation / carries
        if (mode == 2) { u &= 0x807FFFFFu; v &= 0x807FFFFFu; }                      // both denormal
        if (mode == 3) { u = (u & 0x807FFFFFu) | 0x00800000u; v = (v & 0x807FFFFFu) | 0x00800000u; v ^= (u ^ v) & 0x80000000u; v ^= 0x80000000u; }  // smallest normals, opposite signs
        if (mode == 4) { u = (u & 0x80FFFFFFu) | 0x3B000000u; v = (v & 0x80FFFFFFu) | 0x3B000000u; }   // reservoir-like magnitudes
        if (mode == 5) { const int e = (int)((u >> 23) & 0xFF) + (int)(v % 49) - 24; v = (v & 0x807FFFFFu) | ((uint32_t)(e < 1 ? 1 : e > 254 ? 254 : e) << 23); }  // exponents within 24
        std::memcpy(&a[i], &u, 4); std::memcpy(&b[i], &v, 4);
    }
    float *da, *db, *dv, *dl;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dv, dl, n);
    std::vector<float> v(n), l(n);
    hipMemcpy(v.data(), dv, n * 4, hipMemcpyDeviceToHost); hipMemcpy(l.data(), dl, n * 4, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    long diff = 0, diff_nan = 0, diff_den_in = 0, diff_den_out = 0, diff_other = 0, host_diff = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t x, y; std::memcpy(&x, &v[i], 4); std::memcpy(&y, &l[i], 4);
        volatile float h = a[i] + b[i]; float hh = h; uint32_t z; std::memcpy(&z, &hh, 4);
        if (z != x && !(std::isnan(hh) && std::isnan(v[i]))) ++host_diff;
        if (x == y) continue;
        ++diff;
        auto den = [](float f) { return f != 0.0f && std::fabs(f) < 1.17549435e-38f; };
        if (std::isnan(v[i]) || std::isnan(l[i])) ++diff_nan;
        else if (den(a[i]) || den(b[i])) ++diff_den_in;
        else if (den(v[i]) || den(l[i])) ++diff_den_out;
        else { if (diff_other < 5) printf("  other: %a + %a: valu %a lds %a\n", a[i], b[i], v[i], l[i]); ++diff_other; }
    }
    printf("ds_add_f32 vs v_add_f32 on %d pairs: %ld differ (NaN payload/sign %ld, denormal operand %ld, denormal result %ld, other %ld); "
           "v_add_f32 vs the host's IEEE add: %ld differ\n", n, diff, diff_nan, diff_den_in, diff_den_out, diff_other, host_diff);
    return 0;
}
