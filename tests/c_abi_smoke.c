/*
 * Plain-C consumer of include/lsm_hip.h: no Python, no torch.  Builds a small ring-lattice reservoir
 * by hand, runs lsm_reservoir_run on raw hipMalloc buffers and compares every feature with the C
 * oracle (oracle/liblsm_oracle.so, test infrastructure) bit for bit.  Built and run by
 * tests/test_gpu_c_abi.py on the GPU box:
 *   gcc tests/c_abi_smoke.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L<pkg> -llsm_hip
 *       -Loracle -llsm_oracle -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lsm_hip.h"

int orc_lif_run(int N, int C, int T, const int32_t *csr_ptr, const int32_t *csr_pre, const float *csr_w,
                const int32_t *in_ptr, const int32_t *in_chan, float w_in, const float *leak, float theta,
                int refractory, const uint8_t *raster, uint8_t *spike_matrix, float *v_trace, int n_out,
                const int32_t *out_idx, int burst_isi_max, int n_keys, const int32_t *key_ids, float *features);

#define N 192
#define K 12          /* ring neighbours (K/2 per side) */
#define C 6
#define D 4           /* targets per input channel */
#define T 80
#define B 3
#define NOUT 64
#define CHECK(x) do { int rc_ = (x); if (rc_) { printf("FAIL %s -> %d: %s\n", #x, rc_, lsm_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint32_t rng = 12345u;
static uint32_t next(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }

int main(void)
{
    if (lsm_device_count() < 1) { printf("FAIL no device: %s\n", lsm_last_error()); return 1; }
    /* ring lattice: neuron i is connected with i +- 1..K/2; the same matrix in CSR (by post) and CSC (by pre) */
    static int32_t ptr[N + 1], idx[N * K], in_tgt[C * D], in_ptr[N + 1], in_chan[C * D], out_idx[NOUT], keys[8];
    static float w_csr[N * K], w_csc[N * K], leak[N];
    static float wmat[N][N];
    for (int i = 0; i < N; ++i)
        for (int d = 1; d <= K / 2; ++d) {
            wmat[i][(i + d) % N] = 0.05f + 0.002f * (float)(next() % 100);
            wmat[i][(i + N - d) % N] = 0.05f + 0.002f * (float)(next() % 100);
        }
    int e = 0;
    for (int i = 0; i < N; ++i) {                 /* row i = post, columns = pre ascending */
        ptr[i] = e;
        for (int j = 0; j < N; ++j)
            if (wmat[i][j] != 0.0f) { idx[e] = j; w_csr[e] = wmat[i][j]; ++e; }
    }
    ptr[N] = e;
    /* the lattice is symmetric in structure, so CSC by pre has the same ptr/idx; weights transposed */
    e = 0;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i)
            if (wmat[i][j] != 0.0f) w_csc[e++] = wmat[i][j];
    for (int i = 0; i < N; ++i) leak[i] = 0.01f;
    for (int c = 0; c < C; ++c)
        for (int d = 0; d < D; ++d) in_tgt[c * D + d] = (c * 31 + d * 47) % N;      /* distinct per channel */
    for (int c = 0; c < C; ++c)                    /* sort each channel's targets ascending */
        for (int a = 0; a < D; ++a)
            for (int b = a + 1; b < D; ++b)
                if (in_tgt[c * D + b] < in_tgt[c * D + a]) { int t = in_tgt[c * D + a]; in_tgt[c * D + a] = in_tgt[c * D + b]; in_tgt[c * D + b] = t; }
    e = 0;
    for (int i = 0; i < N; ++i) {                  /* input map by post neuron, channels ascending */
        in_ptr[i] = e;
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                if (in_tgt[c * D + d] == i) in_chan[e++] = c;
    }
    in_ptr[N] = e;
    for (int o = 0; o < NOUT; ++o) out_idx[o] = o * 3;
    for (int k = 0; k < 8; ++k) keys[k] = k;
    static uint8_t raster[B][C][T];
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int t = 0; t < T; ++t) raster[b][c][t] = (next() % 100) < (b == 1 ? 0u : 55u);   /* clip 1 silent */

    const float theta = 1.0f, w_in = 0.4f;
    lsm_reservoir *h = NULL;
    CHECK(lsm_reservoir_create(&h, N, C, ptr, idx, w_csc, leak, in_tgt, D, w_in, out_idx, NOUT, theta, 2, 5));
    uint8_t *d_r = NULL; float *d_f = NULL;
    const size_t nf = 8 * NOUT;
    HIP(hipMalloc((void **)&d_r, sizeof(raster)));
    HIP(hipMalloc((void **)&d_f, B * nf * sizeof(float)));
    HIP(hipMemcpy(d_r, raster, sizeof(raster), hipMemcpyHostToDevice));
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    static float got[B][8 * NOUT], ref[8 * NOUT];
    long spikes = 0;
    const int layouts[3] = {0, 1, 2};
    for (int li = 0; li < 3; ++li) {
        CHECK(lsm_reservoir_run(h, d_r, B, T, keys, 8, d_f, NULL, NULL, NULL, layouts[li], st));
        HIP(hipStreamSynchronize(st));
        HIP(hipMemcpy(got, d_f, sizeof(got), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            if (orc_lif_run(N, C, T, ptr, idx, w_csr, in_ptr, in_chan, w_in, leak, theta, 2, &raster[b][0][0], NULL,
                            NULL, NOUT, out_idx, 5, 8, keys, ref)) { printf("FAIL oracle\n"); return 1; }
            if (memcmp(ref, got[b], sizeof(ref)) != 0) { printf("FAIL features differ (clip %d, layout %d)\n", b, layouts[li]); return 1; }
            for (int o = 0; o < NOUT; ++o) spikes += (long)ref[o];
        }
    }
    /* error path: bad key id must be refused with a message, then the handle must still work */
    int32_t bad = 11;
    if (lsm_reservoir_run(h, d_r, B, T, &bad, 1, d_f, NULL, NULL, NULL, 0, st) == 0 || !lsm_last_error()[0]) { printf("FAIL bad key accepted\n"); return 1; }
    CHECK(lsm_reservoir_run(h, d_r, B, T, keys, 8, d_f, NULL, NULL, NULL, 0, st));
    HIP(hipStreamSynchronize(st));
    /* the same batch through the entry point that starts long clips first (caller-owned workspace): same rows */
    {
        void *ws = NULL;
        const long need = lsm_reservoir_order_workspace(B);
        if (need != 8L * B) { printf("FAIL workspace size %ld\n", need); return 1; }
        HIP(hipMalloc(&ws, (size_t)need));
        HIP(hipMemset(d_f, 0xff, B * nf * sizeof(float)));
        CHECK(lsm_reservoir_run_ordered(h, d_r, B, T, keys, 8, d_f, NULL, NULL, NULL, 0, ws, need, st));
        HIP(hipStreamSynchronize(st));
        static float again[B][8 * NOUT];
        HIP(hipMemcpy(again, d_f, sizeof(again), hipMemcpyDeviceToHost));
        if (memcmp(again, got, sizeof(got)) != 0) { printf("FAIL ordered launch differs\n"); return 1; }
        if (lsm_reservoir_run_ordered(h, d_r, B, T, keys, 8, d_f, NULL, NULL, NULL, 0, ws, need - 1, st) == 0) { printf("FAIL short workspace accepted\n"); return 1; }
        HIP(hipFree(ws));
    }
    CHECK(lsm_reservoir_destroy(h));
    if (spikes == 0) { printf("FAIL reservoir never fired\n"); return 1; }
    printf("C ABI OK: version %d, %ld output spikes, features bit-identical to the oracle for 3 layouts\n", lsm_version(), spikes / 3);
    return 0;
}
