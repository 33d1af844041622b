// spikes_body.h -- dB spectrogram of one clip -> normalised spectrogram -> hysteresis spike raster, as a device
// function (/root/reference/create_dataset.py:62-104: min-max normalise with the gammatone branch's floor, zoom to
// TIME_BINS columns, crop, convert_spectrogram_to_spikes_hysteresis, create_pure_redundancy).
#pragma once
#include "lsm_common.h"

namespace lsm_fe {

template <typename T> struct Acc;          // arithmetic type of the normalise step per input dtype
template <> struct Acc<double> { typedef double type; };
template <> struct Acc<float> { typedef float type; };

constexpr int MAX_THR = 8;

template <typename T>
struct SpikeArgs {
    const T *db;            // (B, F, ncols)
    int n_clips, n_filters, ncols, time_bins, apply_floor, n_thr, redundancy;
    T on[MAX_THR], off[MAX_THR];
    uint8_t *raster;        // (B, F*redundancy, time_bins*n_thr) or null
    T *norm_out;            // (B, F, time_bins) or null
};

template <typename T>
__device__ __forceinline__ T block_reduce(T v, bool is_max, T *scratch)
{
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(v, off);
        v = is_max ? (o > v ? o : v) : (o < v ? o : v);
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i)
        r = is_max ? (scratch[i] > r ? scratch[i] : r) : (scratch[i] < r ? scratch[i] : r);
    return r;
}

#ifndef LSM_SPK_ABLATE
#define LSM_SPK_ABLATE 0    // diagnostic builds only (WRONG results): 1 = no minimum/maximum pass, 2 = no values / comparison bits,
#endif                      // 4 = no latches, 8 = no raster bytes (profiles/r05_mel_wave_per_frame.txt)
constexpr int SPK_ROWS = 64;               // spectrogram rows (filters) a workgroup encodes at a time

// LDS bytes of spec_to_spikes_body: 128 of reduction scratch (sixteen waves) + two bit arrays (value above the on-threshold / below the
// off-threshold; the first becomes the latch state) of SPK_ROWS rows x n_thr thresholds x ceil(time_bins / 32) words
__host__ __device__ inline size_t spikes_lds_bytes(int time_bins, int n_thr)
{
    return 128 + 2 * (size_t)SPK_ROWS * (size_t)(n_thr > 0 ? n_thr : 1) * (size_t)((time_bins + 31) / 32) * 4;
}

// One workgroup of 4 to 16 waves, one clip `b`: min/max, floor, normalise, SciPy-exact resize, hysteresis latches, raster.
// `smem`: spikes_lds_bytes(time_bins, n_thr).  Shared by spec_to_spikes_kernel (frontend.hip) and the one-launch mel front
// end (mel.hip), whose last workgroup of a clip runs it on the dB values it has just formed.
// Three passes over groups of SPK_ROWS rows (round 5; before, ONE thread walked a row's time bins, its two loads per bin
// in series with the latch: 40 busy threads and 71 us per 200 clips at 40 filters):
//   1. every wave takes rows, a lane a time bin: normalised value (the reference's arithmetic, element by element), then
//      one ballot per threshold and comparison -> the bit rows "above on[q]" and "below off[q]";
//   2. one thread per (row, threshold) runs the latch over the bits: active' = active ? !below : above
//      (create_dataset.py:88-96: rising and falling are both taken from the latch before the update), 32 bins per addition;
//   3. the raster bytes, four per store, from the latch bits.
template <typename T>
__device__ __forceinline__ void spec_to_spikes_body(const SpikeArgs<T> &a, const int b, unsigned char *smem)
{
    T *scratch = reinterpret_cast<T *>(smem);                 // 16 entries: one per wave of up to 1024 threads
    const int F = a.n_filters, nc = a.ncols, Tb = a.time_bins, nq = a.n_thr;
    const int W = (Tb + 31) >> 5;                             // words of a bit row
    uint32_t *onb = reinterpret_cast<uint32_t *>(smem + 128); // [row in group][q][W]: above on[q]; then the latch state
    uint32_t *offb = onb + (size_t)SPK_ROWS * (nq > 0 ? nq : 1) * W;
    const T *db = a.db + (size_t)b * F * nc;
    const int n = F * nc;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = (int)(blockDim.x >> 6);

    T mx = -INFINITY, mn = INFINITY;
    int nan_seen = 0;
#pragma unroll 8
    for (int i = tid; i < ((LSM_SPK_ABLATE & 1) ? 0 : n); i += blockDim.x) {
        const T v = db[i];
        mx = v > mx ? v : mx;
        mn = v < mn ? v : mn;
        nan_seen |= v != v;
    }
    mx = block_reduce(mx, true, scratch);
    mn = block_reduce(mn, false, scratch);
    // ndarray.max() / .min() / np.maximum (create_dataset.py:59-63) propagate NaN, the comparisons above skip it: one NaN
    // among the dB values makes the maximum, the floor and the minimum NaN, hence every normalised value NaN and the
    // raster all zeros (tests/golden/postfilter_nonfinite.npz: the reference's own output for such inputs)
    if (__syncthreads_or(nan_seen)) mx = mn = (T)NAN;
    // create_dataset.py:60 floors at max-80 before the min is taken: min' = max(min, max-80)
    const T fl = a.apply_floor ? mx - (T)80.0 : -INFINITY;
    const T lo = (mn > fl || mn != mn) ? mn : fl;
    const T hi = mx;
    const bool flat = (hi - lo) < (T)1e-8;
    const T den = (hi - lo) + (T)1e-8;
    const int row_bytes = Tb * nq;
    const double zf = (double)(nc - 1) / (double)(Tb - 1);
    const int C = F * a.redundancy;
    uint8_t *dst = a.raster ? a.raster + (size_t)b * C * row_bytes : nullptr;

    for (int r0 = 0; r0 < F; r0 += SPK_ROWS) {
        const int rows = min(SPK_ROWS, F - r0);
        // ---- 1. values and comparison bits: a wave takes (row, 64 time bins) pieces four at a time, their loads together ----
        const int nh = (Tb + 63) >> 6, npiece = rows * nh;
        for (int p0 = wv; p0 < ((LSM_SPK_ABLATE & 2) ? 0 : npiece); p0 += 4 * nwv) {
            T x0[4], x1[4];
            double w0[4], w1[4];
            bool two[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pc = min(p0 + e * nwv, npiece - 1);     // (a piece past the end repeats the last one; nothing of it is kept)
                const int rl = pc / nh, j = (pc - rl * nh) * 64 + lane;
                const int jc = j < Tb ? j : Tb - 1;               // (lanes past the row recompute its last bin, likewise)
                const T *row = db + (size_t)(r0 + rl) * nc;
                if (nc == Tb) {
                    x0[e] = row[jc]; x1[e] = x0[e]; w0[e] = 1.0; w1[e] = 0.0; two[e] = false;
                } else {
                    // scipy.ndimage.zoom(order=1): double coordinate and weights, w1 = 1 - w0
                    const double cc = (double)jc * zf;
                    const double fc = floor(cc);
                    const int f = (int)fc;
                    w0[e] = 1.0 - (cc - fc);
                    w1[e] = 1.0 - w0[e];
                    two[e] = f + 1 <= nc - 1;
                    x0[e] = row[f]; x1[e] = row[two[e] ? f + 1 : f];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pc = p0 + e * nwv;
                if (pc >= npiece) break;                          // wave-uniform
                const int rl = pc / nh, j0 = (pc - rl * nh) * 64, j = j0 + lane;
                const bool valid = j < Tb;
                T val;
                if (flat) {
                    val = (T)0;
                } else if (nc == Tb) {
                    const T c0 = x0[e] > fl ? x0[e] : fl;
                    val = (c0 - lo) / den;
                } else {
                    const T c0 = x0[e] > fl ? x0[e] : fl;
                    const T n0 = (c0 - lo) / den;
                    double acc = (double)n0 * w0[e];
                    if (two[e]) {
                        const T c1 = x1[e] > fl ? x1[e] : fl;
                        const T n1 = (c1 - lo) / den;
                        acc = acc + (double)n1 * w1[e];
                    }
                    val = (T)acc;
                }
                if (a.norm_out && valid) a.norm_out[((size_t)b * F + r0 + rl) * Tb + j] = val;
#pragma unroll
                for (int q = 0; q < MAX_THR; ++q) {
                    if (q < nq) {
                        const unsigned long long up = __ballot(valid && val > a.on[q]);
                        const unsigned long long dn = __ballot(valid && val < a.off[q]);
                        if (lane == 0) {
                            uint32_t *o = onb + ((size_t)rl * nq + q) * W + (j0 >> 5), *d = offb + ((size_t)rl * nq + q) * W + (j0 >> 5);
                            o[0] = (uint32_t)up; d[0] = (uint32_t)dn;
                            if ((j0 >> 5) + 1 < W) { o[1] = (uint32_t)(up >> 32); d[1] = (uint32_t)(dn >> 32); }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- 2. the latches: where above and below exclude each other (off <= on), active' = above | (active & ~below) is the
        //      carry chain of an addition: generate = above, propagate = ~below, so a word of 32 time bins is one 64-bit add ----
        for (int i = tid; i < ((LSM_SPK_ABLATE & 4) ? 0 : rows * nq); i += blockDim.x) {
            uint32_t *o = onb + (size_t)i * W;
            const uint32_t *d = offb + (size_t)i * W;
            uint64_t active = 0u;
            for (int w = 0; w < W; ++w) {
                const uint32_t up = o[w], dn = d[w];
                if ((up & dn) == 0u) {
                    const uint64_t g = up, p = (uint32_t)~dn;
                    const uint64_t c = (p + g + active) ^ p ^ g;  // bit k: carry INTO bit k; bit k + 1: the latch after bin k
                    o[w] = (uint32_t)(c >> 1);
                    active = (c >> 32) & 1u;
                } else {
                    // an off-threshold ABOVE its on-threshold (negative gap): a value between them flips the latch
                    uint32_t res = 0u, act = (uint32_t)active;
                    for (int k = 0; k < 32; ++k) {
                        act = act ? (~(dn >> k) & 1u) : ((up >> k) & 1u);
                        res |= act << k;
                    }
                    o[w] = res;
                    active = act;
                }
            }
        }
        __syncthreads();
        // ---- 3. raster bytes of the group's rows (create_pure_redundancy: output row c reads filter row c / redundancy) ----
        if (dst && !(LSM_SPK_ABLATE & 8)) {
            const int c0 = r0 * a.redundancy, nrow = rows * a.redundancy;
            if ((row_bytes & 3) == 0) {
                const int rw = row_bytes / 4;
                uint32_t *d4 = reinterpret_cast<uint32_t *>(dst) + (size_t)c0 * rw;
                for (int i = tid; i < nrow * rw; i += blockDim.x) {
                    const int c = i / rw;
                    const int p = (i - c * rw) * 4;
                    const uint32_t *o = onb + (size_t)(c / a.redundancy) * nq * W;
                    uint32_t v = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = (p + e) / nq, q = (p + e) - j * nq;           // raster byte p + e = time bin j, threshold q
                        v |= ((o[q * W + (j >> 5)] >> (j & 31)) & 1u) << (8 * e);
                    }
                    d4[i] = v;
                }
            } else {
                uint8_t *d1 = dst + (size_t)c0 * row_bytes;
                for (int i = tid; i < nrow * row_bytes; i += blockDim.x) {
                    const int c = i / row_bytes;
                    const int p = i - c * row_bytes;
                    const int j = p / nq, q = p - j * nq;
                    d1[i] = (uint8_t)((onb[((size_t)(c / a.redundancy) * nq + q) * W + (j >> 5)] >> (j & 31)) & 1u);
                }
            }
        }
        __syncthreads();                                      // the bit rows are free for the next group
    }
}

}  // namespace lsm_fe
