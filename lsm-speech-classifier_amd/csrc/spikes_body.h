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

// One 256-thread workgroup, one clip `b`: min/max, floor, normalise, SciPy-exact resize, hysteresis latches, raster.
// `smem`: 64 bytes of reduction scratch + F * ceil(time_bins*n_thr/32) words of bit-packed raster stage.  Shared by
// spec_to_spikes_kernel (frontend.hip) and the one-launch mel front end (mel.hip), whose last workgroup of a clip
// runs it on the dB values it has just formed.
template <typename T>
__device__ __forceinline__ void spec_to_spikes_body(const SpikeArgs<T> &a, const int b, unsigned char *smem)
{
    T *scratch = reinterpret_cast<T *>(smem);                 // 8 entries
    // the clip's raster is staged bit-packed (F rows of RW words; bit p of a row = raster byte p), so the
    // kernel needs 6.7 KB of LDS at 128 filters instead of 51 KB and fits beside the workgroups of the
    // other kernels of the pipeline
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + 64);
    const int F = a.n_filters, nc = a.ncols, Tb = a.time_bins;
    const T *db = a.db + (size_t)b * F * nc;
    const int n = F * nc;

    T mx = -INFINITY, mn = INFINITY;
    int nan_seen = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
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
    const int row_bytes = Tb * a.n_thr;
    const int RW = (row_bytes + 31) >> 5;                     // words per staged row
    const double zf = (double)(nc - 1) / (double)(Tb - 1);

    for (int r = threadIdx.x; r < F; r += blockDim.x) {
        const T *row = db + (size_t)r * nc;
        bool active[MAX_THR];
#pragma unroll
        for (int q = 0; q < MAX_THR; ++q) active[q] = false;
        uint32_t word = 0u;
        int pos = 0;                                          // bit position in the row = j*n_thr + q
        for (int j = 0; j < Tb; ++j) {
            T val;
            if (flat) {
                val = (T)0;
            } else if (nc == Tb) {
                T x0 = row[j];
                x0 = x0 > fl ? x0 : fl;
                val = (x0 - lo) / den;
            } else {
                // scipy.ndimage.zoom(order=1): double coordinate and weights, w1 = 1 - w0
                const double cc = (double)j * zf;
                const double fc = floor(cc);
                const int f = (int)fc;
                const double w0 = 1.0 - (cc - fc);
                const double w1 = 1.0 - w0;
                T x0 = row[f];
                x0 = x0 > fl ? x0 : fl;
                const T n0 = (x0 - lo) / den;
                double acc = (double)n0 * w0;
                if (f + 1 <= nc - 1) {
                    T x1 = row[f + 1];
                    x1 = x1 > fl ? x1 : fl;
                    const T n1 = (x1 - lo) / den;
                    acc = acc + (double)n1 * w1;
                }
                val = (T)acc;
            }
            if (a.norm_out) a.norm_out[((size_t)b * F + r) * Tb + j] = val;
#pragma unroll
            for (int q = 0; q < MAX_THR; ++q) {
                if (q < a.n_thr) {
                    const bool rising = (val > a.on[q]) && !active[q];
                    const bool falling = (val < a.off[q]) && active[q];
                    if (rising) active[q] = true;
                    if (falling) active[q] = false;
                    word |= (active[q] ? 1u : 0u) << (pos & 31);
                    if ((pos & 31) == 31) {
                        stage[r * RW + (pos >> 5)] = word;
                        word = 0u;
                    }
                    ++pos;
                }
            }
        }
        if (pos & 31) stage[r * RW + (pos >> 5)] = word;
    }
    __syncthreads();
    if (a.raster) {
        // create_pure_redundancy: output row c reads filter row c / redundancy
        const int C = F * a.redundancy;
        uint8_t *dst = a.raster + (size_t)b * C * row_bytes;
        if ((row_bytes & 3) == 0) {
            // four raster bytes per store: bits p..p+3 (p a multiple of 4, so they share a word)
            // spread to one bit per byte by a multiply
            const int rw = row_bytes / 4;
            uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
            for (int i = threadIdx.x; i < C * rw; i += blockDim.x) {
                const int c = i / rw;
                const int p = (i - c * rw) * 4;
                const uint32_t nib = (stage[(c / a.redundancy) * RW + (p >> 5)] >> (p & 31)) & 0xFu;
                d4[i] = (nib * 0x00204081u) & 0x01010101u;
            }
        } else {
            for (int i = threadIdx.x; i < C * row_bytes; i += blockDim.x) {
                const int c = i / row_bytes;
                const int p = i - c * row_bytes;
                dst[i] = (uint8_t)((stage[(c / a.redundancy) * RW + (p >> 5)] >> (p & 31)) & 1u);
            }
        }
    }
}


}  // namespace lsm_fe
