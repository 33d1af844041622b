// Front end for gfx950: gammatone filterbank -> dB -> min-max normalise -> linear resize ->
// hysteresis spike encoder (SPEC.md §1).
//
// Replaces /root/reference/create_dataset.py:49-60 (gammatone branch of audio_to_spectrogram,
// arithmetic in gammatone==1.0.3 / scipy.signal.lfilter), :62-78 (normalise, zoom, crop),
// :81-98 (convert_spectrogram_to_spikes_hysteresis) and :101-104 (create_pure_redundancy),
// batched over clips.  Every float operation is written in the reference's order and compiled
// with -ffp-contract=off so that results equal the CPU oracle bit for bit (log10 excepted: no
// two libms agree on its last bit).
#include "lsm_common.h"
#include "spikes_body.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------
// Gammatone spectrogram.  One lane = one channel, one wave = 64 channels of ONE clip, so the audio
// sample is wave-uniform and arrives through scalar loads (8 samples per fetch, next chunk
// requested before the current one is consumed).  Per sample: four cascaded second-order
// sections in float64 evaluated like scipy.signal.lfilter (direct form II transposed), / gain,
// squared, then per column the sequential ascending sum of nwin samples, / nwin, sqrt.  Up to
// NWIN_MAX overlapping windows are live at once (nwin <= NWIN_MAX * hop); they sit in named
// registers that shift by one at every hop boundary.  Serial in time, parallel over clips x
// channels: bound by float64 VALU issue (~36 operations per sample), not by memory.
//
// Exactness notes.  (1) A2 == 0 for every channel of this filter design, so x*b2 is a signed zero
// and z1 = x*b2 - y*a2 equals -(y*a2) up to the sign of an exact zero, which can never reach a
// non-zero value downstream; B2ZERO drops that product.  (2) y/gain is evaluated as
// q = y*r, q' = fma(fma(-q, gain, y), r, q) with r = RN(1/gain): Markstein's sequence returns the
// correctly rounded quotient (= the IEEE division the reference performs) unless gain's significand
// is all ones, which the host checks before choosing this path.
// ---------------------------------------------------------------------------------------------
constexpr int NWIN_MAX = 4;
constexpr int GT_MAX_WPB = 8;           // waves per workgroup, at most

// NW = ceil(nwin / hop) windows are live at any sample.  In every hop block exactly one window
// completes: the one of age NW-1, after sample number pos = nwin - (NW-1)*hop of the block.  So a
// block is: [0, pos) with NW windows accumulating, finalise, [pos, hop) with NW-1, rotate.
template <int NW, bool B2ZERO, bool FASTDIV>
__global__ __launch_bounds__(GT_MAX_WPB * 64) void gammatone_kernel(
    const float *__restrict__ audio, int n_clips, int n_samples,
    const double *__restrict__ coefs, int n_filters, int nwin, int hop, int ncols,
    double *__restrict__ spec_out, double *__restrict__ db_out)
{
    // one wave = 64 channels of one clip; the waves of a workgroup are consecutive (clip, channel
    // group) pairs.  A workgroup of 4 waves occupies each SIMD of its CU once: single-wave workgroups
    // are packed by the dispatcher up to the occupancy limit of a CU before the next CU is used
    // (2048 of them ran on half the chip, 3.7 ms instead of 2.6 ms at 1024 clips).
    const int groups = (n_filters + 63) >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(
        (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int b = wid / groups;
    if (b >= n_clips) return;                            // wave-uniform; the kernel has no barrier
    const int chl = (wid - b * groups) * 64 + (int)(threadIdx.x & 63);
    const bool live = chl < n_filters;
    const int ch = live ? chl : n_filters - 1;
    const double *k = coefs + (size_t)ch * 10;
    const double a0 = k[6];
    const double b0 = k[0] / a0, b2 = k[5] / a0;
    const double b11 = k[1] / a0, b12 = k[2] / a0, b13 = k[3] / a0, b14 = k[4] / a0;
    const double a1 = k[7] / a0, a2 = k[8] / a0, gain = k[9];
    const double rgain = 1.0 / gain;
    const float *__restrict__ x = audio + (size_t)b * n_samples;      // wave-uniform

    double z01 = 0, z11 = 0, z02 = 0, z12 = 0, z03 = 0, z13 = 0, z04 = 0, z14 = 0;
    double win[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) win[q] = 0.0;

    auto filt = [&](float xf) -> double {
        const double x0 = (double)xf;
        const double y1 = z01 + b0 * x0;
        z01 = (z11 + x0 * b11) - y1 * a1;
        z11 = B2ZERO ? -(y1 * a2) : x0 * b2 - y1 * a2;
        const double y2 = z02 + b0 * y1;
        z02 = (z12 + y1 * b12) - y2 * a1;
        z12 = B2ZERO ? -(y2 * a2) : y1 * b2 - y2 * a2;
        const double y3 = z03 + b0 * y2;
        z03 = (z13 + y2 * b13) - y3 * a1;
        z13 = B2ZERO ? -(y3 * a2) : y2 * b2 - y3 * a2;
        const double y4 = z04 + b0 * y3;
        z04 = (z14 + y3 * b14) - y4 * a1;
        z14 = B2ZERO ? -(y4 * a2) : y3 * b2 - y4 * a2;
        double o;
        if (FASTDIV) {
            const double q0 = y4 * rgain;
            o = __builtin_fma(__builtin_fma(-q0, gain, y4), rgain, q0);
        } else {
            o = y4 / gain;
        }
        return o * o;
    };
    // samples [n, n_to) with the NACT youngest windows accumulating; the scalar load of the next
    // 8 samples is issued before the current 8 are consumed (clamped inside the clip: the extra chunk
    // at the end of a run is loaded but never used)
#define LSM_RUN(n_to, NACT)                                                         \
    {                                                                               \
        if (n + 8 <= (n_to)) {                                                      \
            float xs[8], nx[8];                                                     \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) xs[u] = x[n + u];         \
            for (; n + 8 <= (n_to); n += 8) {                                       \
                const float *pn = x + min(n + 8, n_samples - 8);                    \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) nx[u] = pn[u];        \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) {                     \
                    const double e = filt(xs[u]);                                   \
                    _Pragma("unroll") for (int q = 0; q < (NACT); ++q) win[q] += e; \
                }                                                                   \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) xs[u] = nx[u];        \
            }                                                                       \
        }                                                                           \
        for (; n < (n_to); ++n) {                                                   \
            const double e = filt(x[n]);                                            \
            _Pragma("unroll") for (int q = 0; q < (NACT); ++q) win[q] += e;         \
        }                                                                           \
    }

    const int pos = nwin - (NW - 1) * hop;               // in (0, hop]
    const int n_end = (ncols - 1) * hop + nwin;          // samples past this feed no column
    const double dn = (double)nwin;
    const size_t obase = ((size_t)b * n_filters + ch) * ncols;
    int n = 0;
    for (int h = 0; n < n_end; ++h) {
        const int base = h * hop;
        const int mid = min(base + pos, n_end);
        LSM_RUN(mid, NW)
        const int c = h - (NW - 1);                      // the column that just completed
        if (n == base + pos && c >= 0 && c < ncols) {
            const double y = sqrt(win[NW - 1] / dn);
            if (live) {
                if (spec_out) spec_out[obase + c] = y;
                if (db_out) db_out[obase + c] = 20 * log10(y + 1e-9);
            }
        }
        const int stop = min(base + hop, n_end);
        LSM_RUN(stop, NW - 1)
#pragma unroll
        for (int q = NW - 1; q > 0; --q) win[q] = win[q - 1];
        win[0] = 0.0;
    }
#undef LSM_RUN
}

// ---------------------------------------------------------------------------------------------
// Per clip: floor (optional) -> min/max -> normalise -> linear resize to time_bins -> hysteresis
// encoder -> uint8 raster (n_filters*redundancy, time_bins*n_thr), staged in LDS and written out
// with coalesced 4-byte stores.  One workgroup of 256 threads per clip; a thread owns a channel
// for the serial part (time_bins x n_thr latch updates).
// ---------------------------------------------------------------------------------------------
using lsm_fe::MAX_THR;
using lsm_fe::SpikeArgs;

#ifndef LSM_SPK_THREADS
#define LSM_SPK_THREADS 256     // threads of a spec_to_spikes_kernel workgroup (one clip).  Alone on the GPU sixteen waves are faster than
                                // four (21.5 against 32.7 us per 200 clips: with one wave per SIMD every dependent instruction waits out
                                // its latency), inside the overlapped pipeline the small workgroup is (cfg1: 1.26-1.29 M clips/s with 256
                                // threads, 1.22-1.26 M with 512, 1.13-1.15 M with 1024, which must find a whole CU's wave slots free;
                                // profiles/r05_mel_wave_per_frame.txt)
#endif
template <typename T>
__global__ __launch_bounds__(LSM_SPK_THREADS) void spec_to_spikes_kernel(const SpikeArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lsm_fe::spec_to_spikes_body<T>(a, (int)blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// Fused front end: filterbank -> dB -> floor/min-max normalise -> linear resize -> hysteresis encoder ->
// uint8 raster in ONE launch (create_dataset.py:49-104 per clip).  The filter loop is the one of
// gammatone_kernel; what differs:
//   * a wave carries NCH channel groups of its clip (NCH = 2: 128 channels, two independent float64
//     dependency chains per lane that share the scalar sample loads and the float32 -> float64
//     conversion), so at 128 filters ONE WAVE IS ONE CLIP and the clip's max/min need no other wave;
//   * the clip's dB columns go to a private scratch, column-major per wave (64 consecutive doubles per
//     store); the lane that wrote a value is the only one that reads it back, after the last window, when
//     the clip's maximum is known.  A row's 98 float64 values do not fit the registers next to the filter
//     state of two chains and 100 KB per clip do not fit the LDS four times per CU;
//   * max/min: running per lane, xor-shuffles per wave, one LDS exchange when a clip spans several waves
//     (the waves of a clip always share a workgroup);
//   * normalise / SciPy-exact resize / 4-threshold latch: one lane per channel, arithmetic and comparison
//     order of spec_to_spikes_kernel<double>; the raster rows of a wave are staged bit-packed in LDS and
//     leave as coalesced 4-byte stores.
// Waves beyond the batch (last workgroup) skip the filter loop but keep the barriers.
// ---------------------------------------------------------------------------------------------
struct FusedArgs {
    const float *audio;
    const double *coefs;
    double *ws;                 // (n_clips, ncols, groups*NCH*64) float64 scratch
    uint8_t *raster;            // (n_clips, n_filters*redundancy, time_bins*n_thr)
    int n_clips, n_samples, n_filters, nwin, hop, ncols;
    int time_bins, n_thr, redundancy, groups;
    int skip_epilogue;          // diagnostic builds (LSM_EXPERIMENT_HOOKS) only: time the filter loop alone
    int prio;                   // wave priority of the filter loop (s_setprio 0..3)
    double on[MAX_THR], off[MAX_THR];
};

// SHB0: every channel has the same first-section gain b0 = A0/B0 (the host verified it, coef_flags bit 2: this filter
// design has A0 = 1/fs and B0 = 1 everywhere), so the product b0*x of a sample is formed once per lane and feeds both
// chains: 70 instead of 71 instructions per sample and lane.
template <int NW, bool FAST, int NCH, bool SHB0 = false>
__global__ __launch_bounds__(GT_MAX_WPB * 64) void gammatone_spikes_kernel(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *xch = reinterpret_cast<double *>(smem);                       // 2 * GT_MAX_WPB doubles
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(smem + 2 * GT_MAX_WPB * sizeof(double));
    const int lane = (int)(threadIdx.x & 63);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int groups = a.groups;                                          // waves per clip
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb)) + wave;
    const int b_raw = wid / groups;
    const bool valid = b_raw < a.n_clips;
    const int b = valid ? b_raw : a.n_clips - 1;
    const int g = wid - b_raw * groups;
    const int F = a.n_filters, nwin = a.nwin, hop = a.hop, ncols = a.ncols, n_samples = a.n_samples;
    const int NG = groups * NCH;                                          // 64-channel groups per clip (padded)

    bool live[NCH];
    double b0[NCH], b11[NCH], b12[NCH], b13[NCH], b14[NCH], b2[NCH], a1[NCH], a2[NCH], gain[NCH], rgain[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int chl = (g * NCH + q) * 64 + lane;
        live[q] = chl < F;
        const double *k = a.coefs + (size_t)(live[q] ? chl : F - 1) * 10;
        const double a0 = k[6];
        b0[q] = k[0] / a0; b2[q] = k[5] / a0;
        b11[q] = k[1] / a0; b12[q] = k[2] / a0; b13[q] = k[3] / a0; b14[q] = k[4] / a0;
        a1[q] = k[7] / a0; a2[q] = k[8] / a0; gain[q] = k[9];
        rgain[q] = 1.0 / gain[q];
    }
    const float *__restrict__ x = a.audio + (size_t)b * n_samples;        // wave-uniform

    double z01[NCH], z11[NCH], z02[NCH], z12[NCH], z03[NCH], z13[NCH], z04[NCH], z14[NCH];
    double win[NCH][NW];
    double mx[NCH], mn[NCH];
    bool nan_seen = false;                              // a NaN among my dB values (the comparisons below skip it)
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        z01[q] = z11[q] = z02[q] = z12[q] = z03[q] = z13[q] = z04[q] = z14[q] = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) win[q][w] = 0.0;
        mx[q] = -INFINITY; mn[q] = INFINITY;
    }

    auto filt = [&](int q, double x0, double bx) __attribute__((always_inline)) -> double {
        const double y1 = z01[q] + (SHB0 ? bx : b0[q] * x0);
        z01[q] = (z11[q] + x0 * b11[q]) - y1 * a1[q];
        z11[q] = FAST ? -(y1 * a2[q]) : x0 * b2[q] - y1 * a2[q];
        const double y2 = z02[q] + b0[q] * y1;
        z02[q] = (z12[q] + y1 * b12[q]) - y2 * a1[q];
        z12[q] = FAST ? -(y2 * a2[q]) : y1 * b2[q] - y2 * a2[q];
        const double y3 = z03[q] + b0[q] * y2;
        z03[q] = (z13[q] + y2 * b13[q]) - y3 * a1[q];
        z13[q] = FAST ? -(y3 * a2[q]) : y2 * b2[q] - y3 * a2[q];
        const double y4 = z04[q] + b0[q] * y3;
        z04[q] = (z14[q] + y3 * b14[q]) - y4 * a1[q];
        z14[q] = FAST ? -(y4 * a2[q]) : y3 * b2[q] - y4 * a2[q];
        double o;
        if (FAST) {
            const double q0 = y4 * rgain[q];
            o = __builtin_fma(__builtin_fma(-q0, gain[q], y4), rgain[q], q0);
        } else {
            o = y4 / gain[q];
        }
        return o * o;
    };
#define LSM_RUNF(n_to, NACT)                                                        \
    {                                                                               \
        if (n + 8 <= (n_to)) {                                                      \
            float xs[8], nx[8];                                                     \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) xs[u] = x[n + u];         \
            for (; n + 8 <= (n_to); n += 8) {                                       \
                const float *pn = x + min(n + 8, n_samples - 8);                    \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) nx[u] = pn[u];        \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) {                     \
                    const double x0 = (double)xs[u];                                \
                    const double bx = b0[0] * x0;                                   \
                    _Pragma("unroll") for (int q = 0; q < NCH; ++q) {               \
                        const double e = filt(q, x0, bx);                           \
                        _Pragma("unroll") for (int w = 0; w < (NACT); ++w) win[q][w] += e; \
                    }                                                               \
                }                                                                   \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) xs[u] = nx[u];        \
            }                                                                       \
        }                                                                           \
        for (; n < (n_to); ++n) {                                                   \
            const double x0 = (double)x[n];                                         \
            const double bx = b0[0] * x0;                                           \
            _Pragma("unroll") for (int q = 0; q < NCH; ++q) {                       \
                const double e = filt(q, x0, bx);                                   \
                _Pragma("unroll") for (int w = 0; w < (NACT); ++w) win[q][w] += e;  \
            }                                                                       \
        }                                                                           \
    }

    // scratch of this wave: column c, chain q at wsw[(c * NG + q) * 64]
    double *wsw = a.ws + ((size_t)b * ncols * NG + (size_t)g * NCH) * 64 + lane;
    // The filter loop is the throughput-critical stream of the pipeline: a wave alone on its SIMD issues a float64
    // operation every 5.0 cycles, which leaves the reservoir waves sharing the SIMD one issue slot in five -- about
    // what they need.  At the default priority the reservoir kernel's raised-priority phases (lif_dense.h) take
    // slots from it instead.
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    if (valid) {
        const int pos = nwin - (NW - 1) * hop;
        const int n_end = (ncols - 1) * hop + nwin;
        const double dn = (double)nwin;
        int n = 0;
        for (int h = 0; n < n_end; ++h) {
            const int base = h * hop;
            const int mid = min(base + pos, n_end);
            LSM_RUNF(mid, NW)
            const int c = h - (NW - 1);
            if (n == base + pos && c >= 0 && c < ncols) {
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const double y = sqrt(win[q][NW - 1] / dn);
                    const double v = 20 * log10(y + 1e-9);
                    wsw[((size_t)c * NG + q) * 64] = v;
                    if (live[q]) {
                        mx[q] = v > mx[q] ? v : mx[q];
                        mn[q] = v < mn[q] ? v : mn[q];
                        nan_seen |= v != v;
                    }
                }
            }
            const int stop = min(base + hop, n_end);
            LSM_RUNF(stop, NW - 1)
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
#pragma unroll
                for (int w = NW - 1; w > 0; --w) win[q][w] = win[q][w - 1];
                win[q][0] = 0.0;
            }
        }
    }
#undef LSM_RUNF

    __builtin_amdgcn_s_setprio(0);
#if LSM_EXPERIMENT_HOOKS
    if (a.skip_epilogue) return;
#endif
    // ---- the clip's max / min -----------------------------------------------------------------
    double hi = mx[0], lo = mn[0];
#pragma unroll
    for (int q = 1; q < NCH; ++q) {
        hi = mx[q] > hi ? mx[q] : hi;
        lo = mn[q] < lo ? mn[q] : lo;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double oh = __shfl_xor(hi, off), ol = __shfl_xor(lo, off);
        hi = oh > hi ? oh : hi;
        lo = ol < lo ? ol : lo;
    }
    // spec_db.max() / .min() / np.maximum (create_dataset.py:59-63) propagate NaN: one NaN among the clip's dB values makes
    // max, floor and min NaN -- every normalised value NaN, the raster all zeros (tests/golden/postfilter_nonfinite.npz)
    if (__builtin_amdgcn_ballot_w64(nan_seen) != 0ull) hi = lo = (double)NAN;
    if (groups > 1) {                                   // uniform over the launch
        if (lane == 0) { xch[2 * wave] = hi; xch[2 * wave + 1] = lo; }
        __syncthreads();
        const int w0 = wave - g;                        // first wave of this clip in the workgroup
        hi = xch[2 * w0]; lo = xch[2 * w0 + 1];
        bool nan_any = hi != hi;
        for (int i = 1; i < groups; ++i) {
            const double oh = xch[2 * (w0 + i)], ol = xch[2 * (w0 + i) + 1];
            hi = oh > hi ? oh : hi;
            lo = ol < lo ? ol : lo;
            nan_any = nan_any || oh != oh;
        }
        if (nan_any) hi = lo = (double)NAN;
    }
    // create_dataset.py:60 floors at max-80 before the min is taken: min' = max(min, max-80)
    const double fl = hi - 80.0;
    lo = lo > fl ? lo : fl;                             // (NaN: fl)
    const bool flat = (hi - lo) < 1e-8;
    const double den = (hi - lo) + 1e-8;
    const int Tb = a.time_bins, n_thr = a.n_thr;
    const int row_bits = Tb * n_thr;
    const double zf = (double)(ncols - 1) / (double)(Tb - 1);
    // (x - lo) / den for 2 x 100 values per channel: den is wave-uniform, so its correctly rounded reciprocal is
    // computed once and every quotient is q = x*r, q' = fma(fma(-q, den, x), r, q) -- Markstein's sequence, the
    // correctly rounded quotient (= the IEEE division spec_to_spikes_kernel performs) unless den's significand is
    // all ones or den leaves the normal range (broken audio: inf/NaN); those clips take the true division.  The
    // numerators need no check of their own: every x is clamped into [lo, hi] (NaN and -inf become the floor), so
    // 0 <= x - lo <= hi - lo < den, a difference of two dB values: zero or far above the denormal range.
    const double rden = 1.0 / den;
    const unsigned long long den_bits = (unsigned long long)__double_as_longlong(den);
    const bool fastdiv = den > 1e-200 && den < 1e200 &&
                         (den_bits & 0x000FFFFFFFFFFFFFull) != 0x000FFFFFFFFFFFFFull;
    // Per group of JPW bins (no dependency between the bins of a group, the loads of the whole group in flight together):
    // per bin j and threshold t the two comparisons the latch needs, S = val > on[t] in bit t and R = val < off[t] in bit
    // FBH + t of the bin's field; then -- round 4: in the same iteration, from registers -- the set/reset latches of all
    // thresholds at once, A = (S & ~A) | (A & ~R), and the raster bits appended to a 64-bit accumulator that is stored to
    // the lane's LDS row a word at a time.  The first version staged the FIELDS of all bins in LDS (pass A) and ran the
    // latch as a second pass over them (pass B): 100 bytes of stage per channel instead of 50, i.e. 51 KB instead of
    // 27 KB per 4-wave workgroup -- which mattered once a ring-kernel clip pair (2 x 64 KB) had to fit beside it on a CU.
    // The only state carried from group to group is the latch (4-8 bits per chain) and the bit accumulator.
    const int FBH = n_thr <= 4 ? 4 : 8;                 // bits per half field
    const int JPW = 16 / FBH;                           // bins per 32-bit word (4 or 2)
    const int NG4 = (Tb + JPW - 1) / JPW;               // groups of JPW bins
    const int RW = (row_bits + 31) >> 5;                // words per staged raster row
    uint32_t *stage = stage_all + (size_t)wave * (NCH * 64) * RW;
    const size_t cs = (size_t)NG * 64;                  // doubles between two columns of the scratch
    // straight-line variants (fast division or not, <= 4 thresholds or not, resize or not): one uniform choice
    // per wave instead of uniform branches inside the loop, which would keep the compiler from batching the loads
    auto pass_a = [&](auto fast_c, auto nt_c, auto zoom_c) __attribute__((always_inline)) {
        constexpr bool FD = decltype(fast_c)::value;
        constexpr int NT = decltype(nt_c)::value;       // thresholds compared (tables are padded with +inf / -inf)
        constexpr bool ZOOM = decltype(zoom_c)::value;
        constexpr int H = NT <= 4 ? 4 : 8, G = 16 / H;  // = FBH, JPW
        auto norm1 = [&](double v) __attribute__((always_inline)) -> double {
            v = v > fl ? v : fl;
            const double d = v - lo;
            if (FD) {
                const double q0 = d * rden;
                return __builtin_fma(__builtin_fma(-q0, den, d), rden, q0);
            }
            return d / den;
        };
#ifndef LSM_GTF_PASS_A_UNROLL
#define LSM_GTF_PASS_A_UNROLL 1
#endif
        uint32_t act[NCH];
        unsigned long long acc[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) { act[q] = 0u; acc[q] = 0ull; }
        int fill = 0, wout = 0;                         // the same for every chain of the lane
        const uint32_t tmask_a = (1u << n_thr) - 1u;
#pragma unroll LSM_GTF_PASS_A_UNROLL
        for (int jw = 0; jw < NG4; ++jw) {
            double x0[G][NCH], x1[G][NCH], w0[G], w1[G];
            bool two[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int j = min(jw * G + u, Tb - 1);
                // scipy.ndimage.zoom(order=1): double coordinate and weights, w1 = 1 - w0
                const double cc = (double)j * zf;
                const double fc = floor(cc);
                const int f = ZOOM ? (int)fc : j;
                w0[u] = 1.0 - (cc - fc);
                w1[u] = 1.0 - w0[u];
                two[u] = f + 1 <= ncols - 1;
                const int f1 = two[u] ? f + 1 : f;
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const double *col = wsw + (size_t)q * 64;
                    x0[u][q] = col[(size_t)f * cs];
                    if (ZOOM) x1[u][q] = col[(size_t)f1 * cs];
                }
            }
            uint32_t fw[NCH];
#pragma unroll
            for (int q = 0; q < NCH; ++q) fw[q] = 0u;
#pragma unroll
            for (int u = 0; u < G; ++u) {
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const double n0 = norm1(x0[u][q]);
                    double val = n0;
                    if (ZOOM) {
                        const double n1 = norm1(x1[u][q]);
                        const double acc = n0 * w0[u];
                        // the last column has no right neighbour; its weight w1 is 0 there and, on the fast path,
                        // n1 is finite and acc >= +0, so acc + n1*0 == acc bit for bit: no branch needed
                        val = (FD || two[u]) ? acc + n1 * w1[u] : acc;
                    }
                    val = flat ? 0.0 : val;
                    uint32_t field = 0u;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        field |= (val > a.on[t] ? 1u : 0u) << t;
                        field |= (val < a.off[t] ? 1u : 0u) << (H + t);
                    }
                    fw[q] |= field << (u * 2 * H);
                }
            }
            // the latches of this group's bins, in bin order (bins past the last one repeat it and are skipped)
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (jw * G + u < Tb) {                  // wave-uniform
#pragma unroll
                    for (int q = 0; q < NCH; ++q) {
                        const uint32_t field = fw[q] >> (u * 2 * H);
                        const uint32_t S = field & tmask_a, R = (field >> H) & tmask_a;
                        act[q] = (S & ~act[q]) | (act[q] & ~R);     // create_dataset.py:90-95 (both from the old latch)
                        acc[q] |= (unsigned long long)act[q] << fill;
                    }
                    fill += n_thr;
                    if (fill >= 32) {
#pragma unroll
                        for (int q = 0; q < NCH; ++q) {
                            stage[(q * 64 + lane) * RW + wout] = (uint32_t)acc[q];
                            acc[q] >>= 32;
                        }
                        ++wout;
                        fill -= 32;
                    }
                }
            }
        }
        if (fill > 0) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) stage[(q * 64 + lane) * RW + wout] = (uint32_t)acc[q];
        }
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        using N4 = std::integral_constant<int, 4>;
        using N8 = std::integral_constant<int, MAX_THR>;
        const bool zoom = ncols != Tb;
        if (n_thr <= 4) {
            if (fastdiv) { if (zoom) pass_a(T{}, N4{}, T{}); else pass_a(T{}, N4{}, F{}); }
            else         { if (zoom) pass_a(F{}, N4{}, T{}); else pass_a(F{}, N4{}, F{}); }
        } else {
            if (fastdiv) { if (zoom) pass_a(T{}, N8{}, T{}); else pass_a(T{}, N8{}, F{}); }
            else         { if (zoom) pass_a(F{}, N8{}, T{}); else pass_a(F{}, N8{}, F{}); }
        }
    }
    __syncthreads();                                    // the wave's own staged rows (and a uniform barrier count)
    if (valid) {
        // create_pure_redundancy: output row c reads filter row c / redundancy
        const int R = a.redundancy;
        const int fbase = g * NCH * 64;
        const int rows_out = min(NCH * 64, F - fbase) * R;
        const int row_bytes = row_bits;
        uint8_t *dst = a.raster + ((size_t)b * F + fbase) * R * row_bytes;
        if ((row_bytes & 3) == 0) {
            const int rw = row_bytes >> 2;
            uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
            for (int i = lane; i < rows_out * rw; i += 64) {
                const int c = i / rw;
                const int p = (i - c * rw) * 4;
                const uint32_t nib = (stage[(c / R) * RW + (p >> 5)] >> (p & 31)) & 0xFu;
                d4[i] = (nib * 0x00204081u) & 0x01010101u;
            }
        } else {
            for (int i = lane; i < rows_out * row_bytes; i += 64) {
                const int c = i / row_bytes;
                const int p = i - c * row_bytes;
                dst[i] = (uint8_t)((stage[(c / R) * RW + (p >> 5)] >> (p & 31)) & 1u);
            }
        }
    }
}

// Stand-alone encoder on an already normalised spectrogram (B, F, n_bins): one lane per channel.
template <typename T>
__global__ __launch_bounds__(64) void encode_kernel(const T *__restrict__ spec, int n_rows,
                                                    int n_bins, int n_thr, SpikeArgs<T> thr,
                                                    uint8_t *__restrict__ out)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rows) return;
    const T *row = spec + (size_t)r * n_bins;
    uint8_t *o = out + (size_t)r * n_bins * n_thr;
    bool active[MAX_THR];
#pragma unroll
    for (int q = 0; q < MAX_THR; ++q) active[q] = false;
    for (int j = 0; j < n_bins; ++j) {
        const T val = row[j];
#pragma unroll
        for (int q = 0; q < MAX_THR; ++q) {
            if (q < n_thr) {
                const bool rising = (val > thr.on[q]) && !active[q];
                const bool falling = (val < thr.off[q]) && active[q];
                if (rising) active[q] = true;
                if (falling) active[q] = false;
                o[j * n_thr + q] = active[q] ? 1 : 0;
            }
        }
    }
}

template <typename T>
int launch_spec_to_spikes(const T *db, int n_clips, int n_filters, int ncols, int time_bins,
                          int apply_floor, const T *thr_on, const T *thr_off, int n_thr,
                          int redundancy, uint8_t *raster, T *norm_out, void *stream)
{
    LSM_REQUIRE(n_clips >= 0 && n_filters >= 1 && ncols >= 2 && time_bins >= 2, "bad shape");
    LSM_REQUIRE(db != nullptr || n_clips == 0, "spec_to_spikes: null input");
    LSM_REQUIRE(n_thr >= 0 && n_thr <= MAX_THR, "n_thr=%d outside [0, %d]", n_thr, MAX_THR);
    LSM_REQUIRE(redundancy >= 1, "redundancy must be >= 1");
    LSM_REQUIRE(n_thr == 0 || (thr_on && thr_off), "null threshold table");
    LSM_REQUIRE(raster == nullptr || n_thr >= 1, "a raster needs at least one threshold");
    if (n_clips == 0) return LSM_OK;
    SpikeArgs<T> a;
    a.db = db; a.n_clips = n_clips; a.n_filters = n_filters; a.ncols = ncols;
    a.time_bins = time_bins; a.apply_floor = apply_floor; a.n_thr = n_thr;
    a.redundancy = redundancy; a.raster = raster; a.norm_out = norm_out;
    for (int q = 0; q < MAX_THR; ++q) { a.on[q] = q < n_thr ? thr_on[q] : (T)0; a.off[q] = q < n_thr ? thr_off[q] : (T)0; }
    const size_t lds = lsm_fe::spikes_lds_bytes(time_bins, n_thr);
    LSM_REQUIRE(lds <= 160 * 1024, "latch bit rows of %zu bytes (time_bins x thresholds) exceed one CU's LDS", lds);
    auto fn = spec_to_spikes_kernel<T>;
    if (lds > 64 * 1024) lsm_allow_big_lds(reinterpret_cast<const void *>(fn));
    hipLaunchKernelGGL(fn, dim3(n_clips), dim3(LSM_SPK_THREADS), lds, (hipStream_t)stream, a);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

template <typename T>
int launch_encode(const T *spec, int n_rows, int n_bins, const T *thr_on, const T *thr_off,
                  int n_thr, uint8_t *out, void *stream)
{
    LSM_REQUIRE(n_rows >= 0 && n_bins >= 1, "encode: bad shape (%d rows, %d bins)", n_rows, n_bins);
    LSM_REQUIRE(n_thr >= 1 && n_thr <= MAX_THR, "n_thr=%d outside [1, %d]", n_thr, MAX_THR);
    if (n_rows == 0) return LSM_OK;
    LSM_REQUIRE(spec && out && thr_on && thr_off, "encode: null buffer");
    SpikeArgs<T> a{};
    for (int q = 0; q < MAX_THR; ++q) { a.on[q] = q < n_thr ? thr_on[q] : (T)0; a.off[q] = q < n_thr ? thr_off[q] : (T)0; }
    hipLaunchKernelGGL(encode_kernel<T>, dim3((n_rows + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       spec, n_rows, n_bins, n_thr, a, out);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

// Bit-packed rasters (File 1, packed variant): byte q of a row holds time steps 8q..8q+7, step 8q+k in
// bit k.  One thread per packed byte; rows whose length is a multiple of 8 move 8 raster bytes per
// thread as one 64-bit access (a wave then touches 512 contiguous raster bytes and 64 packed ones).
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *__restrict__ raster,
                                                        long n_rows, int T, int TP,
                                                        uint8_t *__restrict__ packed)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * TP) return;
    const long r = i / TP;
    const int q = (int)(i - r * TP);
    const uint8_t *src = raster + r * T + (long)q * 8;
    uint32_t byte = 0;
    if ((T & 7) == 0) {
        const uint2 v = *reinterpret_cast<const uint2 *>(src);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            byte |= ((v.x >> (8 * k)) & 0xFFu) ? (1u << k) : 0u;
            byte |= ((v.y >> (8 * k)) & 0xFFu) ? (1u << (k + 4)) : 0u;
        }
    } else {
        for (int k = 0; k < 8 && q * 8 + k < T; ++k) byte |= src[k] ? (1u << k) : 0u;
    }
    packed[i] = (uint8_t)byte;
}

__global__ __launch_bounds__(256) void unpack_bits_kernel(const uint8_t *__restrict__ packed,
                                                          long n_rows, int T, int TP,
                                                          uint8_t *__restrict__ raster)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * TP) return;
    const long r = i / TP;
    const int q = (int)(i - r * TP);
    const uint32_t byte = packed[i];
    uint8_t *dst = raster + r * T + (long)q * 8;
    if ((T & 7) == 0) {
        uint2 v;
        v.x = (byte & 1u) | ((byte & 2u) << 7) | ((byte & 4u) << 14) | ((byte & 8u) << 21);
        v.y = ((byte >> 4) & 1u) | ((byte & 32u) << 3) | ((byte & 64u) << 10) | ((byte & 128u) << 17);
        *reinterpret_cast<uint2 *>(dst) = v;
    } else {
        for (int k = 0; k < 8 && q * 8 + k < T; ++k) dst[k] = (uint8_t)((byte >> k) & 1u);
    }
}

}  // namespace

// CU count and LDS per CU of the current device (queried once per device)
struct DevInfo { int cus; long lds_per_cu; };
static DevInfo dev_info()
{
    constexpr int MAXDEV = 64;
    static DevInfo tab[MAXDEV];
    static std::once_flag once[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
        (void)hipGetLastError();
        return DevInfo{0, 0};
    }
    std::call_once(once[dev], [dev] {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess)
            tab[dev] = DevInfo{prop.multiProcessorCount, (long)prop.maxSharedMemoryPerMultiProcessor};
        else
            (void)hipGetLastError();
    });
    return tab[dev];
}

#define LSM_API extern "C" __attribute__((visibility("default")))

LSM_API int lsm_gammatone_spec_f64(const float *audio, int n_clips, int n_samples,
                                   const double *coefs, int n_filters, int nwin, int hop,
                                   int ncols, double *spec_out, double *db_out, int coef_flags,
                                   void *stream)
{
    LSM_REQUIRE(n_clips >= 0 && n_filters >= 1 && n_samples >= 1, "bad shape");
    // SPEC.md 1.1: one channel would take NumPy's pairwise window sums in the reference; these kernels sum element by element
    LSM_REQUIRE(n_filters >= 2, "the gammatone filterbank needs n_filters >= 2 (one channel: NumPy's pairwise window sums, "
                "SPEC.md 1.1)");
    if (n_clips == 0) return LSM_OK;
    LSM_REQUIRE(audio && coefs, "gammatone: null input");
    LSM_REQUIRE(spec_out || db_out, "gammatone: both outputs null");
    LSM_REQUIRE(nwin >= 1 && hop >= 1 && ncols >= 1, "bad window");
    LSM_REQUIRE(nwin <= NWIN_MAX * hop, "nwin=%d needs more than %d overlapping windows of hop=%d",
                nwin, NWIN_MAX, hop);
    LSM_REQUIRE((long)(ncols - 1) * hop + nwin <= n_samples, "columns exceed the clip");
    LSM_REQUIRE(n_samples >= 8, "clips shorter than 8 samples are not supported");
    const long n_waves = (long)((n_filters + 63) / 64) * n_clips;
#if LSM_EXPERIMENT_HOOKS                                // diagnostic builds only (exp/wpb_sweep.sh, exp/resv_sweep.sh)
    static const int wpb_env = [] {
        const char *e = getenv("LSM_GT_WPB");
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= GT_MAX_WPB ? v : 0;
    }();
    const int wpb = wpb_env ? wpb_env : 4;
    static const int lds_env = [] {
        const char *e = getenv("LSM_GT_LDS");
        const int v = e ? atoi(e) : -1;
        return v >= 0 && v <= 160 * 1024 ? v : -1;
    }();
#else
    constexpr int wpb = 4, lds_env = -1;
#endif
    // CU-exclusive placement for small launches.  The kernel uses no LDS; the reservation (half of
    // a CU's LDS plus 1 KB) only caps the dispatcher at ONE gammatone workgroup per CU, so launches
    // that overlap on other streams spread over the free CUs instead of stacking their waves on the
    // SIMDs an earlier launch already occupies (measured at 256 clips x 128 filters, 3 streams:
    // 0.85 -> 0.68 ms per launch, whole pipeline 1.18 -> 0.95 ms).  The other half of the LDS
    // stays free for the reservoir workgroups (21.5 KB each at cfg2; two or three per CU is best).  Launches with more workgroups than CUs need several
    // per CU and get no reservation.
    int lds = 0;
    {
        const DevInfo di = dev_info();
        const long n_wgs = (n_waves + wpb - 1) / wpb;
        if (di.cus > 0 && n_wgs <= di.cus && di.lds_per_cu >= 4096)
            lds = (int)(di.lds_per_cu / 2 + 1024);
    }
    if (lds_env >= 0) lds = lds_env;
    LSM_REQUIRE((n_waves + wpb - 1) / wpb <= 0x7fffffffL, "too many clips for one launch");
    const dim3 grid((unsigned)((n_waves + wpb - 1) / wpb)), block((unsigned)(64 * wpb));
    const bool fast = (coef_flags & 3) == 3;       // both properties verified by the host
    const int nw = (nwin + hop - 1) / hop;
#define LSM_GT(NW)                                                                            \
    {                                                                                         \
        if (lds > 64 * 1024)                                                                  \
            lsm_allow_big_lds(fast ? reinterpret_cast<const void *>(&gammatone_kernel<NW, true, true>)   \
                                   : reinterpret_cast<const void *>(&gammatone_kernel<NW, false, false>)); \
        if (fast)                                                                             \
            hipLaunchKernelGGL((gammatone_kernel<NW, true, true>), grid, block, lds,          \
                               (hipStream_t)stream, audio, n_clips, n_samples, coefs,         \
                               n_filters, nwin, hop, ncols, spec_out, db_out);                \
        else                                                                                  \
            hipLaunchKernelGGL((gammatone_kernel<NW, false, false>), grid, block, lds,        \
                               (hipStream_t)stream, audio, n_clips, n_samples, coefs,         \
                               n_filters, nwin, hop, ncols, spec_out, db_out);                \
    }
    switch (nw) {
    case 1: LSM_GT(1) break;
    case 2: LSM_GT(2) break;
    case 3: LSM_GT(3) break;
    default: LSM_GT(4) break;
    }
#undef LSM_GT
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

constexpr int GTF_DEFAULT_PRIO = 0;

// Layout of the fused launch for `n_filters`: chains per lane, waves per clip, waves per workgroup.
struct FusedPlan { int nch, groups, wpb; };
static bool fused_plan(int n_filters, int launch_flags, FusedPlan *p)
{
    // two chains per lane above 64 filters (fewest CU-cycles per clip), unless the caller asks for the low-latency
    // layout: one chain per lane = twice the waves, each half as long (1.47 against 2.43 ms for 256 clips x 128
    // filters on an idle GPU, for 12 % more CU time)
    int nch = (n_filters > 64 && !(launch_flags & 1)) ? 2 : 1;       // (bit 1 of the flags concerns the LDS reservation only)
#if LSM_EXPERIMENT_HOOKS
    static const int nch_env = [] { const char *e = getenv("LSM_GTF_NCH"); return e ? atoi(e) : 0; }();
    if (nch_env == 1 || nch_env == 2) nch = nch_env;
#endif
    // the low-latency flag is a layout hint, never a reason to refuse: 513..1024 filters need more than GT_MAX_WPB
    // one-chain waves per clip, so they keep the two-chain layout whatever the flag says
    if (nch == 1 && (n_filters + 63) / 64 > GT_MAX_WPB) nch = 2;
    const int groups = (n_filters + 64 * nch - 1) / (64 * nch);
    if (groups > GT_MAX_WPB) return false;
    int wpb = groups >= 4 ? groups : groups * (4 / groups);     // a multiple of `groups`, about 4 waves
#if LSM_EXPERIMENT_HOOKS
    static const int wpb_env = [] { const char *e = getenv("LSM_GTF_WPB"); return e ? atoi(e) : 0; }();
    if (wpb_env >= 1 && wpb_env <= GT_MAX_WPB && wpb_env % groups == 0) wpb = wpb_env;
#endif
    p->nch = nch; p->groups = groups; p->wpb = wpb;
    return true;
}

LSM_API long lsm_gammatone_spikes_workspace(int n_clips, int n_filters, int ncols)
{
    if (n_clips < 0 || n_filters < 1 || ncols < 1) return 0;
    // 64-channel groups padded to an even number: covers both the one- and the two-chain layout
    return (long)n_clips * ncols * (((n_filters + 127) / 128) * 2) * 64 * (long)sizeof(double);
}

LSM_API int lsm_gammatone_spikes_f64(const float *audio, int n_clips, int n_samples,
                                     const double *coefs, int n_filters, int nwin, int hop, int ncols,
                                     int time_bins, const double *thr_on, const double *thr_off, int n_thr,
                                     int redundancy, uint8_t *raster, void *workspace, long workspace_bytes,
                                     int coef_flags, int launch_flags, void *stream)
{
    LSM_REQUIRE((launch_flags & ~3) == 0, "launch_flags: only bit 0 (low-latency layout) and bit 1 (no LDS reservation) are defined");
    LSM_REQUIRE(n_clips >= 0 && n_filters >= 1 && n_samples >= 1, "bad shape");
    // SPEC.md 1.1: one channel would take NumPy's pairwise window sums in the reference; these kernels sum element by element
    LSM_REQUIRE(n_filters >= 2, "the gammatone filterbank needs n_filters >= 2 (one channel: NumPy's pairwise window sums, "
                "SPEC.md 1.1)");
    LSM_REQUIRE(nwin >= 1 && hop >= 1 && ncols >= 2 && time_bins >= 2, "bad window");
    LSM_REQUIRE(nwin <= NWIN_MAX * hop, "nwin=%d needs more than %d overlapping windows of hop=%d",
                nwin, NWIN_MAX, hop);
    LSM_REQUIRE((long)(ncols - 1) * hop + nwin <= n_samples, "columns exceed the clip");
    LSM_REQUIRE(n_samples >= 8, "clips shorter than 8 samples are not supported");
    LSM_REQUIRE(n_thr >= 1 && n_thr <= MAX_THR, "n_thr=%d outside [1, %d]", n_thr, MAX_THR);
    LSM_REQUIRE(redundancy >= 1, "redundancy must be >= 1");
    LSM_REQUIRE(thr_on && thr_off, "null threshold table");
    if (n_clips == 0) return LSM_OK;
    LSM_REQUIRE(audio && coefs && raster && workspace, "gammatone_spikes: null buffer");
    LSM_REQUIRE(workspace_bytes >= lsm_gammatone_spikes_workspace(n_clips, n_filters, ncols),
                "workspace of %ld bytes, need %ld (lsm_gammatone_spikes_workspace)", workspace_bytes,
                lsm_gammatone_spikes_workspace(n_clips, n_filters, ncols));
    LSM_REQUIRE(((uintptr_t)workspace & 7u) == 0, "workspace must be 8-byte aligned");
    const int row_bytes = time_bins * n_thr;
    LSM_REQUIRE((row_bytes & 3) != 0 || ((uintptr_t)raster & 3u) == 0,
                "the raster must be 4-byte aligned when a row is a multiple of 4 bytes");
    FusedPlan pl;
    if (!fused_plan(n_filters, launch_flags, &pl)) {
        lsm_set_error("gammatone_spikes: %d filters need more than %d waves per clip; use the split entry points",
                      n_filters, GT_MAX_WPB);
        return LSM_ERR_UNSUPPORTED;
    }
    FusedArgs a;
    a.audio = audio; a.coefs = coefs; a.ws = static_cast<double *>(workspace); a.raster = raster;
    a.n_clips = n_clips; a.n_samples = n_samples; a.n_filters = n_filters; a.nwin = nwin; a.hop = hop;
    a.ncols = ncols; a.time_bins = time_bins; a.n_thr = n_thr; a.redundancy = redundancy; a.groups = pl.groups;
    a.skip_epilogue = 0;
    a.prio = GTF_DEFAULT_PRIO;
#if LSM_EXPERIMENT_HOOKS
    static const int prio_env = [] { const char *e = getenv("LSM_GTF_PRIO"); return e ? atoi(e) : -1; }();
    if (prio_env >= 0 && prio_env <= 3) a.prio = prio_env;
    static const int skip_env = [] { const char *e = getenv("LSM_GTF_SKIP_EPILOGUE"); return e ? atoi(e) : 0; }();
    a.skip_epilogue = skip_env;
#endif
    // unused table entries never fire: nothing is > +inf or < -inf (the kernel compares 4 or 8 thresholds)
    for (int q = 0; q < MAX_THR; ++q) { a.on[q] = q < n_thr ? thr_on[q] : INFINITY; a.off[q] = q < n_thr ? thr_off[q] : -INFINITY; }
    const long n_waves = (long)pl.groups * n_clips;
    const long n_wgs = (n_waves + pl.wpb - 1) / pl.wpb;
    LSM_REQUIRE(n_wgs <= 0x7fffffffL, "too many clips for one launch");
    // per lane row: the comparison fields of pass A (4 bins per word up to 4 thresholds, else 2), reused for the raster bits
    // per lane row: the raster bits of the channel (time_bins * n_thr bits), staged for the coalesced write-out
    const int RW = (time_bins * n_thr + 31) / 32;
    long lds = 2 * GT_MAX_WPB * (long)sizeof(double) + (long)pl.wpb * pl.nch * 64 * RW * 4;
    LSM_REQUIRE(lds <= 160 * 1024, "raster stage of %ld bytes exceeds one CU's LDS", lds);
    // CU-exclusive placement of small launches, as in lsm_gammatone_spec_f64
    {
        const DevInfo di = dev_info();
        long resv = di.lds_per_cu / 2 + 1024;
#if LSM_EXPERIMENT_HOOKS
        static const long lds_env = [] { const char *e = getenv("LSM_GTF_LDS"); return e ? atol(e) : -1L; }();
        if (lds_env >= 0) resv = lds_env;
#endif
        // launch_flags bit 1: the caller runs LDS-hungry workgroups of another kernel beside this launch (the ring-row
        // reservoir kernel: 64-74 KB per clip, two per CU) -- no reservation then: the launch asks for what it uses
        // (27 KB per 4-wave workgroup at 4 thresholds x 100 bins) and fits beside such a pair
        if (!(launch_flags & 2) && di.cus > 0 && n_wgs <= di.cus && di.lds_per_cu >= 4096 && resv > lds && resv <= di.lds_per_cu)
            lds = resv;
    }
    const dim3 grid((unsigned)n_wgs), block((unsigned)(64 * pl.wpb));
    const bool fast = (coef_flags & 3) == 3;
    const bool shb0 = fast && (coef_flags & 4) != 0;       // one b0 for every channel: shared between the two chains
    const int nw = (nwin + hop - 1) / hop;
#define LSM_GTF2(NW, FAST, NCH, SHB0)                                                         \
    {                                                                                         \
        auto fn = gammatone_spikes_kernel<NW, FAST, NCH, SHB0>;                               \
        if (lds > 64 * 1024) lsm_allow_big_lds(reinterpret_cast<const void *>(fn));           \
        hipLaunchKernelGGL(fn, grid, block, (size_t)lds, (hipStream_t)stream, a);             \
    }
#define LSM_GTF(NW)                                                                           \
    {                                                                                         \
        if (fast) {                                                                           \
            if (pl.nch == 2) { if (shb0) LSM_GTF2(NW, true, 2, true) else LSM_GTF2(NW, true, 2, false) } \
            else LSM_GTF2(NW, true, 1, false)                                                 \
        } else {                                                                              \
            if (pl.nch == 2) LSM_GTF2(NW, false, 2, false) else LSM_GTF2(NW, false, 1, false) \
        }                                                                                     \
    }
    switch (nw) {
    case 1: LSM_GTF(1) break;
    case 2: LSM_GTF(2) break;
    case 3: LSM_GTF(3) break;
    default: LSM_GTF(4) break;
    }
#undef LSM_GTF
#undef LSM_GTF2
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

LSM_API int lsm_spec_to_spikes_f64(const double *db, int n_clips, int n_filters, int ncols,
                                   int time_bins, int apply_floor, const double *thr_on,
                                   const double *thr_off, int n_thr, int redundancy,
                                   uint8_t *raster, double *norm_out, void *stream)
{
    return launch_spec_to_spikes<double>(db, n_clips, n_filters, ncols, time_bins, apply_floor,
                                         thr_on, thr_off, n_thr, redundancy, raster, norm_out, stream);
}

LSM_API int lsm_spec_to_spikes_f32(const float *db, int n_clips, int n_filters, int ncols,
                                   int time_bins, int apply_floor, const float *thr_on,
                                   const float *thr_off, int n_thr, int redundancy,
                                   uint8_t *raster, float *norm_out, void *stream)
{
    return launch_spec_to_spikes<float>(db, n_clips, n_filters, ncols, time_bins, apply_floor,
                                        thr_on, thr_off, n_thr, redundancy, raster, norm_out, stream);
}

LSM_API int lsm_encode_hysteresis_f64(const double *spec, int n_rows, int n_bins,
                                      const double *thr_on, const double *thr_off, int n_thr,
                                      uint8_t *out, void *stream)
{
    return launch_encode<double>(spec, n_rows, n_bins, thr_on, thr_off, n_thr, out, stream);
}

LSM_API int lsm_encode_hysteresis_f32(const float *spec, int n_rows, int n_bins,
                                      const float *thr_on, const float *thr_off, int n_thr,
                                      uint8_t *out, void *stream)
{
    return launch_encode<float>(spec, n_rows, n_bins, thr_on, thr_off, n_thr, out, stream);
}

static int check_bits_args(const void *a, const void *b, const void *raster, long n_rows, int T,
                           const char *what)
{
    LSM_REQUIRE(n_rows >= 0 && T >= 1, "%s: bad shape", what);
    if (n_rows == 0) return 1;
    LSM_REQUIRE(a && b, "%s: null buffer", what);
    // rows of a multiple of 8 steps move as 64-bit words; other lengths go byte by byte
    LSM_REQUIRE((T & 7) != 0 || ((uintptr_t)raster & 7u) == 0,
                "%s: the raster must be 8-byte aligned when n_steps is a multiple of 8", what);
    LSM_REQUIRE(n_rows * (long)((T + 7) / 8) <= 0x7fffffffL * 256L, "%s: too many rows", what);
    return LSM_OK;
}

LSM_API int lsm_raster_pack_bits(const uint8_t *raster, long n_rows, int n_steps, uint8_t *packed,
                                 void *stream)
{
    const int rc = check_bits_args(raster, packed, raster, n_rows, n_steps, "pack_bits");
    if (rc != LSM_OK) return rc == 1 ? LSM_OK : rc;
    const int TP = (n_steps + 7) / 8;
    const long n = n_rows * TP;
    hipLaunchKernelGGL(pack_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, raster, n_rows, n_steps, TP, packed);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

LSM_API int lsm_raster_unpack_bits(const uint8_t *packed, long n_rows, int n_steps, uint8_t *raster,
                                   void *stream)
{
    const int rc = check_bits_args(packed, raster, raster, n_rows, n_steps, "unpack_bits");
    if (rc != LSM_OK) return rc == 1 ? LSM_OK : rc;
    const int TP = (n_steps + 7) / 8;
    const long n = n_rows * TP;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, packed, n_rows, n_steps, TP, raster);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}
