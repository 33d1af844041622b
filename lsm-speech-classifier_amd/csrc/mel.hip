// Mel branch of the front end for gfx950 (SPEC.md §1.5).
//
// Replaces librosa.feature.melspectrogram(y, sr=16000, n_mels=F, hop_length=hop) and
// librosa.power_to_db(S, ref=np.max) as called at /root/reference/create_dataset.py:44-48 (librosa
// 0.11 defaults: n_fft = win_length = 2048, periodic Hann, centred frames with zero padding,
// power 2, Slaney mel basis; amin 1e-10, top_db 80).  librosa multiplies the float64 window into the
// float32 frame, transforms in float64 and stores complex64; the kernel follows the same dtypes.
// One WAVE = one frame (round 5; profiles/r05_mel_power_parts.txt: with a 256-thread workgroup per frame and five
// radix-4 passes the kernel spent 71 % of its time waiting at LDS round trips and workgroup barriers): the real-input
// FFT as 1024 complex points, sixteen per lane in registers, passes of radix 16, 16 and 4 through one 17 KB LDS
// buffer with no barrier; |.|^2 in float32; then the triangular mel filters, four lanes per filter.
#include "lsm_common.h"
#include "spikes_body.h"
#include <cstdlib>

namespace {

#ifndef LSM_MEL_ABLATE
#define LSM_MEL_ABLATE 0    // diagnostic builds only (WRONG results): 1 = no FFT passes, 2 = no mel projection (every filter gets
#endif                      // its first bin), 4 = no unpack / power (profiles/r05_mel_power_parts.txt: where the kernel's time goes)

constexpr int NFFT = 2048;
constexpr int N2 = NFFT / 2;                 // the real frame is transformed as N2 complex points
constexpr int NBINS = NFFT / 2 + 1;
constexpr int ZPAD = N2 + N2 / 16;           // a wave's point buffer: one pad element after every 16 (see zpad)

// W_2048^m for any m in [0, 2048) from the table of the first 1024 powers (W^(m+1024) = -W^m)
__device__ __forceinline__ double2 tw2048(const double2 *__restrict__ t, int m)
{
    const double2 w = t[m & (N2 - 1)];
    return (m & N2) ? make_double2(-w.x, -w.y) : w;
}
__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// four-point transform in place: a_c <- sum_q a_q W_4^(q c), W_4 = -i
__device__ __forceinline__ void bfly4(double2 &a0, double2 &a1, double2 &a2, double2 &a3)
{
    const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3);
    const double2 t3 = make_double2(a1.y - a3.y, -(a1.x - a3.x));          // -i (a1 - a3)
    a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}
// sixteen-point transform in registers, as 4 x 4: input q = 4a + b sits in u[q]; output r = c + 4d ends in u[d + 4c]
__device__ __forceinline__ void bfly16(double2 (&u)[16])
{
    constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H = 0.70710678118654752440;
#pragma unroll
    for (int b = 0; b < 4; ++b) bfly4(u[b], u[b + 4], u[b + 8], u[b + 12]);     // u[b + 4c] = sum_a u[4a + b] W_4^(a c)
    // times W_16^(b c)
    u[1 + 4] = cmul(u[1 + 4], make_double2(C1, -S1));
    u[1 + 8] = cmul(u[1 + 8], make_double2(H, -H));
    u[1 + 12] = cmul(u[1 + 12], make_double2(S1, -C1));
    u[2 + 4] = cmul(u[2 + 4], make_double2(H, -H));
    u[2 + 8] = make_double2(u[2 + 8].y, -u[2 + 8].x);                          // W_16^4 = -i
    u[2 + 12] = cmul(u[2 + 12], make_double2(-H, -H));
    u[3 + 4] = cmul(u[3 + 4], make_double2(S1, -C1));
    u[3 + 8] = cmul(u[3 + 8], make_double2(-H, -H));
    u[3 + 12] = cmul(u[3 + 12], make_double2(-C1, S1));
#pragma unroll
    for (int c = 0; c < 4; ++c) bfly4(u[4 * c], u[4 * c + 1], u[4 * c + 2], u[4 * c + 3]);
}
// LDS element of point e: a pad element after every sixteen points, so that the sixteen consecutive points a lane writes
// in the first pass (and the runs of sixteen of the second) start in different banks from lane to lane
__device__ __forceinline__ int zpad(int e) { return e + (e >> 4); }
// a wave's LDS accesses of one pass are seen by its other lanes in the next (LDS keeps a wave's order; this keeps the compiler's)
__device__ __forceinline__ void mel_wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// W_32^i = exp(-2 pi i / 32), i <= 16, as {cos, sin} of pi i / 16
__device__ constexpr double W32_COS[17] = {1.0, 0.98078528040323044913, 0.92387953251128673848, 0.83146961230254523708,
                                           0.70710678118654752440, 0.55557023301960222474, 0.38268343236508978178,
                                           0.19509032201612826785, 0.0, -0.19509032201612826785, -0.38268343236508978178,
                                           -0.55557023301960222474, -0.70710678118654752440, -0.83146961230254523708,
                                           -0.92387953251128673848, -0.98078528040323044913, -1.0};
__device__ constexpr double W32_SIN[17] = {0.0, 0.19509032201612826785, 0.38268343236508978178, 0.55557023301960222474,
                                           0.70710678118654752440, 0.83146961230254523708, 0.92387953251128673848,
                                           0.98078528040323044913, 1.0, 0.98078528040323044913, 0.92387953251128673848,
                                           0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
                                           0.38268343236508978178, 0.19509032201612826785, 0.0};

struct MelArgs {
    const float *audio;
    int n_samples, hop, n_frames, n_mels;
    const double *window;
    const double2 *twiddle;     // W_2048^k, k < 1024
    const float *basis;
    const int *lo, *hi;
    float *power_out;
};

// One wave = one frame (clip b, frame t).  The 2048 windowed real samples are packed as 1024 complex points
// z[n] = x[2n] + i x[2n+1]; lane l holds the points l + 64 s, s < 16, of every pass.  Stockham passes (natural order in and
// out) with strides p = 1 (radix 16, inputs straight from memory, no twiddles), p = 16 (radix 16) and p = 256 (radix 4, four
// butterflies per lane): out[j + r p] = sum_q W_R^(q r) W_(R p)^(q k) in[i + q N/R], k = i mod p, j = (i - k) R + k.  Then
// the 1025 bins of the real transform, X[k] = E[k] - i W^k O[k], E/O = (Z[k] +- conj Z[N2-k]) / 2.  All in float64.
// `z`: this wave's ZPAD points of LDS; the power values reuse its first NBINS floats.  (Twiddle products instead of table entries move the float64
// spectrum by a few 1e-16 relative: the table's own rounding.)
__device__ __forceinline__ void mel_frame_wave(const MelArgs &a, double2 *z, const int b, const int t)
{
    float *pw = reinterpret_cast<float *>(z);
    const float *__restrict__ audio = a.audio;
    const int n_samples = a.n_samples, n_frames = a.n_frames, n_mels = a.n_mels;
    const double *__restrict__ window = a.window;
    const double2 *__restrict__ twiddle = a.twiddle;
    const float *__restrict__ basis = a.basis;
    const int *__restrict__ lo = a.lo, *__restrict__ hi = a.hi;
    float *__restrict__ power_out = a.power_out;
    const int lane = threadIdx.x & 63;
    const float *clip = audio + (size_t)b * n_samples;
    const int start = t * a.hop - NFFT / 2;     // centred frame, zero padding outside the clip

    // Every twiddle factor of the frame is a product of EIGHT table entries fetched here, with the samples (one exposed memory
    // latency per frame), and of constants: W_2048^(8 q k) from the powers 1, 2, 4, 8 of W_256^k (second pass), W_2048^(2 q (l + 64 c))
    // = W_1024^(q l) W_16^(q c) (third pass), W_2048^(l + 64 i) = W_2048^l W_32^i (unpacking).
    const int k = lane & 15;
    const double2 wb1 = twiddle[8 * k], wb2 = twiddle[16 * k], wb4 = twiddle[32 * k], wb8 = twiddle[64 * k];
    const double2 wc1 = twiddle[2 * lane], wc2 = twiddle[4 * lane], wc3 = twiddle[6 * lane];
    const double2 wu = twiddle[lane];
    double2 u[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int n = lane + 64 * s;
        // (samples outside the clip: the load goes to a clamped index and its value is dropped -- a guarded load is a branch
        //  with a wait of its own, and thirty-two of them in a row were most of this kernel's time)
        const int s0 = start + 2 * n, s1 = s0 + 1;
        const float x0 = clip[min(max(s0, 0), n_samples - 1)], x1 = clip[min(max(s1, 0), n_samples - 1)];
        const double v0 = (s0 >= 0 && s0 < n_samples) ? (double)x0 : 0.0;
        const double v1 = (s1 >= 0 && s1 < n_samples) ? (double)x1 : 0.0;
        const double2 wn = *reinterpret_cast<const double2 *>(window + 2 * n);
        u[s] = make_double2(wn.x * v0, wn.y * v1);
    }
    if (!(LSM_MEL_ABLATE & 1)) {
        constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H = 0.70710678118654752440;
        // p = 1: i = lane, k = 0, j = 16 lane
        bfly16(u);
#pragma unroll
        for (int s = 0; s < 16; ++s) z[zpad(16 * lane + (s >> 2) + 4 * (s & 3))] = u[s];
        mel_wave_fence();
        // p = 16: i = lane, k = lane mod 16, j = (lane - k) 16 + k; twiddles W_256^(q k) = W_2048^(8 q k)
#pragma unroll
        for (int s = 0; s < 16; ++s) u[s] = z[zpad(lane + 64 * s)];
        {
            const double2 w3 = cmul(wb2, wb1), w5 = cmul(wb4, wb1), w6 = cmul(wb4, wb2), w7 = cmul(wb4, w3);
            u[1] = cmul(u[1], wb1); u[2] = cmul(u[2], wb2); u[3] = cmul(u[3], w3); u[4] = cmul(u[4], wb4);
            u[5] = cmul(u[5], w5); u[6] = cmul(u[6], w6); u[7] = cmul(u[7], w7); u[8] = cmul(u[8], wb8);
            u[9] = cmul(u[9], cmul(wb8, wb1)); u[10] = cmul(u[10], cmul(wb8, wb2)); u[11] = cmul(u[11], cmul(wb8, w3));
            u[12] = cmul(u[12], cmul(wb8, wb4)); u[13] = cmul(u[13], cmul(wb8, w5)); u[14] = cmul(u[14], cmul(wb8, w6));
            u[15] = cmul(u[15], cmul(wb8, w7));
        }
        bfly16(u);
        mel_wave_fence();                       // every lane has read its points before any is overwritten
        const int j = (lane - k) * 16 + k;
#pragma unroll
        for (int s = 0; s < 16; ++s) z[zpad(j + 16 * ((s >> 2) + 4 * (s & 3)))] = u[s];
        mel_wave_fence();
        // p = 256, radix 4: butterflies i = lane + 64 c, k = i, j = i; twiddles W_1024^(q k) = W_1024^(q lane) W_16^(q c)
#pragma unroll
        for (int s = 0; s < 16; ++s) u[s] = z[zpad(lane + 64 * s)];
        mel_wave_fence();
        const double2 w16[10] = {make_double2(1.0, 0.0), make_double2(C1, -S1), make_double2(H, -H), make_double2(S1, -C1),
                                 make_double2(0.0, -1.0), make_double2(0.0, 0.0), make_double2(-H, -H), make_double2(0.0, 0.0),
                                 make_double2(0.0, 0.0), make_double2(-C1, S1)};        // W_16^m for m = q c
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int kk = lane + 64 * c;
            if (c == 0) {
                u[4] = cmul(u[4], wc1); u[8] = cmul(u[8], wc2); u[12] = cmul(u[12], wc3);
            } else {
                u[c + 4] = cmul(u[c + 4], cmul(wc1, w16[c]));
                u[c + 8] = cmul(u[c + 8], cmul(wc2, w16[2 * c]));
                u[c + 12] = cmul(u[c + 12], cmul(wc3, w16[3 * c]));
            }
            bfly4(u[c], u[c + 4], u[c + 8], u[c + 12]);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[zpad(kk + 256 * r)] = u[c + 4 * r];
        }
    } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) z[zpad(lane + 64 * s)] = u[s];
    }
    mel_wave_fence();
    // The 1025 power values go where the points were (`pw` aliases `z`): every lane forms its seventeen first, then they are stored.
    float pv[17];
#pragma unroll
    for (int i = 0; i <= 16; ++i) {
        const int f = lane + 64 * i;
        pv[i] = 0.0f;
        if (!(LSM_MEL_ABLATE & 4) && (i < 16 || lane == 0)) {
            const double2 zk = z[zpad(f & (N2 - 1))];
            const double2 zr = z[zpad((N2 - f) & (N2 - 1))];
            const double2 E = make_double2(0.5 * (zk.x + zr.x), 0.5 * (zk.y - zr.y));    // (Zk + conj Zr)/2
            const double2 O = make_double2(0.5 * (zk.x - zr.x), 0.5 * (zk.y + zr.y));    // (Zk - conj Zr)/2
            const double2 wf = i == 0 ? wu : cmul(wu, make_double2(W32_COS[i], -W32_SIN[i]));      // W_2048^f
            const double2 D = cmul(wf, O);
            const float re = (float)(E.x + D.y), im = (float)(E.y - D.x);               // complex64 storage
            const float mag = hypotf(re, im);                   // np.abs on complex64
            pv[i] = mag * mag;                                  // ** 2.0 in float32
        }
    }
    mel_wave_fence();                           // every lane has read its points
#pragma unroll
    for (int i = 0; i <= 16; ++i)
        if (i < 16 || lane == 0) pw[lane + 64 * i] = pv[i];
    mel_wave_fence();
    // mel projection: four lanes per filter, lane q takes bins lo+q, lo+q+4, ... (ascending), the four
    // partial sums are combined as (p0 + p1) + (p2 + p3)
    for (int m0 = 0; m0 < n_mels; m0 += 16) {
        const int m = m0 + (lane >> 2), q = lane & 3;
        float acc = 0.0f;
        if (m < n_mels) {
            const float *row = basis + (size_t)m * NBINS;
            const int h = (LSM_MEL_ABLATE & 2) ? lo[m] + 1 : hi[m];
            int f = lo[m] + q;
            for (; f + 28 < h; f += 32) {       // eight terms at a time, their loads together; the additions keep their order
                float r[8], pq[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { r[e] = row[f + 4 * e]; pq[e] = pw[f + 4 * e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += r[e] * pq[e];
            }
            if (f + 12 < h) {
                const float r0 = row[f], r1 = row[f + 4], r2 = row[f + 8], r3 = row[f + 12];
                const float p0 = pw[f], p1 = pw[f + 4], p2 = pw[f + 8], p3 = pw[f + 12];
                acc += r0 * p0; acc += r1 * p1; acc += r2 * p2; acc += r3 * p3;
                f += 16;
            }
            for (; f < h; f += 4) acc += row[f] * pw[f];
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (m < n_mels && q == 0) power_out[((size_t)b * n_mels + m) * n_frames + t] = acc;
    }
}

// grid = (frames, clips), one wave each
__global__ __launch_bounds__(64) void mel_power_kernel(const MelArgs a)
{
    __shared__ double2 z[ZPAD];                 // 17 KB
    mel_frame_wave(a, z, (int)blockIdx.y, (int)blockIdx.x);
}

// librosa.power_to_db(S, ref=np.max): per clip, float32; one 256-thread workgroup, `red`: 4 floats of LDS.
__device__ __forceinline__ void power_to_db_body(const float *__restrict__ p, float *__restrict__ o, int n, float amin,
                                                 float top_db, float *red)
{
    float mx = -INFINITY;
    int nan_seen = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = p[i];
        mx = fmaxf(mx, v);
        nan_seen |= v != v;
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    // np.max / np.maximum propagate NaN (fmaxf drops it): one NaN power value makes the reference level, and with it every
    // dB value of the clip, NaN (librosa.power_to_db(S, ref=np.max), SPEC.md 1.5)
    const int any_nan = __syncthreads_or(nan_seen);
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (any_nan) {
        for (int i = threadIdx.x; i < n; i += 256) o[i] = NAN;
        return;
    }
    const float refdb = 10.0f * log10f(fmaxf(amin, mx));
    // the maximum of (10*log10(max(amin, S)) - refdb) is exactly 0, so the floor is -top_db
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = 10.0f * log10f(fmaxf(amin, p[i])) - refdb;
        o[i] = fmaxf(v, 0.0f - top_db);
    }
}

__global__ __launch_bounds__(256) void power_to_db_kernel(const float *__restrict__ power, int n,
                                                          float amin, float top_db,
                                                          float *__restrict__ db_out)
{
    __shared__ float red[4];
    power_to_db_body(power + (size_t)blockIdx.x * n, db_out + (size_t)blockIdx.x * n, n, amin, top_db, red);
}

// The whole mel front end of a batch in ONE launch (create_dataset.py:43-48 + :62-104 per clip): grid = (frames, clips)
// as in mel_power_kernel; every workgroup transforms its frame and projects it onto the mel filters, then counts itself
// in on its clip.  The workgroup that arrives LAST for a clip -- all of the clip's power values are in memory then --
// finishes the clip: power_to_db (the clip's maximum is the reference), min-max normalise, resize, hysteresis latches,
// raster (lsm_fe::spec_to_spikes_body, the code of the split path).  No workgroup ever waits for another one: the
// count is one atomic add whose returned value says who is last (the pattern MI355X_MICROARCH.md tabulates for a
// consumer "told by the value its add returned"; the adder's stores are made visible by the fence before the add,
// the finisher's loads are ordered by the fence after it).  The launch function zeroes the counters on the stream first.
struct MelSpikeArgs {
    MelArgs mel;                        // power_out = the power workspace (n_clips, n_mels, n_frames)
    lsm_fe::SpikeArgs<float> sp;        // db = the dB workspace, same shape: written and read by the finishing workgroup
    unsigned int *counters;             // (n_clips) zero
    float amin, top_db;
    int frames_per_wg;                  // a workgroup transforms this many consecutive frames: grid.x = ceil(n_frames / it)
};

constexpr size_t MEL_SPIKES_LDS = 4 * sizeof(double2) * ZPAD + 32;

__global__ __launch_bounds__(256) void mel_spikes_kernel(const MelSpikeArgs a)
{
    // four waves, a point buffer each (dynamic: 68 KB); the finishing workgroup reuses them as the raster stage
    extern __shared__ __attribute__((aligned(16))) unsigned char mel_lds[];
    double2 *buf = reinterpret_cast<double2 *>(mel_lds);
    float *red = reinterpret_cast<float *>(mel_lds + 4 * sizeof(double2) * ZPAD);       // 4 floats
    int &last = *reinterpret_cast<int *>(red + 4);
    const int b = blockIdx.y, tid = threadIdx.x, w = tid >> 6;
    // A workgroup takes `frames_per_wg` consecutive frames, its four waves one frame each at a time: the release fence below
    // writes the XCD's dirty L2 lines back (the eight L2s are not coherent with each other), which costs about a microsecond,
    // so it is paid once per group of frames, not once per frame (one frame per workgroup: 2.4 instead of 0.3 ms per
    // 200 clips, profiles/r04_mel_one_launch_ab.txt).
    const int t0 = (int)blockIdx.x * a.frames_per_wg;
    const int t1 = min(t0 + a.frames_per_wg, a.mel.n_frames);
    for (int t = t0 + w; t < t1; t += 4) {
        mel_frame_wave(a.mel, buf + w * ZPAD, b, t);
        mel_wave_fence();                       // the wave's LDS buffers are free again
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // my power values are visible device-wide before my count is
    __syncthreads();
    if (tid == 0) {
        const unsigned int prev = atomicAdd(a.counters + b, 1u);
        last = prev == gridDim.x - 1u;
        if (last) a.counters[b] = 0u;           // every workgroup of the clip has counted: reset for the next launch
    }
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the other workgroups' values, not whatever this CU may have cached
    const int n = a.mel.n_mels * a.mel.n_frames;
    float *db = const_cast<float *>(a.sp.db) + (size_t)b * n;
    power_to_db_body(a.mel.power_out + (size_t)b * n, db, n, a.amin, a.top_db, red);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // dB values written by other threads of this workgroup
    __syncthreads();
    lsm_fe::spec_to_spikes_body<float>(a.sp, b, mel_lds);
}

}  // namespace

#define LSM_API extern "C" __attribute__((visibility("default")))

LSM_API int lsm_mel_power_f32(const float *audio, int n_clips, int n_samples, int n_fft, int hop,
                              int n_frames, const double *window_dev, const double *twiddle_dev,
                              const float *basis_dev, const int32_t *lo_dev, const int32_t *hi_dev,
                              int n_mels, float *power_out, void *stream)
{
    LSM_REQUIRE(n_clips >= 0 && n_samples >= 1 && hop >= 1 && n_frames >= 1 && n_mels >= 1, "bad shape");
    LSM_REQUIRE(n_fft == NFFT, "n_fft must be %d (librosa's default), got %d", NFFT, n_fft);
    LSM_REQUIRE(n_clips <= 65535, "at most 65535 clips per call (grid.y)");
    if (n_clips == 0) return LSM_OK;
    LSM_REQUIRE(audio && window_dev && twiddle_dev && basis_dev && lo_dev && hi_dev && power_out,
                "mel: null buffer");
    LSM_REQUIRE((((uintptr_t)window_dev | (uintptr_t)twiddle_dev) & 15u) == 0, "mel: window and twiddle tables must be 16-byte aligned");
    MelArgs a;
    a.audio = audio; a.n_samples = n_samples; a.hop = hop; a.n_frames = n_frames; a.n_mels = n_mels;
    a.window = window_dev; a.twiddle = reinterpret_cast<const double2 *>(twiddle_dev); a.basis = basis_dev;
    a.lo = lo_dev; a.hi = hi_dev; a.power_out = power_out;
    hipLaunchKernelGGL(mel_power_kernel, dim3(n_frames, n_clips), dim3(64), 0, (hipStream_t)stream, a);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

LSM_API long lsm_mel_spikes_workspace(int n_clips, int n_mels, int n_frames)
{
    if (n_clips < 0 || n_mels < 1 || n_frames < 1) return 0;
    // power + dB spectrograms (float32 each) + one counter per clip, the counters first (they must start zeroed)
    return (long)(((size_t)n_clips * 4 + 255) / 256 * 256) + 2L * n_clips * n_mels * n_frames * (long)sizeof(float);
}

LSM_API int lsm_mel_spikes_f32(const float *audio, int n_clips, int n_samples, int n_fft, int hop, int n_frames,
                               const double *window_dev, const double *twiddle_dev, const float *basis_dev,
                               const int32_t *lo_dev, const int32_t *hi_dev, int n_mels, float amin, float top_db,
                               int time_bins, const float *thr_on, const float *thr_off, int n_thr, int redundancy,
                               uint8_t *raster, void *workspace, long workspace_bytes, void *stream)
{
    LSM_REQUIRE(n_clips >= 0 && n_samples >= 1 && hop >= 1 && n_frames >= 2 && n_mels >= 1 && time_bins >= 2, "bad shape");
    LSM_REQUIRE(n_fft == NFFT, "n_fft must be %d (librosa's default), got %d", NFFT, n_fft);
    LSM_REQUIRE(n_clips <= 65535, "at most 65535 clips per call (grid.y)");
    LSM_REQUIRE(n_thr >= 1 && n_thr <= lsm_fe::MAX_THR, "n_thr=%d outside [1, %d]", n_thr, lsm_fe::MAX_THR);
    LSM_REQUIRE(redundancy >= 1, "redundancy must be >= 1");
    LSM_REQUIRE(thr_on && thr_off, "null threshold table");
    if (n_clips == 0) return LSM_OK;
    LSM_REQUIRE(audio && window_dev && twiddle_dev && basis_dev && lo_dev && hi_dev && raster && workspace,
                "mel_spikes: null buffer");
    LSM_REQUIRE((((uintptr_t)window_dev | (uintptr_t)twiddle_dev) & 15u) == 0, "mel_spikes: window and twiddle tables must be 16-byte aligned");
    LSM_REQUIRE(workspace_bytes >= lsm_mel_spikes_workspace(n_clips, n_mels, n_frames),
                "workspace of %ld bytes, need %ld (lsm_mel_spikes_workspace)", workspace_bytes,
                lsm_mel_spikes_workspace(n_clips, n_mels, n_frames));
    LSM_REQUIRE(((uintptr_t)workspace & 255u) == 0, "workspace must be 256-byte aligned");
    const int row_bytes = time_bins * n_thr;
    LSM_REQUIRE((row_bytes & 3) != 0 || ((uintptr_t)raster & 3u) == 0,
                "the raster must be 4-byte aligned when a row is a multiple of 4 bytes");
    // the finishing workgroup keeps its latch bit rows in the waves' point buffers
    const size_t stage = lsm_fe::spikes_lds_bytes(time_bins, n_thr);
    if (stage > 4 * sizeof(double2) * ZPAD) {
        lsm_set_error("mel_spikes: latch bit rows of %zu bytes exceed the kernel's %zu; use the split entry points", stage,
                      4 * sizeof(double2) * ZPAD);
        return LSM_ERR_UNSUPPORTED;
    }
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    const size_t cbytes = ((size_t)n_clips * 4 + 255) / 256 * 256;
    float *power = reinterpret_cast<float *>(ws + cbytes);
    float *db = power + (size_t)n_clips * n_mels * n_frames;
    MelSpikeArgs a;
    a.mel.audio = audio; a.mel.n_samples = n_samples; a.mel.hop = hop; a.mel.n_frames = n_frames; a.mel.n_mels = n_mels;
    a.mel.window = window_dev; a.mel.twiddle = reinterpret_cast<const double2 *>(twiddle_dev); a.mel.basis = basis_dev;
    a.mel.lo = lo_dev; a.mel.hi = hi_dev; a.mel.power_out = power;
    a.sp.db = db; a.sp.n_clips = n_clips; a.sp.n_filters = n_mels; a.sp.ncols = n_frames; a.sp.time_bins = time_bins;
    a.sp.apply_floor = 0;                       // the reference floors the gammatone branch only (create_dataset.py:60)
    a.sp.n_thr = n_thr; a.sp.redundancy = redundancy; a.sp.raster = raster; a.sp.norm_out = nullptr;
    for (int q = 0; q < lsm_fe::MAX_THR; ++q) { a.sp.on[q] = q < n_thr ? thr_on[q] : 0.0f; a.sp.off[q] = q < n_thr ? thr_off[q] : 0.0f; }
    a.counters = reinterpret_cast<unsigned int *>(ws);
    a.amin = amin; a.top_db = top_db;
    // Frames per workgroup (profiles/r04_mel_one_launch_ab.txt, 40 filters x 200 clips, ms per step front ends alone /
    // on one stream): 1 frame 1.72 / 1.75, 4: 0.89 / 0.79, 8: 0.58 / 0.60, 16: 0.45 / 0.54, 32: 0.36 / 0.55, the whole clip
    // (101) 0.30 / 0.90; the three split launches 0.28 / 0.33.  A batch that gives every other CU a clip takes the whole
    // clip per workgroup (best throughput when launches overlap), smaller batches 32 frames (more workgroups).
    int fpw = n_clips >= 128 ? n_frames : 32;
#if LSM_EXPERIMENT_HOOKS
    static const int fpw_env = [] { const char *e = getenv("LSM_MEL_FRAMES_PER_WG"); return e ? atoi(e) : 0; }();
    if (fpw_env >= 1) fpw = fpw_env;
#endif
    fpw = fpw > n_frames ? n_frames : fpw;
    a.frames_per_wg = fpw;
    // The per-clip arrival counters are zeroed HERE, on the launch stream (4 bytes per clip; a memset node when captured): the
    // kernel resets a counter only when a clip's last workgroup arrives, so a launch that failed, or a workspace the caller
    // did not zero, would otherwise leave every later launch without a finishing workgroup.
    LSM_CHECK_HIP(hipMemsetAsync(a.counters, 0, (size_t)n_clips * sizeof(unsigned int), (hipStream_t)stream));
    lsm_allow_big_lds(reinterpret_cast<const void *>(mel_spikes_kernel));       // 68 KB of dynamic LDS
    hipLaunchKernelGGL(mel_spikes_kernel, dim3((n_frames + fpw - 1) / fpw, n_clips), dim3(256), MEL_SPIKES_LDS, (hipStream_t)stream, a);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

LSM_API int lsm_power_to_db_f32(const float *power, int n_clips, int n_per_clip, float amin,
                                float top_db, float *db_out, void *stream)
{
    LSM_REQUIRE(n_clips >= 0 && n_per_clip >= 1, "bad shape");
    if (n_clips == 0) return LSM_OK;
    LSM_REQUIRE(power && db_out, "power_to_db: null buffer");
    hipLaunchKernelGGL(power_to_db_kernel, dim3(n_clips), dim3(256), 0, (hipStream_t)stream, power,
                       n_per_clip, amin, top_db, db_out);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}
