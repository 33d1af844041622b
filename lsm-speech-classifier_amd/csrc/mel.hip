// Mel branch of the front end for gfx950 (SPEC.md §1.5).
//
// Replaces librosa.feature.melspectrogram(y, sr=16000, n_mels=F, hop_length=hop) and
// librosa.power_to_db(S, ref=np.max) as called at /root/reference/create_dataset.py:44-48 (librosa
// 0.11 defaults: n_fft = win_length = 2048, periodic Hann, centred frames with zero padding,
// power 2, Slaney mel basis; amin 1e-10, top_db 80).  librosa multiplies the float64 window into the
// float32 frame, transforms in float64 and stores complex64; the kernel follows the same dtypes.
// One workgroup = one frame: real-input FFT as 1024 complex points, five radix-4 passes in LDS (float64),
// |.|^2 in float32, then the triangular mel filters (four lanes per filter).
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

// One workgroup = one frame.  The 2048 windowed real samples are packed as 1024 complex points
// z[n] = x[2n] + i x[2n+1], transformed by five radix-4 Stockham passes (one butterfly per thread and
// pass, ping-pong between two LDS buffers, natural order in and out), and unpacked to the 1025 bins of
// the real transform: X[k] = E[k] - i W^k O[k], E/O = (Z[k] +- conj Z[N2-k]) / 2.  All in float64.
struct MelArgs {
    const float *audio;
    int n_samples, hop, n_frames, n_mels;
    const double *window;
    const double2 *twiddle;     // W_2048^k, k < 1024
    const float *basis;
    const int *lo, *hi;
    float *power_out;
};

__device__ __forceinline__ void mel_power_body(const MelArgs &a, double2 (*buf)[N2], float *pw, const int t)
{
    const float *__restrict__ audio = a.audio;
    const int n_samples = a.n_samples, hop = a.hop, n_frames = a.n_frames, n_mels = a.n_mels;
    const double *__restrict__ window = a.window;
    const double2 *__restrict__ twiddle = a.twiddle;
    const float *__restrict__ basis = a.basis;
    const int *__restrict__ lo = a.lo, *__restrict__ hi = a.hi;
    float *__restrict__ power_out = a.power_out;
    const int b = blockIdx.y, tid = threadIdx.x;
    const float *clip = audio + (size_t)b * n_samples;
    const int start = t * hop - NFFT / 2;       // centred frame, zero padding outside the clip

    for (int n = tid; n < N2; n += 256) {
        const int s0 = start + 2 * n, s1 = s0 + 1;
        const double v0 = (s0 >= 0 && s0 < n_samples) ? (double)clip[s0] : 0.0;
        const double v1 = (s1 >= 0 && s1 < n_samples) ? (double)clip[s1] : 0.0;
        buf[0][n] = make_double2(window[2 * n] * v0, window[2 * n + 1] * v1);
    }
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int p = 1; p < ((LSM_MEL_ABLATE & 1) ? 1 : N2); p <<= 2) {          // p = 1, 4, 16, 64, 256
        const double2 *in = buf[cur];
        double2 *out = buf[cur ^ 1];
        const int k = tid & (p - 1);
        const int j = ((tid - k) << 2) + k;
        const int e = k * (512 / p);            // exponent of W_2048 for this butterfly's first twiddle
        const double2 u0 = in[tid];
        const double2 u1 = cmul(in[tid + 256], tw2048(twiddle, e));
        const double2 u2 = cmul(in[tid + 512], tw2048(twiddle, 2 * e));
        const double2 u3 = cmul(in[tid + 768], tw2048(twiddle, 3 * e));
        const double2 a0 = make_double2(u0.x + u2.x, u0.y + u2.y);
        const double2 a1 = make_double2(u0.x - u2.x, u0.y - u2.y);
        const double2 a2 = make_double2(u1.x + u3.x, u1.y + u3.y);
        const double2 a3 = make_double2(u1.y - u3.y, -(u1.x - u3.x));      // -i (u1 - u3)
        out[j] = make_double2(a0.x + a2.x, a0.y + a2.y);
        out[j + p] = make_double2(a1.x + a3.x, a1.y + a3.y);
        out[j + 2 * p] = make_double2(a0.x - a2.x, a0.y - a2.y);
        out[j + 3 * p] = make_double2(a1.x - a3.x, a1.y - a3.y);
        cur ^= 1;
        __syncthreads();
    }
    const double2 *Z = buf[cur];
    for (int f = tid; f < ((LSM_MEL_ABLATE & 4) ? 0 : NBINS); f += 256) {
        const double2 zk = Z[f & (N2 - 1)];
        const double2 zr = Z[(N2 - f) & (N2 - 1)];
        const double2 E = make_double2(0.5 * (zk.x + zr.x), 0.5 * (zk.y - zr.y));    // (Zk + conj Zr)/2
        const double2 O = make_double2(0.5 * (zk.x - zr.x), 0.5 * (zk.y + zr.y));    // (Zk - conj Zr)/2
        const double2 D = cmul(tw2048(twiddle, f), O);
        const float re = (float)(E.x + D.y), im = (float)(E.y - D.x);               // complex64 storage
        const float mag = hypotf(re, im);                       // np.abs on complex64
        pw[f] = mag * mag;                                      // ** 2.0 in float32
    }
    __syncthreads();
    // mel projection: four lanes per filter, lane q takes bins lo+q, lo+q+4, ... (ascending), the four
    // partial sums are combined as (p0 + p1) + (p2 + p3)
    for (int m0 = 0; m0 < n_mels; m0 += 64) {
        const int m = m0 + (tid >> 2), q = tid & 3;
        float acc = 0.0f;
        if (m < n_mels) {
            const float *row = basis + (size_t)m * NBINS;
            const int h = hi[m];
            for (int f = lo[m] + q; f < ((LSM_MEL_ABLATE & 2) ? lo[m] + 1 : h); f += 4) acc += row[f] * pw[f];
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (m < n_mels && q == 0) power_out[((size_t)b * n_mels + m) * n_frames + t] = acc;
    }
}

__global__ __launch_bounds__(256) void mel_power_kernel(const MelArgs a)
{
    __shared__ double2 buf[2][N2];              // 2 x 16 KB
    __shared__ float pw[NBINS + 3];
    mel_power_body(a, buf, pw, (int)blockIdx.x);
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

__global__ __launch_bounds__(256) void mel_spikes_kernel(const MelSpikeArgs a)
{
    __shared__ double2 buf[2][N2];              // 2 x 16 KB; the finishing workgroup reuses it as the raster stage
    __shared__ float pw[NBINS + 3];
    __shared__ int last;
    const int b = blockIdx.y, tid = threadIdx.x;
    // A workgroup takes `frames_per_wg` consecutive frames, one after the other: the release fence below writes the
    // XCD's dirty L2 lines back (the eight L2s are not coherent with each other), which costs about a microsecond, so
    // it is paid once per group of frames, not once per frame (one frame per workgroup: 2.4 instead of 0.3 ms per
    // 200 clips, profiles/r04_mel_one_launch_ab.txt).
    const int t0 = (int)blockIdx.x * a.frames_per_wg;
    const int t1 = min(t0 + a.frames_per_wg, a.mel.n_frames);
    for (int t = t0; t < t1; ++t) {
        if (t > t0) __syncthreads();            // the previous frame's LDS buffers are free again
        mel_power_body(a.mel, buf, pw, t);
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
    power_to_db_body(a.mel.power_out + (size_t)b * n, db, n, a.amin, a.top_db, pw);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // dB values written by other threads of this workgroup
    __syncthreads();
    lsm_fe::spec_to_spikes_body<float>(a.sp, b, reinterpret_cast<unsigned char *>(buf));
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
    MelArgs a;
    a.audio = audio; a.n_samples = n_samples; a.hop = hop; a.n_frames = n_frames; a.n_mels = n_mels;
    a.window = window_dev; a.twiddle = reinterpret_cast<const double2 *>(twiddle_dev); a.basis = basis_dev;
    a.lo = lo_dev; a.hi = hi_dev; a.power_out = power_out;
    hipLaunchKernelGGL(mel_power_kernel, dim3(n_frames, n_clips), dim3(256), 0, (hipStream_t)stream, a);
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
    LSM_REQUIRE(workspace_bytes >= lsm_mel_spikes_workspace(n_clips, n_mels, n_frames),
                "workspace of %ld bytes, need %ld (lsm_mel_spikes_workspace)", workspace_bytes,
                lsm_mel_spikes_workspace(n_clips, n_mels, n_frames));
    LSM_REQUIRE(((uintptr_t)workspace & 255u) == 0, "workspace must be 256-byte aligned");
    const int row_bytes = time_bins * n_thr;
    LSM_REQUIRE((row_bytes & 3) != 0 || ((uintptr_t)raster & 3u) == 0,
                "the raster must be 4-byte aligned when a row is a multiple of 4 bytes");
    // the finishing workgroup stages the clip's raster bit-packed in the FFT's two LDS buffers (32 KB)
    const size_t stage = 64 + (size_t)n_mels * (((size_t)row_bytes + 31) / 32) * 4;
    if (stage > sizeof(double2) * 2 * N2) {
        lsm_set_error("mel_spikes: a raster stage of %zu bytes exceeds the kernel's 32 KB; use the split entry points", stage);
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
    hipLaunchKernelGGL(mel_spikes_kernel, dim3((n_frames + fpw - 1) / fpw, n_clips), dim3(256), 0, (hipStream_t)stream, a);
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
