// Mel branch of the front end for gfx950 (SPEC.md §1.5).
//
// Replaces librosa.feature.melspectrogram(y, sr=16000, n_mels=F, hop_length=hop) and
// librosa.power_to_db(S, ref=np.max) as called at /root/reference/create_dataset.py:44-48 (librosa
// 0.11 defaults: n_fft = win_length = 2048, periodic Hann, centred frames with zero padding,
// power 2, Slaney mel basis; amin 1e-10, top_db 80).  librosa multiplies the float64 window into the
// float32 frame, transforms in float64 and stores complex64; the kernel follows the same dtypes.
// One workgroup = one frame: 2048-point radix-2 FFT in LDS (float64), |.|^2 in float32, then the
// triangular mel filters (each thread one filter, sequential over its non-zero bins).
#include "lsm_common.h"

namespace {

constexpr int NFFT = 2048;
constexpr int LOG2N = 11;
constexpr int NBINS = NFFT / 2 + 1;

__global__ __launch_bounds__(256) void mel_power_kernel(
    const float *__restrict__ audio, int n_samples, int hop, int n_frames,
    const double *__restrict__ window, const double2 *__restrict__ twiddle,   // W^k, k < NFFT/2
    const float *__restrict__ basis, const int *__restrict__ lo, const int *__restrict__ hi,
    int n_mels, float *__restrict__ power_out)
{
    __shared__ double2 x[NFFT];                 // 32 KB
    __shared__ float pw[NBINS + 3];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *clip = audio + (size_t)b * n_samples;
    const int start = t * hop - NFFT / 2;       // centred frame, zero padding outside the clip

    for (int n = tid; n < NFFT; n += 256) {
        const int src = start + n;
        const double v = (src >= 0 && src < n_samples) ? (double)clip[src] : 0.0;
        const int r = (int)(__brev((unsigned)n) >> (32 - LOG2N));
        x[r] = make_double2(window[n] * v, 0.0);
    }
    __syncthreads();
    for (int s = 1; s <= LOG2N; ++s) {
        const int half = 1 << (s - 1);
        const int tstep = (NFFT / 2) >> (s - 1);
        for (int q = tid; q < NFFT / 2; q += 256) {
            const int k = q & (half - 1);
            const int i0 = ((q >> (s - 1)) << s) + k;
            const int i1 = i0 + half;
            const double2 w = twiddle[k * tstep];
            const double2 a = x[i0], c = x[i1];
            const double tr = c.x * w.x - c.y * w.y;
            const double ti = c.x * w.y + c.y * w.x;
            x[i0] = make_double2(a.x + tr, a.y + ti);
            x[i1] = make_double2(a.x - tr, a.y - ti);
        }
        __syncthreads();
    }
    for (int f = tid; f < NBINS; f += 256) {
        const float re = (float)x[f].x, im = (float)x[f].y;     // complex64 storage
        const float mag = hypotf(re, im);                       // np.abs on complex64
        pw[f] = mag * mag;                                      // ** 2.0 in float32
    }
    __syncthreads();
    for (int m = tid; m < n_mels; m += 256) {
        const float *row = basis + (size_t)m * NBINS;
        float acc = 0.0f;
        for (int f = lo[m]; f < hi[m]; ++f) acc += row[f] * pw[f];
        power_out[((size_t)b * n_mels + m) * n_frames + t] = acc;
    }
}

// librosa.power_to_db(S, ref=np.max): per clip, float32.
__global__ __launch_bounds__(256) void power_to_db_kernel(const float *__restrict__ power, int n,
                                                          float amin, float top_db,
                                                          float *__restrict__ db_out)
{
    __shared__ float red[4];
    const float *p = power + (size_t)blockIdx.x * n;
    float *o = db_out + (size_t)blockIdx.x * n;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, p[i]);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float refdb = 10.0f * log10f(fmaxf(amin, mx));
    // the maximum of (10*log10(max(amin, S)) - refdb) is exactly 0, so the floor is -top_db
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = 10.0f * log10f(fmaxf(amin, p[i])) - refdb;
        o[i] = fmaxf(v, 0.0f - top_db);
    }
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
    hipLaunchKernelGGL(mel_power_kernel, dim3(n_frames, n_clips), dim3(256), 0, (hipStream_t)stream,
                       audio, n_samples, hop, n_frames, window_dev,
                       reinterpret_cast<const double2 *>(twiddle_dev), basis_dev, lo_dev, hi_dev,
                       n_mels, power_out);
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
