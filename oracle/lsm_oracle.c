/*
 * CPU oracle, plain C — TEST INFRASTRUCTURE ONLY (the checker and the timed CPU baseline).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path never links or calls it.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so every float
 * operation below rounds exactly as written, like the NumPy code it restates).
 *
 * Pinning:
 *   PINNED by the reference's in-tree code through tests/golden fixtures:
 *     orc_normalise_resize_*  (/root/reference/create_dataset.py:62-78)
 *     orc_encode_hysteresis_* (/root/reference/create_dataset.py:81-98)
 *   PARITY UNPINNED (third-party arithmetic absent from /root/reference and from this image):
 *     orc_gammatone_spec  — gammatone==1.0.3 gtgram, called at create_dataset.py:51-58
 *     orc_mel_*           — librosa==0.11.0, called at create_dataset.py:45-48
 *     orc_lif_*           — snn_reservoir_py==2.0.0, called at extract_lsm_features.py:79-83;
 *                           follows this build's SPEC.md §3-§4.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_version(void) { return 1; }

/* ---------------------------------------------------------------------------------------------
 * Gammatone spectrogram (gammatone.gtgram.gtgram): per channel four cascaded second-order
 * sections evaluated exactly like scipy.signal.lfilter's direct-form-II-transposed loop
 *     y = z0 + b0*x;  z0 = (z1 + x*b1) - y*a1;  z1 = x*b2 - y*a2
 * in float64, output divided by the channel gain, squared, then for every column the SEQUENTIAL
 * ascending sum of nwin samples (NumPy reduces the F-ordered fancy-index result that way for
 * F >= 2; for F == 1 it sums pairwise, restated below: the product refuses one gammatone filter), / nwin, sqrt.
 * coefs rows: [A0, A11, A12, A13, A14, A2, B0, B1, B2, gain], B0 == 1.
 * -------------------------------------------------------------------------------------------*/
/* NumPy's pairwise summation of a contiguous float64 run (numpy/core/src/umath/loops_utils.h, pairwise_sum_DOUBLE):
 * below 8 elements a plain loop, up to 128 eight interleaved partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
 * plus the tail, above that two halves, the first a multiple of 8 long. */
static double np_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

ORC_API int orc_gammatone_spec(const float *audio, int n_samples, const double *coefs,
                               int n_filters, int nwin, int hop, int ncols, double *out)
{
    if (ncols < 1 || (ncols - 1) * hop + nwin > n_samples) return -1;
    double *xe = (double *)malloc(sizeof(double) * (size_t)n_samples);
    if (!xe) return -2;
    for (int ch = 0; ch < n_filters; ++ch) {
        const double *k = coefs + (size_t)ch * 10;
        const double b0 = k[0] / k[6], b2 = k[5] / k[6];
        const double b1[4] = { k[1] / k[6], k[2] / k[6], k[3] / k[6], k[4] / k[6] };
        const double a1 = k[7] / k[6], a2 = k[8] / k[6], gain = k[9];
        double z0[4] = {0, 0, 0, 0}, z1[4] = {0, 0, 0, 0};
        for (int n = 0; n < n_samples; ++n) {
            double x = (double)audio[n];
            for (int s = 0; s < 4; ++s) {
                double y = z0[s] + b0 * x;
                z0[s] = (z1[s] + x * b1[s]) - y * a1;
                z1[s] = x * b2 - y * a2;
                x = y;
            }
            double v = x / gain;
            xe[n] = v * v;
        }
        for (int c = 0; c < ncols; ++c) {
            const double *seg = xe + (size_t)c * hop;
            double acc = seg[0];
            /* one filter: the fancy-index result (1, nwin) is contiguous along the reduced axis and NumPy sums it
             * PAIRWISE; two and more filters: element by element, starting from the first */
            if (n_filters == 1) acc = np_pairwise_sum(seg, nwin);
            else for (int i = 1; i < nwin; ++i) acc += seg[i];
            out[(size_t)ch * ncols + c] = sqrt(acc / (double)nwin);
        }
    }
    free(xe);
    return 0;
}

/* create_dataset.py:59-60: 20*log10(x + 1e-9), floored at (global max - 80). In place.
 * spec_db.max() and np.maximum propagate NaN: one NaN value makes the maximum, the floor and with it every element NaN
 * (tests/golden/postfilter_nonfinite.npz holds the reference's output for such inputs). */
ORC_API void orc_gammatone_db(double *spec, int n)
{
    double mx = -INFINITY;
    int nan_seen = 0;
    for (int i = 0; i < n; ++i) {
        spec[i] = 20 * log10(spec[i] + 1e-9);
        if (spec[i] > mx) mx = spec[i];
        nan_seen |= spec[i] != spec[i];
    }
    const double fl = nan_seen ? (double)NAN : mx - 80.0;
    for (int i = 0; i < n; ++i) if (!(spec[i] >= fl)) spec[i] = fl;
}

/* create_dataset.py:62-78 on float64 input (gammatone branch). out is (n_filters, time_bins).
 * Returns 1 when the input is flat (all-zero output, as the reference returns zeros). */
ORC_API int orc_normalise_resize_f64(const double *db, int n_filters, int ncols, int time_bins,
                                     double *out)
{
    double lo = INFINITY, hi = -INFINITY;
    const int n = n_filters * ncols;
    int nan_seen = 0;
    for (int i = 0; i < n; ++i) { if (db[i] < lo) lo = db[i]; if (db[i] > hi) hi = db[i]; nan_seen |= db[i] != db[i]; }
    if (nan_seen) lo = hi = (double)NAN;      /* ndarray.min() / .max() propagate NaN (create_dataset.py:62-63) */
    if ((hi - lo) < 1e-8) {
        memset(out, 0, sizeof(double) * (size_t)n_filters * time_bins);
        return 1;
    }
    const double den = hi - lo + 1e-8;
    if (ncols == time_bins) {
        for (int i = 0; i < n; ++i) out[i] = (db[i] - lo) / den;
        return 0;
    }
    /* scipy.ndimage.zoom(order=1) along time; output length round(ncols*(time_bins/ncols)) is
     * time_bins for every ncols used here, then [:, :time_bins]. */
    const double zf = (double)(ncols - 1) / (double)(time_bins - 1);
    for (int j = 0; j < time_bins; ++j) {
        const double cc = (double)j * zf;
        const double fl = floor(cc);
        const int f = (int)fl;
        const double w0 = 1.0 - (cc - fl);
        const double w1 = 1.0 - w0;
        for (int r = 0; r < n_filters; ++r) {
            const double *row = db + (size_t)r * ncols;
            const double a = (row[f] - lo) / den;
            double v = a * w0;
            if (f + 1 <= ncols - 1) v = v + ((row[f + 1] - lo) / den) * w1;
            out[(size_t)r * time_bins + j] = v;
        }
    }
    return 0;
}

/* Same on float32 input (mel branch): NumPy keeps float32 for (x-lo)/(hi-lo+1e-8) — the
 * Python-float 1e-8 is a weak scalar — while SciPy's zoom interpolates in double and casts
 * the result back to float32. */
ORC_API int orc_normalise_resize_f32(const float *db, int n_filters, int ncols, int time_bins,
                                     float *out)
{
    float lo = INFINITY, hi = -INFINITY;
    const int n = n_filters * ncols;
    int nan_seen = 0;
    for (int i = 0; i < n; ++i) { if (db[i] < lo) lo = db[i]; if (db[i] > hi) hi = db[i]; nan_seen |= db[i] != db[i]; }
    if (nan_seen) lo = hi = NAN;              /* ndarray.min() / .max() propagate NaN (create_dataset.py:62-63) */
    if ((hi - lo) < 1e-8f) {
        memset(out, 0, sizeof(float) * (size_t)n_filters * time_bins);
        return 1;
    }
    const float den = (hi - lo) + 1e-8f;
    if (ncols == time_bins) {
        for (int i = 0; i < n; ++i) out[i] = (db[i] - lo) / den;
        return 0;
    }
    const double zf = (double)(ncols - 1) / (double)(time_bins - 1);
    for (int j = 0; j < time_bins; ++j) {
        const double cc = (double)j * zf;
        const double fl = floor(cc);
        const int f = (int)fl;
        const double w0 = 1.0 - (cc - fl);
        const double w1 = 1.0 - w0;
        for (int r = 0; r < n_filters; ++r) {
            const float *row = db + (size_t)r * ncols;
            const float a = (row[f] - lo) / den;
            double v = (double)a * w0;
            if (f + 1 <= ncols - 1) v = v + (double)((row[f + 1] - lo) / den) * w1;
            out[(size_t)r * time_bins + j] = (float)v;
        }
    }
    return 0;
}

/* create_dataset.py:81-98. on[]/off[] are the thresholds sorted descending and thr-gap, already
 * rounded to the spectrogram dtype by the caller. out is (n_filters, n_bins*n_thr) uint8. */
ORC_API void orc_encode_hysteresis_f64(const double *spec, int n_filters, int n_bins,
                                       const double *on, const double *off, int n_thr,
                                       uint8_t *out)
{
    for (int r = 0; r < n_filters; ++r)
        for (int k = 0; k < n_thr; ++k) {
            int active = 0;
            for (int b = 0; b < n_bins; ++b) {
                const double x = spec[(size_t)r * n_bins + b];
                const int rising = (x > on[k]) && !active;
                const int falling = (x < off[k]) && active;
                if (rising) active = 1;
                if (falling) active = 0;
                out[(size_t)r * n_bins * n_thr + (size_t)b * n_thr + k] = (uint8_t)active;
            }
        }
}

ORC_API void orc_encode_hysteresis_f32(const float *spec, int n_filters, int n_bins,
                                       const float *on, const float *off, int n_thr,
                                       uint8_t *out)
{
    for (int r = 0; r < n_filters; ++r)
        for (int k = 0; k < n_thr; ++k) {
            int active = 0;
            for (int b = 0; b < n_bins; ++b) {
                const float x = spec[(size_t)r * n_bins + b];
                const int rising = (x > on[k]) && !active;
                const int falling = (x < off[k]) && active;
                if (rising) active = 1;
                if (falling) active = 0;
                out[(size_t)r * n_bins * n_thr + (size_t)b * n_thr + k] = (uint8_t)active;
            }
        }
}

/* The gammatone front end of a whole batch, one clip per OpenMP thread: spectrogram -> dB + floor ->
 * normalise/resize -> hysteresis encoder, exactly the per-clip calls above in the order
 * /root/reference/create_dataset.py:148-157 makes them.  Used by bench.py's cpu_baseline to MEASURE the
 * all-cores front-end rate (and by a test that checks it against the per-clip calls).
 * rasters: (n_clips, n_filters, time_bins*n_thr) uint8. */
ORC_API int orc_gammatone_frontend_batch(const float *audio, int n_clips, int n_samples, const double *coefs,
                                         int n_filters, int nwin, int hop, int ncols, int time_bins,
                                         const double *thr_on, const double *thr_off, int n_thr,
                                         int n_threads, uint8_t *rasters)
{
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (int b = 0; b < n_clips; ++b) {
        double *spec = (double *)malloc(sizeof(double) * (size_t)n_filters * (size_t)ncols);
        double *norm = (double *)malloc(sizeof(double) * (size_t)n_filters * (size_t)time_bins);
        if (!spec || !norm ||
            orc_gammatone_spec(audio + (size_t)b * n_samples, n_samples, coefs, n_filters, nwin, hop, ncols, spec)) {
#pragma omp atomic write
            bad = 1;
        } else {
            orc_gammatone_db(spec, n_filters * ncols);
            (void)orc_normalise_resize_f64(spec, n_filters, ncols, time_bins, norm);   /* flat input: zeros */
            orc_encode_hysteresis_f64(norm, n_filters, time_bins, thr_on, thr_off, n_thr,
                                      rasters + (size_t)b * n_filters * time_bins * n_thr);
        }
        free(spec);
        free(norm);
    }
    return bad ? -1 : 0;
}

/* ---------------------------------------------------------------------------------------------
 * LIF reservoir, SPEC.md §3 (gather form, literal) + §4 features.  float32 throughout, one
 * accumulator per neuron: presynaptic j ascending, then w_in * (active input count).
 * key ids: 0 spike_counts, 1 spike_variances, 2 mean_spike_times, 3 first_spike_times,
 *          4 last_spike_times, 5 mean_isi, 6 isi_variances, 7 burst_counts.
 * features: (n_keys, n_out) float32 with NaN already replaced by 0 (extract_lsm_features.py:85).
 * -------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t n, first, last, bursts;
    int64_t sum_t, sum_isi2;
} orc_acc_t;

ORC_API int orc_lif_run(int N, int C, int T,
                        const int32_t *csr_ptr, const int32_t *csr_pre, const float *csr_w,
                        const int32_t *in_ptr, const int32_t *in_chan, float w_in,
                        const float *leak, float theta, int refractory,
                        const uint8_t *raster,          /* (C, T) */
                        uint8_t *spike_matrix,          /* (T, N) or NULL */
                        float *v_trace,                 /* (T, N) or NULL */
                        int n_out, const int32_t *out_idx, int burst_isi_max,
                        int n_keys, const int32_t *key_ids,
                        float *features)                /* (n_keys, n_out) or NULL */
{
    float *v = (float *)calloc((size_t)N, sizeof(float));
    int32_t *ref = (int32_t *)calloc((size_t)N, sizeof(int32_t));
    uint8_t *s_prev = (uint8_t *)calloc((size_t)N, 1);
    uint8_t *s_now = (uint8_t *)calloc((size_t)N, 1);
    orc_acc_t *acc = (orc_acc_t *)calloc((size_t)N, sizeof(orc_acc_t));
    if (!v || !ref || !s_prev || !s_now || !acc) return -2;
    (void)C;

    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < N; ++i) {
            /* SPEC.md §3: one float32 accumulator over the presynaptic neurons, ascending; then the
             * input term w_in * (number of active input channels of neuron i at step t) */
            float cur = 0.0f;
            for (int e = csr_ptr[i]; e < csr_ptr[i + 1]; ++e)
                cur += csr_w[e] * (float)s_prev[csr_pre[e]];
            int m_in = 0;
            for (int e = in_ptr[i]; e < in_ptr[i + 1]; ++e)
                m_in += raster[(size_t)in_chan[e] * T + t] != 0;              /* any non-zero byte is a spike */
            cur += w_in * (float)m_in;
            int fire = 0;
            if (ref[i] > 0) {
                ref[i] -= 1;
                v[i] = 0.0f;
            } else {
                const float m = leak[i] * v[i];
                const float d = v[i] - m;
                const float vn = d + cur;
                if (vn >= theta) { fire = 1; v[i] = 0.0f; ref[i] = refractory; }
                else v[i] = vn;
            }
            s_now[i] = (uint8_t)fire;
            if (fire) {
                orc_acc_t *a = &acc[i];
                if (a->n == 0) a->first = t;
                else {
                    const int isi = t - a->last;
                    a->sum_isi2 += (int64_t)isi * isi;
                    if (isi <= burst_isi_max) a->bursts += 1;
                }
                a->last = t;
                a->n += 1;
                a->sum_t += t;
            }
        }
        if (spike_matrix) memcpy(spike_matrix + (size_t)t * N, s_now, (size_t)N);
        if (v_trace) memcpy(v_trace + (size_t)t * N, v, sizeof(float) * (size_t)N);
        uint8_t *tmp = s_prev; s_prev = s_now; s_now = tmp;
    }

    if (features) {
        for (int kq = 0; kq < n_keys; ++kq)
            for (int o = 0; o < n_out; ++o) {
                const orc_acc_t *a = &acc[out_idx[o]];
                const int n = a->n;
                double f = 0.0;
                switch (key_ids[kq]) {
                case 0: f = (double)n; break;
                case 1: { const double p = (double)n / (double)T; f = p * (1.0 - p); } break;
                case 2: f = n >= 1 ? (double)a->sum_t / (double)n : 0.0; break;
                case 3: f = n >= 1 ? (double)a->first : 0.0; break;
                case 4: f = n >= 1 ? (double)a->last : 0.0; break;
                case 5: f = n >= 2 ? (double)(a->last - a->first) / (double)(n - 1) : 0.0; break;
                case 6:
                    if (n >= 2) {
                        const double m = (double)(a->last - a->first) / (double)(n - 1);
                        f = (double)a->sum_isi2 / (double)(n - 1) - m * m;
                    }
                    break;
                case 7: f = (double)a->bursts; break;
                default: return -3;
                }
                features[(size_t)kq * n_out + o] = (float)f;
            }
    }
    free(v); free(ref); free(s_prev); free(s_now); free(acc);
    return 0;
}

/* Batch driver: one clip per OpenMP thread (n_threads <= 0: leave the runtime default). */
#ifdef _OPENMP
#include <omp.h>
#endif
ORC_API int orc_lif_run_batch(int B, int n_threads, int N, int C, int T,
                              const int32_t *csr_ptr, const int32_t *csr_pre, const float *csr_w,
                              const int32_t *in_ptr, const int32_t *in_chan, float w_in,
                              const float *leak, float theta, int refractory,
                              const uint8_t *rasters,   /* (B, C, T) */
                              int n_out, const int32_t *out_idx, int burst_isi_max,
                              int n_keys, const int32_t *key_ids,
                              float *features)          /* (B, n_keys*n_out) */
{
    int rc = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
        int r = orc_lif_run(N, C, T, csr_ptr, csr_pre, csr_w, in_ptr, in_chan, w_in, leak, theta,
                            refractory, rasters + (size_t)b * C * T, NULL, NULL, n_out, out_idx,
                            burst_isi_max, n_keys, key_ids,
                            features + (size_t)b * n_keys * n_out);
        if (r) rc = r;
    }
    return rc;
}
