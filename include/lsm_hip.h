/*
 * liblsm_hip.so — C ABI of the MI355X (gfx950) hot path of the LSM speech pipeline.
 *
 * The reference (adelitoo/lsm-speech-classifier) is pure Python and has no FFI of its own; the
 * seam this library sits behind is the set of Python call sites listed per function below
 * (file:line under /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative LSM_ERR_* code; lsm_last_error() gives
 *     the thread-local message of the last failure on the calling thread;
 *   - every pointer named *_dev / audio / db / spikes / *_out is DEVICE memory owned by the
 *     caller (e.g. torch.Tensor.data_ptr()); tables named coefs / thr_* / key_ids and every
 *     argument of lsm_reservoir_create are HOST memory, copied during the call;
 *   - every launch function takes a hipStream_t as `void *stream` and is asynchronous on it;
 *     none of them allocates, frees or synchronises, and after a kernel's first use its launch is the
 *     only runtime call made (safe under hipGraph capture: tests/test_gpu_graph.py);
 *   - the only device memory the library owns is inside an lsm_reservoir handle;
 *   - no global mutable state: distinct handles/streams may be used from distinct threads.
 */
#ifndef LSM_HIP_H
#define LSM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSM_OK 0
#define LSM_ERR_ARG (-1)
#define LSM_ERR_HIP (-2)
#define LSM_ERR_NOMEM (-3)
#define LSM_ERR_UNSUPPORTED (-4)

int lsm_version(void);                 /* major*10000 + minor*100 + patch of the package version the library was built as */
/* "LSM_BUILD_ID=<24 hex digits>": hash of every source and header of csrc/, this header, the compiler flags and the
 * version, linked in by lsm-speech-classifier_amd/build.py.  The Python loader refuses a library whose id is not the one
 * of the tree it sits in (a stale binary cannot pass for a build of the sources beside it). */
const char *lsm_build_id(void);
const char *lsm_last_error(void);
int lsm_device_count(void);

/* ---- front end ------------------------------------------------------------------------- */

/* Replaces gammatone.gtgram.gtgram(wave, fs, window_time, hop_time, channels, f_min) as called
 * at create_dataset.py:51-58, plus the dB conversion of create_dataset.py:59.
 *   audio   (n_clips, n_samples) float32, device
 *   coefs   (n_filters, 10) float64, HOST-side values uploaded by the caller to DEVICE memory:
 *           rows [A0, A11, A12, A13, A14, A2, B0, B1, B2, gain], low -> high centre frequency
 *   spec_out (n_clips, n_filters, ncols) float64 or NULL: sqrt(mean of squared filter output)
 *   db_out   (n_clips, n_filters, ncols) float64 or NULL: 20*log10(spec + 1e-9)
 * nwin/hop are in samples (400/160 for the reference call), nwin <= 4*hop.
 * coef_flags (properties of the coefficient table the HOST has verified; 0 is always valid):
 *   bit 0  every A2 is exactly 0  -> the x*A2 products (signed zeros) are not evaluated
 *   bit 1  no gain has an all-ones significand -> y/gain through the exact Markstein FMA sequence
 *   bit 2  A0/B0 is the same float64 for every channel (lsm_gammatone_spikes_f64 only: the product with the sample
 *          is then formed once for the two channels a lane carries) */
int lsm_gammatone_spec_f64(const float *audio, int n_clips, int n_samples, const double *coefs_dev,
                           int n_filters, int nwin, int hop, int ncols, double *spec_out,
                           double *db_out, int coef_flags, void *stream);

/* The whole gammatone front end of one clip in ONE launch: replaces the per-clip body of the loop at
 * create_dataset.py:143-157 -- audio_to_spectrogram (gammatone branch, :49-78: gtgram, dB, max-80 floor,
 * min-max normalise with eps 1e-8, flat input -> zeros, scipy zoom(order=1) to time_bins columns, crop),
 * convert_spectrogram_to_spikes_hysteresis (:81-98) and create_pure_redundancy (:101-104).  Same results,
 * bit for bit, as lsm_gammatone_spec_f64 followed by lsm_spec_to_spikes_f64(apply_floor = 1), without the
 * (n_clips, n_filters, ncols) float64 hand-off between two launches.
 *   audio, coefs_dev, nwin/hop/ncols, coef_flags   as lsm_gammatone_spec_f64
 *   thr_on/thr_off/n_thr/redundancy/raster         as lsm_spec_to_spikes_f64 (raster must not be NULL)
 *   workspace   DEVICE scratch of at least lsm_gammatone_spikes_workspace(n_clips, n_filters, ncols) bytes,
 *               8-byte aligned, owned by the caller, contents undefined before and after the call
 *   launch_flags bit 0: low-latency layout -- one channel group per wave instead of two: twice the waves, each half
 *               as long (for a launch onto an idle GPU: 1.47 instead of 2.43 ms per 256 clips x 128 filters, 12 % more
 *               CU time); above 512 filters the flag is ignored (the one-chain layout would need more than 8 waves
 *               per clip).  bit 1: no LDS reservation -- a launch of at most one workgroup per CU normally reserves
 *               half a CU's LDS so that overlapping launches spread one workgroup per CU; beside a kernel whose
 *               workgroups need most of the LDS themselves (the ring-row reservoir kernel) the caller sets this bit and
 *               the launch asks for what it uses (27 KB per 4-wave workgroup at 4 thresholds x 100 bins).
 *               Results do not depend on the flags.  0 = the throughput layout with the reservation.
 * Returns LSM_ERR_UNSUPPORTED for more than 1024 filters (use the two split entry points). */
long lsm_gammatone_spikes_workspace(int n_clips, int n_filters, int ncols);
int lsm_gammatone_spikes_f64(const float *audio, int n_clips, int n_samples, const double *coefs_dev,
                             int n_filters, int nwin, int hop, int ncols, int time_bins,
                             const double *thr_on, const double *thr_off, int n_thr, int redundancy,
                             uint8_t *raster, void *workspace, long workspace_bytes, int coef_flags,
                             int launch_flags, void *stream);

/* Replaces create_dataset.py:60 (apply_floor: max-80 dB floor), :62-78 (min-max normalise with
 * eps 1e-8, flat input -> zeros, scipy zoom(order=1) to time_bins columns, crop), :81-98
 * (hysteresis encoder) and :101-104 (row repeat), one clip per workgroup.
 *   db      (n_clips, n_filters, ncols) device
 *   thr_on  n_thr ON thresholds sorted DESCENDING, thr_off the matching `thr - gap` values, both
 *           already rounded to the spectrogram dtype by the caller (HOST memory, <= 8 entries)
 *   raster  (n_clips, n_filters*redundancy, time_bins*n_thr) uint8 or NULL
 *   norm_out (n_clips, n_filters, time_bins) or NULL: the normalised, resized spectrogram */
int lsm_spec_to_spikes_f64(const double *db, int n_clips, int n_filters, int ncols, int time_bins,
                           int apply_floor, const double *thr_on, const double *thr_off, int n_thr,
                           int redundancy, uint8_t *raster, double *norm_out, void *stream);
int lsm_spec_to_spikes_f32(const float *db, int n_clips, int n_filters, int ncols, int time_bins,
                           int apply_floor, const float *thr_on, const float *thr_off, int n_thr,
                           int redundancy, uint8_t *raster, float *norm_out, void *stream);

/* Replaces librosa.feature.melspectrogram(y, sr=16000, n_mels, hop_length) as called at
 * create_dataset.py:44-47 (n_fft must be 2048, librosa's default): centred zero-padded frames,
 * float64 window x float32 frame, float64 FFT stored as complex64, |.|^2 in float32, mel basis.
 *   window_dev (n_fft) f64, twiddle_dev (n_fft/2, 2) f64 {cos, -sin}, basis_dev (n_mels, n_fft/2+1)
 *   f32, lo_dev/hi_dev (n_mels) i32 non-zero bin range of each filter: DEVICE tables built by the host;
 *   window_dev and twiddle_dev 16-byte aligned (the kernel fetches them two doubles at a time)
 *   power_out (n_clips, n_mels, n_frames) float32, n_frames = 1 + n_samples / hop */
int lsm_mel_power_f32(const float *audio, int n_clips, int n_samples, int n_fft, int hop,
                      int n_frames, const double *window_dev, const double *twiddle_dev,
                      const float *basis_dev, const int32_t *lo_dev, const int32_t *hi_dev,
                      int n_mels, float *power_out, void *stream);

/* The whole mel front end of a batch in ONE launch: melspectrogram (as lsm_mel_power_f32) -> power_to_db(ref=max) ->
 * min-max normalise -> resize to time_bins columns -> hysteresis encoder -> row repeat (create_dataset.py:43-48 and
 * :62-104 per clip).  Rasters equal those of lsm_mel_power_f32 + lsm_power_to_db_f32 + lsm_spec_to_spikes_f32 bit for
 * bit (same code).  The workgroup that finishes a clip's last frame finishes the clip; nothing waits.
 *   workspace        caller-owned, 256-byte aligned, >= lsm_mel_spikes_workspace(n_clips, n_mels, n_frames) bytes;
 *                    its first n_clips*4 bytes (one counter per clip) must be ZERO on entry and are zero again when
 *                    the launch has finished (allocate once with zeros, reuse for ever -- stream-ordered launches only)
 *   raster           (n_clips, n_mels*redundancy, time_bins*n_thr) uint8
 * LSM_ERR_UNSUPPORTED when the latch bit rows of the finishing workgroup (64 rows x n_thr x ceil(time_bins / 32) words, twice)
 * exceed the kernel's 68 KB of LDS (more than ~1000 time bins at 4 thresholds; any number of filters fits): use the three
 * split entry points. */
long lsm_mel_spikes_workspace(int n_clips, int n_mels, int n_frames);
int lsm_mel_spikes_f32(const float *audio, int n_clips, int n_samples, int n_fft, int hop, int n_frames,
                       const double *window_dev, const double *twiddle_dev, const float *basis_dev,
                       const int32_t *lo_dev, const int32_t *hi_dev, int n_mels, float amin, float top_db,
                       int time_bins, const float *thr_on, const float *thr_off, int n_thr, int redundancy,
                       uint8_t *raster, void *workspace, long workspace_bytes, void *stream);

/* Replaces librosa.power_to_db(S, ref=np.max) (create_dataset.py:48) per clip, float32:
 * 10*log10(max(amin, S)) - 10*log10(max(amin, max S)), floored at -top_db. */
int lsm_power_to_db_f32(const float *power, int n_clips, int n_per_clip, float amin, float top_db,
                        float *db_out, void *stream);

/* Replaces convert_spectrogram_to_spikes_hysteresis (create_dataset.py:81-98) on an already
 * normalised spectrogram: spec (n_rows, n_bins) -> out (n_rows, n_bins*n_thr) uint8. */
int lsm_encode_hysteresis_f64(const double *spec, int n_rows, int n_bins, const double *thr_on,
                              const double *thr_off, int n_thr, uint8_t *out, void *stream);
int lsm_encode_hysteresis_f32(const float *spec, int n_rows, int n_bins, const float *thr_on,
                              const float *thr_off, int n_thr, uint8_t *out, void *stream);

/* Bit-packed rasters: the packed variant of File 1 (same content as the X_spikes
 * array written by /root/reference/create_dataset.py:168-176, 8x smaller on disk and over PCIe).
 * raster (n_rows, n_steps) uint8, any non-zero byte is a spike; packed (n_rows, ceil(n_steps/8)) uint8,
 * time step 8q+k of a row in bit k of its byte q (numpy.packbits(..., bitorder="little")), unused
 * high bits of the last byte zero.  Unpacking writes 0/1 bytes.  The raster must be 8-byte aligned when
 * n_steps is a multiple of 8 (it then moves as 64-bit words). */
int lsm_raster_pack_bits(const uint8_t *raster, long n_rows, int n_steps, uint8_t *packed, void *stream);
int lsm_raster_unpack_bits(const uint8_t *packed, long n_rows, int n_steps, uint8_t *raster, void *stream);

/* ---- reservoir ------------------------------------------------------------------------- */

typedef struct lsm_reservoir lsm_reservoir;

/* Replaces SNN(simulation_params=...) (extract_lsm_features.py:188): uploads one reservoir's
 * wiring (built on the host per SPEC.md §2) to the current device.  All arrays are HOST memory.
 *   csc_ptr (N+1), csc_post (nnz, ascending within a column), csc_w (nnz): synapses grouped by
 *           PRESYNAPTIC neuron j;  leak (N);  in_tgt (n_channels, in_fanout) target neurons of
 *           each input channel;  out_idx (n_out) strictly ascending output neurons. */
int lsm_reservoir_create(lsm_reservoir **out, int num_neurons, int n_channels,
                         const int32_t *csc_ptr, const int32_t *csc_post, const float *csc_w,
                         const float *leak, const int32_t *in_tgt, int in_fanout, float w_in,
                         const int32_t *out_idx, int n_out, float theta, int refractory,
                         int burst_isi_max);
int lsm_reservoir_destroy(lsm_reservoir *h);

/* Kernel used by lsm_reservoir_run for this handle: 0 = choose (register accumulation over dense
 * presynaptic rows; over ring rows -- dense ring window + list of the synapses outside it -- for ring-like
 * reservoirs whose dense table exceeds the L2 caches), 1 = sparse CSC scatter through LDS, 2 = dense rows
 * (for a reservoir that mode 0 serves with ring rows the dense table is built by THIS call: it allocates and
 * synchronises, once),
 * 3 = ring rows (refused when the reservoir is not ring-like or has fewer than ~700 neurons), 4 = ring rows
 * restricted to the layouts with contiguous quad ownership (tests; 3 prefers the strided ones).  All produce
 * bit-identical results (SPEC.md §3).  num_neurons <= 8192. */
int lsm_reservoir_set_kernel(lsm_reservoir *h, int mode);

/* The kernel lsm_reservoir_run would launch for this handle now: 1 sparse, 2 dense rows, 3 ring rows -- for the
 * reference's 400 time steps and waves_per_clip = 0; lsm_reservoir_plan answers for any launch. */
int lsm_reservoir_kernel_in_use(const lsm_reservoir *h);

/* Replaces, for a whole batch, the per-clip loop body of extract_all_features
 * (extract_lsm_features.py:78-87): reset -> set_input_spike_times -> simulate ->
 * extract_features_from_spikes -> nan_to_num -> concatenate in key order.
 *   spikes_u8   (n_clips, n_channels, n_steps) uint8, device (File-1 layout, create_dataset.py:168)
 *   key_ids     n_keys entries (HOST) from {0 spike_counts, 1 spike_variances, 2 mean_spike_times,
 *               3 first_spike_times, 4 last_spike_times, 5 mean_isi, 6 isi_variances,
 *               7 burst_counts} (FEATURE_SETS, extract_lsm_features.py:19-28)
 *   features_out     (n_clips, n_keys*n_out) float32, device
 *   spike_matrix_out (n_clips, n_steps, N) uint8 or NULL   (lsm.spike_matrix, :113-116)
 *   v_trace_out      (n_clips, n_steps, N) float32 or NULL (membrane potential after each step)
 *   stats_out        (n_clips, 2) int32 or NULL: per clip {neurons that fired at least once, spikes of the
 *                    whole reservoir} -- what run_network_diagnostics (:119-133) derives from lsm.spike_matrix
 *                    (participation, dead neurons, mean spikes per neuron), accumulated inside the kernel
 *   waves_per_clip   0 = choose for a lone launch of this batch size; -1 = choose for a launch that shares
 *                    the GPU with other kernels of an overlapped pipeline; else 1, 2, 4, 8 or 16
 * Fails (LSM_ERR, "no ... layout") when no layout's per-clip LDS image fits a CU's 160 KB: roughly
 * 10*N + 16*n_out + n_steps*ceil(n_channels/32)*4 bytes for ring rows (e.g. N = 8000 with more than ~4500 output
 * neurons); every BASELINE configuration fits (N = 8000, n_out = 3200, 256 channels: 145 KB). */
int lsm_reservoir_run(const lsm_reservoir *h, const uint8_t *spikes_u8, int n_clips, int n_steps,
                      const int32_t *key_ids, int n_keys, float *features_out,
                      uint8_t *spike_matrix_out, float *v_trace_out, int32_t *stats_out,
                      int waves_per_clip, void *stream);

/* lsm_reservoir_run with the clips of the batch STARTED longest first.  A clip's time in the kernel grows with its
 * activity, which follows its input spike count (rank correlation 0.99 at N = 4000; clips of one batch differ ninefold);
 * a launch with more clips than the chip holds at once is otherwise as long as whichever clip starts last.  Two small
 * launches on the same stream count every clip's input spikes and rank them into `workspace` (caller-owned device memory,
 * lsm_reservoir_order_workspace(n_clips) bytes, private to this call until the stream has passed it); workgroup g of the
 * LIF kernel then takes the clip with the g-th most input spikes.  Every output is the one lsm_reservoir_run writes, at the
 * same place (row b belongs to clip b).  Batches of at most one clip per compute unit are launched as they are. */
long lsm_reservoir_order_workspace(int n_clips);
int lsm_reservoir_run_ordered(const lsm_reservoir *h, const uint8_t *spikes_u8, int n_clips, int n_steps,
                              const int32_t *key_ids, int n_keys, float *features_out,
                              uint8_t *spike_matrix_out, float *v_trace_out, int32_t *stats_out,
                              int waves_per_clip, void *workspace, long workspace_bytes, void *stream);

/* Layout that lsm_reservoir_run would use: waves per clip, 64-neuron slots per lane, LDS bytes. */
int lsm_reservoir_layout(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip,
                         int *wpc_out, int *slots_out, int *lds_bytes_out);

/* The whole decision lsm_reservoir_run makes for (handle, batch, steps, waves_per_clip) -- run, layout and
 * kernel_in_use all go through it: kernel (1 sparse, 2 dense rows, 3 ring rows), waves per clip, slots per lane,
 * LDS bytes per clip, and the bytes of the weight table that kernel gathers its rows from (dense rows: N x ld x 4;
 * ring rows: windows + this layout's lists; sparse: the CSC arrays).  Any out pointer may be NULL.  In auto mode a
 * reservoir that prefers ring rows falls back to the dense (else sparse) kernel when no ring layout fits the LDS or
 * offers the requested waves per clip; an explicit ring request (modes 3, 4) fails instead. */
int lsm_reservoir_plan(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip, int *kernel_out,
                       int *wpc_out, int *slots_out, int *lds_bytes_out, long *table_bytes_out);

/* Bytes ONE reservoir spike makes the planned kernel request from its weight table, mean over the presynaptic
 * neurons (what bench.py's `roofline.row_gather` charges per spike; padding of the table is not charged): dense rows
 * ld x 4; ring rows the window's existing bytes (128-byte-aligned start .. window end, 16-byte granules) + 8 bytes per
 * list entry + the (waves + 1) row pointers; sparse 8 bytes per synapse + the two row pointers. */
int lsm_reservoir_row_request_bytes(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip,
                                    double *mean_bytes_out);

/* How the planned kernel forms the input drive (introspection for tests): dense rows 0 = input-map entries streamed
 * from global memory + LDS atomics, 1 = entries in registers + LDS atomics, 2 = per-neuron channel masks, four
 * popcounts, 3 = channel masks at coloured bit positions, one popcount (C <= 128 and an assignment exists in which
 * the channels feeding one neuron differ mod 32); ring rows 10 = packed entries streamed, 11 = packed entries in
 * registers, 12 / 13 = per-neuron channel masks at natural / coloured positions (C <= 128, every neuron the same leak
 * coefficient, strided quad ownership with at most two quads per wave), 14 / 15 = the same masks in the pair-block form of the
 * ring rows (C <= 128; any leak coefficients); sparse kernel 20. */
int lsm_reservoir_input_mode(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip);

/* Host-only (no GPU, no HIP call): the ring-window table, the pair-block lists ({LDS byte offset of the target's accumulator,
 * weight bits} per entry) and the 16-byte row records lsm_reservoir_create would build for these CSC arrays and `wpc` (4, 8, 16)
 * waves per clip, with the two tables placed at the given (fictitious) addresses.  Call with null arrays for the sizes
 * (band_floats, n_list_entries, pitch_bytes), then with band_out[band_floats], rem_out[2 * n_list_entries],
 * rec_out[4 * num_neurons * wpc].  Returns the blocks per wave, 0 when the reservoir has no pair layout with that many waves,
 * < 0 on a bad argument.  tests/test_pair_layout.py applies every row from these tables alone, on the CPU. */
int lsm_debug_pair_layout(int num_neurons, const int32_t *csc_ptr, const int32_t *csc_post, const float *csc_w, int wpc,
                          unsigned long long band_addr, unsigned long long rem_addr, long *band_floats, long *n_list_entries,
                          int *pitch_bytes, float *band_out, uint32_t *rem_out, uint32_t *rec_out);

/* Diagnostic builds (-DLSM_STAMP=1) only: per-phase s_memtime sums of the reservoir kernel
 * (out8: 8 counters, HOST memory); all zeros in the shipped build. */
int lsm_debug_lif_stamps(unsigned long long *out8, int reset);

#ifdef __cplusplus
}
#endif
#endif /* LSM_HIP_H */
