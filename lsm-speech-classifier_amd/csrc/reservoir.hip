// LIF reservoir time loop + spike features for gfx950 (SPEC.md §3-§5).
//
// Replaces, batched over clips, the per-clip sequence
//     lsm.reset(); lsm.set_input_spike_times(sample); lsm.simulate();
//     lsm.extract_features_from_spikes()
// of /root/reference/extract_lsm_features.py:79-83 (arithmetic in snn_reservoir_py, absent;
// this build's SPEC.md is normative).
//
// One workgroup = one clip = WPC wavefronts.  Wave w owns neurons [w*SL*64, (w+1)*SL*64): their
// membrane state in registers (SL per lane), their synaptic-current accumulators in LDS.  Per
// step:
//   1. every wave reads the previous step's spike bitmap (LDS) and, for each spiking neuron j in
//      ascending order, adds the weights of j's synapses onto ITS OWN targets with ds_add_f32
//      (event-driven scatter over a CSC copy of W; per target the additions arrive in ascending
//      j, and only zero terms are skipped, so the fp32 sum equals the oracle's gather sum bit for
//      bit);
//   2. input drive: every (channel -> target) entry whose channel bit is set at this step adds
//      w_in (one constant, so the order of these equal addends cannot matter);
//   3. leak/integrate/threshold/reset/refractory for the wave's neurons, __ballot -> 64-bit spike
//      words into the other bitmap buffer, feature accumulators (LDS) updated by the lanes that
//      fired; one workgroup barrier.
// No MFMA: the update is sparse and integer/byte dominated.
#include "lsm_common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

struct LifArgs {
    int N, C, T, B;
    int n_out, CW, EinW, refractory, burst_isi_max;
    float theta, w_in;
    const uint8_t *raster;     // (B, C, T) uint8
    const uint2 *seg;          // (N, WPC) {begin, end} into syn
    const uint2 *syn;          // CSC entries {target neuron, weight bits}
    const float *leak;         // (NPAD)
    const int *oslot;          // (NPAD) output slot or -1
    const uint32_t *in_ent;    // (WPC, EinW) (channel << 16) | target, 0xFFFFFFFF = padding
    int n_keys;
    int key_ids[8];
    float *features;           // (B, n_keys * n_out)
    uint8_t *spike_matrix;     // (B, T, N) or null
    float *v_trace;            // (B, T, N) or null
};

__device__ __forceinline__ void lds_add(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int SL, int WPC>
__global__ __launch_bounds__(WPC * 64) void lif_kernel(const LifArgs a)
{
    constexpr int NPW = SL * 64;
    constexpr int NPAD = NPW * WPC;
    constexpr int NW64 = NPAD / 64;
    constexpr int NT = WPC * 64;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                                   // NPAD
    unsigned long long *bitmap = reinterpret_cast<unsigned long long *>(acc + NPAD); // 2*NW64
    uint4 *feat = reinterpret_cast<uint4 *>(bitmap + 2 * NW64);                      // n_out
    uint32_t *in_ent = reinterpret_cast<uint32_t *>(feat + a.n_out);                 // WPC*EinW
    uint32_t *bits = in_ent + WPC * a.EinW;                                          // T*CW
    uint32_t *slist = bits + a.T * a.CW;                                             // WPC*64

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int N = a.N, T = a.T, CW = a.CW;

    // ---- prologue: zero LDS state, stage input map, bit-pack the clip's raster time-major ----
    for (int i = tid; i < NPAD; i += NT) acc[i] = 0.0f;
    for (int i = tid; i < 2 * NW64; i += NT) bitmap[i] = 0ull;
    for (int i = tid; i < a.n_out; i += NT) feat[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < WPC * a.EinW; i += NT) in_ent[i] = a.in_ent[i];
    for (int i = tid; i < T * CW; i += NT) bits[i] = 0u;
    __syncthreads();
    {
        const uint8_t *clip = a.raster + (size_t)b * a.C * T;
        if ((T & 3) == 0) {
            const uint32_t *clip4 = reinterpret_cast<const uint32_t *>(clip);
            const int nd = a.C * T / 4;
            for (int q = tid; q < nd; q += NT) {
                const uint32_t v = clip4[q];
                if (v == 0) continue;
                const int c = (q * 4) / T;
                const int t0 = (q * 4) - c * T;
                const uint32_t bit = 1u << (c & 31);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((v >> (8 * k)) & 0xFFu) atomicOr(&bits[(t0 + k) * CW + (c >> 5)], bit);
            }
        } else {
            const int nb = a.C * T;
            for (int q = tid; q < nb; q += NT)
                if (clip[q]) {
                    const int c = q / T;
                    atomicOr(&bits[(q - c * T) * CW + (c >> 5)], 1u << (c & 31));
                }
        }
    }

    float v[SL], lam[SL];
    int ref[SL], os[SL];
#pragma unroll
    for (int r = 0; r < SL; ++r) {
        const int i = (w * SL + r) * 64 + lane;
        v[r] = 0.0f;
        ref[r] = 0;
        lam[r] = a.leak[i];
        os[r] = a.oslot[i];
    }
    const float theta = a.theta, w_in = a.w_in;
    uint32_t *my_list = slist + w * 64;
    const uint32_t *my_ent = in_ent + w * a.EinW;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        const unsigned long long *bm_prev = bitmap + (cur ^ 1) * NW64;

        // ---- 1. recurrent scatter: spikes of step t-1, ascending j, batches of <= 64 ----
        int cnt = 0;
        for (int r = 0; r <= NW64; ++r) {
            unsigned long long m = 0ull;
            int pc = 0;
            if (r < NW64) {
                m = bm_prev[r];
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)m);
                const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(m >> 32));
                m = ((unsigned long long)hi << 32) | lo;
                if (m == 0ull) continue;
                pc = __popcll(m);
            }
            if (cnt > 0 && (r == NW64 || cnt + pc > 64)) {
                // flush the batch: lane l takes the l-th spiking neuron
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint2 sp = make_uint2(0u, 0u);
                if (lane < cnt) sp = a.seg[(size_t)my_list[lane] * WPC + w];
                for (int s0 = 0; s0 < cnt; s0 += 4) {
                    uint32_t beg[4], end[4];
                    uint2 ent[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = s0 + u;
                        const int ss = s < cnt ? s : 0;
                        beg[u] = __builtin_amdgcn_readlane(sp.x, ss);
                        end[u] = __builtin_amdgcn_readlane(sp.y, ss);
                        if (s >= cnt) end[u] = beg[u];
                        const uint32_t e = beg[u] + lane;
                        ent[u] = make_uint2(0xFFFFFFFFu, 0u);
                        if (e < end[u]) ent[u] = a.syn[e];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ent[u].x != 0xFFFFFFFFu) lds_add(acc + ent[u].x, __uint_as_float(ent[u].y));
                        for (uint32_t base = beg[u] + 64; base < end[u]; base += 64) {
                            const uint32_t e = base + lane;
                            if (e < end[u]) {
                                const uint2 x = a.syn[e];
                                lds_add(acc + x.x, __uint_as_float(x.y));
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                cnt = 0;
            }
            if (r < NW64) {
                if ((m >> lane) & 1ull) {
                    const int rank = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    my_list[rank] = (uint32_t)(r * 64 + lane);
                }
                cnt += pc;
            }
        }

        // ---- 2. input drive at step t ----
        {
            const uint32_t *row = bits + t * CW;
            for (int e = lane; e < a.EinW; e += 64) {
                const uint32_t x = my_ent[e];
                if (x != 0xFFFFFFFFu) {
                    const uint32_t c = x >> 16;
                    if ((row[c >> 5] >> (c & 31)) & 1u) lds_add(acc + (x & 0xFFFFu), w_in);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- 3. neuron update ----
#pragma unroll
        for (int r = 0; r < SL; ++r) {
            const int i = (w * SL + r) * 64 + lane;
            const float cin = acc[i];
            acc[i] = 0.0f;
            bool fire = false;
            if (ref[r] > 0) {
                ref[r] -= 1;
                v[r] = 0.0f;
            } else {
                const float m = lam[r] * v[r];
                const float d = v[r] - m;
                const float vn = d + cin;
                fire = vn >= theta;
                v[r] = fire ? 0.0f : vn;
                ref[r] = fire ? a.refractory : 0;
            }
            const unsigned long long bal = __ballot(fire);
            if (lane == 0) bitmap[cur * NW64 + w * SL + r] = bal;
            if (i < N) {
                if (a.spike_matrix) a.spike_matrix[((size_t)b * T + t) * N + i] = fire ? 1 : 0;
                if (a.v_trace) a.v_trace[((size_t)b * T + t) * N + i] = v[r];
            }
            if (fire && os[r] >= 0) {
                uint4 f = feat[os[r]];
                uint32_t n = f.x & 0xFFFFu, bursts = f.x >> 16;
                uint32_t first = f.y & 0xFFFFu, last = f.y >> 16;
                if (n == 0) {
                    first = (uint32_t)t;
                } else {
                    const uint32_t isi = (uint32_t)t - last;
                    f.w += isi * isi;
                    if ((int)isi <= a.burst_isi_max) bursts += 1;
                }
                last = (uint32_t)t;
                n += 1;
                f.z += (uint32_t)t;
                f.x = n | (bursts << 16);
                f.y = first | (last << 16);
                feat[os[r]] = f;
            }
        }
        __syncthreads();
    }

    // ---- epilogue: SPEC.md §4 features from the integer accumulators (float64, then float32) ----
    const int nf = a.n_keys * a.n_out;
    for (int idx = tid; idx < nf; idx += NT) {
        const int kq = idx / a.n_out;
        const int o = idx - kq * a.n_out;
        const uint4 f = feat[o];
        const int n = (int)(f.x & 0xFFFFu), bursts = (int)(f.x >> 16);
        const int first = (int)(f.y & 0xFFFFu), last = (int)(f.y >> 16);
        double val = 0.0;
        switch (a.key_ids[kq]) {
        case 0: val = (double)n; break;
        case 1: { const double p = (double)n / (double)T; val = p * (1.0 - p); } break;
        case 2: val = n >= 1 ? (double)f.z / (double)n : 0.0; break;
        case 3: val = n >= 1 ? (double)first : 0.0; break;
        case 4: val = n >= 1 ? (double)last : 0.0; break;
        case 5: val = n >= 2 ? (double)(last - first) / (double)(n - 1) : 0.0; break;
        case 6:
            if (n >= 2) {
                const double m = (double)(last - first) / (double)(n - 1);
                val = (double)f.w / (double)(n - 1) - m * m;
            }
            break;
        default: val = (double)bursts; break;
        }
        a.features[(size_t)b * nf + idx] = (float)val;
    }
}

typedef void (*lif_fn_t)(const LifArgs);

template <int SL>
lif_fn_t pick_wpc(int wpc)
{
    switch (wpc) {
    case 1: return lif_kernel<SL, 1>;
    case 2: return lif_kernel<SL, 2>;
    case 4: return lif_kernel<SL, 4>;
    case 8: return lif_kernel<SL, 8>;
    case 16: return lif_kernel<SL, 16>;
    default: return nullptr;
    }
}

lif_fn_t pick_kernel(int sl, int wpc)
{
    switch (sl) {
    case 1: return pick_wpc<1>(wpc);
    case 2: return pick_wpc<2>(wpc);
    case 4: return pick_wpc<4>(wpc);
    case 8: return pick_wpc<8>(wpc);
    case 16: return pick_wpc<16>(wpc);
    default: return nullptr;
    }
}

struct Variant {            // per waves-per-clip layout
    int wpc = 0, sl = 0, einw = 0;
    uint2 *seg = nullptr;
    float *leak = nullptr;
    int *oslot = nullptr;
    uint32_t *in_ent = nullptr;
};

}  // namespace

struct lsm_reservoir {
    int N = 0, C = 0, n_out = 0, refractory = 0, burst_isi_max = 0;
    float theta = 0, w_in = 0;
    int device = 0;
    size_t nnz = 0;
    uint2 *syn = nullptr;
    Variant var[5];         // wpc = 1, 2, 4, 8, 16 (wpc == 0: not available)
};

static int free_reservoir(lsm_reservoir *h)
{
    if (!h) return LSM_OK;
    if (h->syn) (void)hipFree(h->syn);
    for (auto &v : h->var) {
        if (v.seg) (void)hipFree(v.seg);
        if (v.leak) (void)hipFree(v.leak);
        if (v.oslot) (void)hipFree(v.oslot);
        if (v.in_ent) (void)hipFree(v.in_ent);
    }
    delete h;
    return LSM_OK;
}

template <typename T>
static int upload(T **dst, const std::vector<T> &src)
{
    LSM_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(1, src.size()) * sizeof(T)));
    if (!src.empty())
        LSM_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return LSM_OK;
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_create(lsm_reservoir **out, int num_neurons, int n_channels,
                         const int32_t *csc_ptr, const int32_t *csc_post, const float *csc_w,
                         const float *leak, const int32_t *in_tgt, int in_fanout, float w_in,
                         const int32_t *out_idx, int n_out, float theta, int refractory,
                         int burst_isi_max)
{
    LSM_REQUIRE(out != nullptr, "lsm_reservoir_create: out is null");
    *out = nullptr;
    const int N = num_neurons, C = n_channels;
    LSM_REQUIRE(N >= 1 && N <= 16384, "num_neurons=%d outside [1, 16384]", N);
    LSM_REQUIRE(C >= 1 && C <= 65535, "n_channels=%d outside [1, 65535]", C);
    LSM_REQUIRE(n_out >= 1 && n_out <= N, "num_output_neurons=%d outside [1, %d]", n_out, N);
    LSM_REQUIRE(theta > 0.0f, "membrane_threshold must be > 0");
    LSM_REQUIRE(refractory >= 0 && in_fanout >= 1 && in_fanout <= N, "bad refractory/in_fanout");
    LSM_REQUIRE(csc_ptr && csc_post && csc_w && leak && in_tgt && out_idx, "null array argument");
    LSM_REQUIRE(csc_ptr[0] == 0, "csc_ptr[0] must be 0");
    const size_t nnz = (size_t)csc_ptr[N];
    for (int j = 0; j < N; ++j) {
        LSM_REQUIRE(csc_ptr[j + 1] >= csc_ptr[j], "csc_ptr not monotone at %d", j);
        for (int e = csc_ptr[j]; e < csc_ptr[j + 1]; ++e) {
            LSM_REQUIRE(csc_post[e] >= 0 && csc_post[e] < N, "csc_post[%d] out of range", e);
            LSM_REQUIRE(e == csc_ptr[j] || csc_post[e] > csc_post[e - 1],
                        "csc_post not strictly ascending within column %d", j);
        }
    }
    for (int o = 0; o < n_out; ++o)
        LSM_REQUIRE(out_idx[o] >= 0 && out_idx[o] < N && (o == 0 || out_idx[o] > out_idx[o - 1]),
                    "out_idx must be strictly ascending in [0, N)");
    for (int e = 0; e < C * in_fanout; ++e)
        LSM_REQUIRE(in_tgt[e] >= 0 && in_tgt[e] < N, "in_tgt[%d] out of range", e);

    lsm_reservoir *h = new lsm_reservoir();
    h->N = N; h->C = C; h->n_out = n_out; h->refractory = refractory;
    h->burst_isi_max = burst_isi_max; h->theta = theta; h->w_in = w_in; h->nnz = nnz;
    (void)hipGetDevice(&h->device);

    std::vector<uint2> syn(nnz);
    for (size_t e = 0; e < nnz; ++e) {
        uint32_t bits;
        std::memcpy(&bits, &csc_w[e], 4);
        syn[e] = make_uint2((uint32_t)csc_post[e], bits);
    }
    int rc = upload(&h->syn, syn);
    if (rc) { free_reservoir(h); return rc; }

    const int wpcs[5] = {1, 2, 4, 8, 16};
    for (int vi = 0; vi < 5; ++vi) {
        const int wpc = wpcs[vi];
        const int need = ((N + wpc - 1) / wpc + 63) / 64;      // 64-neuron slots per wave
        int sl = 1;
        while (sl < need) sl <<= 1;
        if (sl > 16) continue;
        const int npw = sl * 64, npad = npw * wpc;
        Variant &v = h->var[vi];
        std::vector<uint2> seg((size_t)N * wpc);
        for (int j = 0; j < N; ++j) {
            int e = csc_ptr[j];
            for (int w = 0; w < wpc; ++w) {
                const int hi = (w + 1) * npw;
                const int beg = e;
                while (e < csc_ptr[j + 1] && csc_post[e] < hi) ++e;
                seg[(size_t)j * wpc + w] = make_uint2((uint32_t)beg, (uint32_t)e);
            }
        }
        std::vector<float> lk(npad, 0.0f);
        std::vector<int> os(npad, -1);
        for (int i = 0; i < N; ++i) lk[i] = leak[i];
        for (int o = 0; o < n_out; ++o) os[out_idx[o]] = o;
        std::vector<std::vector<uint32_t>> per(wpc);
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < in_fanout; ++d) {
                const int tgt = in_tgt[(size_t)c * in_fanout + d];
                per[tgt / npw].push_back(((uint32_t)c << 16) | (uint32_t)tgt);
            }
        size_t mx = 1;
        for (auto &p : per) mx = std::max(mx, p.size());
        const int einw = (int)((mx + 63) / 64 * 64);
        std::vector<uint32_t> ent((size_t)wpc * einw, 0xFFFFFFFFu);
        for (int w = 0; w < wpc; ++w)
            std::copy(per[w].begin(), per[w].end(), ent.begin() + (size_t)w * einw);
        if ((rc = upload(&v.seg, seg)) || (rc = upload(&v.leak, lk)) || (rc = upload(&v.oslot, os)) ||
            (rc = upload(&v.in_ent, ent))) {
            free_reservoir(h);
            return rc;
        }
        v.wpc = wpc; v.sl = sl; v.einw = einw;
    }
    *out = h;
    return LSM_OK;
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_destroy(lsm_reservoir *h) { return free_reservoir(h); }

static size_t lif_lds_bytes(const lsm_reservoir *h, const Variant &v, int T)
{
    const size_t npad = (size_t)v.sl * 64 * v.wpc;
    const size_t cw = (size_t)(h->C + 31) / 32;
    return npad * 4 + 2 * (npad / 64) * 8 + (size_t)h->n_out * 16 + (size_t)v.wpc * v.einw * 4 +
           (size_t)T * cw * 4 + (size_t)v.wpc * 64 * 4;
}

// Pick the waves-per-clip layout: enough wavefronts to cover the chip's 1024 SIMDs about twice,
// among the layouts this reservoir supports and whose LDS image fits one CU.
static const Variant *choose_variant(const lsm_reservoir *h, int B, int T, int requested)
{
    const Variant *best = nullptr;
    if (requested > 0) {
        for (const auto &v : h->var)
            if (v.wpc == requested && lif_lds_bytes(h, v, T) <= 160 * 1024) return &v;
        return nullptr;
    }
    for (const auto &v : h->var) {
        if (!v.wpc || lif_lds_bytes(h, v, T) > 160 * 1024) continue;
        if (!best) { best = &v; continue; }
        if ((long)B * best->wpc < 2048) best = &v;       // keep widening while the chip is underfilled
    }
    return best;
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_run(const lsm_reservoir *h, const uint8_t *spikes_u8, int n_clips, int n_steps,
                      const int32_t *key_ids, int n_keys, float *features_out,
                      uint8_t *spike_matrix_out, float *v_trace_out, int waves_per_clip,
                      void *stream)
{
    LSM_REQUIRE(h != nullptr, "lsm_reservoir_run: null handle");
    LSM_REQUIRE(n_clips >= 0 && n_steps >= 1 && n_steps <= 65535, "bad n_clips/n_steps");
    LSM_REQUIRE(n_keys >= 1 && n_keys <= 8 && key_ids, "n_keys must be in [1, 8]");
    LSM_REQUIRE(spikes_u8 && features_out, "null buffer");
    if (n_clips == 0) return LSM_OK;
    const Variant *v = choose_variant(h, n_clips, n_steps, waves_per_clip);
    LSM_REQUIRE(v != nullptr, "no reservoir layout for waves_per_clip=%d (N=%d, T=%d)",
                waves_per_clip, h->N, n_steps);
    lif_fn_t fn = pick_kernel(v->sl, v->wpc);
    LSM_REQUIRE(fn != nullptr, "no kernel for SL=%d WPC=%d", v->sl, v->wpc);

    LifArgs a;
    a.N = h->N; a.C = h->C; a.T = n_steps; a.B = n_clips;
    a.n_out = h->n_out; a.CW = (h->C + 31) / 32; a.EinW = v->einw;
    a.refractory = h->refractory; a.burst_isi_max = h->burst_isi_max;
    a.theta = h->theta; a.w_in = h->w_in;
    a.raster = spikes_u8; a.seg = v->seg; a.syn = h->syn; a.leak = v->leak; a.oslot = v->oslot;
    a.in_ent = v->in_ent;
    a.n_keys = n_keys;
    for (int k = 0; k < 8; ++k) a.key_ids[k] = 0;
    for (int k = 0; k < n_keys; ++k) {
        LSM_REQUIRE(key_ids[k] >= 0 && key_ids[k] < 8, "key id %d out of range", key_ids[k]);
        a.key_ids[k] = key_ids[k];
    }
    a.features = features_out; a.spike_matrix = spike_matrix_out; a.v_trace = v_trace_out;

    const size_t lds = lif_lds_bytes(h, *v, n_steps);
    if (lds > 64 * 1024)
        LSM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3(n_clips), dim3(v->wpc * 64), lds, (hipStream_t)stream, a);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

// Introspection for tests and the bench: layout chosen for a batch, LDS bytes per workgroup.
extern "C" __attribute__((visibility("default")))
int lsm_reservoir_layout(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip,
                         int *wpc_out, int *slots_out, int *lds_bytes_out)
{
    LSM_REQUIRE(h != nullptr, "lsm_reservoir_layout: null handle");
    const Variant *v = choose_variant(h, n_clips, n_steps, waves_per_clip);
    LSM_REQUIRE(v != nullptr, "no reservoir layout for waves_per_clip=%d", waves_per_clip);
    if (wpc_out) *wpc_out = v->wpc;
    if (slots_out) *slots_out = v->sl;
    if (lds_bytes_out) *lds_bytes_out = (int)lif_lds_bytes(h, *v, n_steps);
    return LSM_OK;
}
