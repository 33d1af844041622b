// lif_kernel.h -- LIF reservoir time loop + spike features for gfx950 (SPEC.md §3-§5).
//
// Replaces, batched over clips, the per-clip sequence
//     lsm.reset(); lsm.set_input_spike_times(sample); lsm.simulate();
//     lsm.extract_features_from_spikes()
// of /root/reference/extract_lsm_features.py:79-83 (arithmetic in snn_reservoir_py, absent;
// this build's SPEC.md is normative).
//
// One workgroup = one clip = WPC wavefronts.  Wave w owns neurons [w*SL*64, (w+1)*SL*64): their
// membrane state in registers (SL per lane), their synaptic-current accumulators in LDS.  At small
// batches the kernel is bound by the per-step dependency chain and, on gfx950, by TAKEN BRANCHES
// (~20-30 cycles each, measured), so every step is written as near-straight-line code:
//   a. spike lists come from the PRODUCER: in the update of step t-1 each wave turns its
//      __ballot masks into ranks (mbcnt) and writes its spiking neurons, ascending, into its own
//      LDS list plus one count; the consumer reads the WPC counts (one LDS read), forms their
//      prefix in scalar registers and lane l picks the l-th spiking neuron of the whole clip
//      (wave order x rank = ascending j) with a compare chain -- no bit scanning, no loop;
//   b. ONE LDS read fetches each listed neuron's segment bounds (LDS-resident table when it
//      fits); a ballot keeps only the neurons with synapses onto this wave's targets (at 16 waves per
//      clip three quarters of the segments are empty) and the synapse entries {target, weight} of up to 8 spiking neurons are loaded at
//      once from the CSC copy of W (L2-resident at N=1000), through nested "one more?" tests so
//      that a group of n <= 8 costs one taken branch;
//   c. while those loads fly, the input drive of step t is COUNTED: every (channel -> target)
//      entry does ds_add_u32(count[target], channel bit) unconditionally (integer LDS atomics run
//      at full rate; float ones take ~3 cycles per lane on gfx950, measured);
//   d. the recurrent weights are added onto the wave's OWN targets, one spiking neuron after the
//      other in ascending j, with a plain LDS read + v_add_f32 + write: the targets of one
//      neuron are distinct, and LDS executes a wave's instructions in order, so this is the
//      oracle's per-target sum "presynaptic j ascending" with only its zero terms skipped --
//      bit-identical in fp32;
//   e. update from registers: current = LDS sum + w_in * count[target] (SPEC.md §3: the input term
//      comes after the recurrent terms),
//      leak/integrate/threshold/reset/refractory by select, spike lists + feature accumulators
//      (LDS) touched only inside one "any lane of this wave fired" branch; one barrier.
// No MFMA: the update is sparse and integer/byte dominated.
#pragma once
#include "lsm_common.h"

namespace lsm_lif {

#ifndef LSM_STAMP
#define LSM_STAMP 0         // diagnostic builds only: per-phase s_memtime sums (wave 0 of each clip)
#endif
#if LSM_STAMP
namespace { __device__ unsigned long long g_lif_stamps[8]; }   // one copy per translation unit
#define STAMP(k)                                                                             \
    do {                                                                                     \
        unsigned long long _t;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        st_sum[k] += _t - st_last;                                                           \
        st_last = _t;                                                                        \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

constexpr int IN_REG_SLOTS = 6;       // input-map entries per lane kept in registers
constexpr int SPIKE_GROUP = 8;        // spiking neurons whose synapse loads are in flight together

struct LifArgs {
    int N, C, T, B;
    int n_out, CW, EinW, refractory, burst_isi_max;
    float theta, w_in;
    const uint8_t *raster;     // (B, C, T) uint8
    const uint32_t *seg;       // (N*WPC + 1) begin offsets into syn, row-major (neuron, wave)
    const uint32_t *rowptr;    // (N + 1) first synapse of each presynaptic neuron (= CSC pointer)
    const uint16_t *segoff;    // (N, WPC + 1) segment starts relative to rowptr[j]; [WPC] = row length
    const uint2 *syn;          // CSC entries {target neuron, weight bits}
    const float *leak;         // (NPAD)
    const int *oslot;          // (NPAD) output slot or -1
    const uint32_t *in_ent;    // (WPC, EinW) (channel << 16) | target, 0xFFFFFFFF = padding
    int n_keys;
    int key_ids[8];
    float *features;           // (B, n_keys * n_out)
    uint8_t *spike_matrix;     // (B, T, N) or null
    float *v_trace;            // (B, T, N) or null
    int32_t *stats;            // (B, 2) {neurons that fired at least once, spikes of the whole reservoir} or null
    const int32_t *order;      // (B) clip of workgroup g, or null (g): lsm_reservoir_run_ordered starts long clips first
};

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_rank(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// INREG: the wave's input-map entries fit IN_REG_SLOTS registers per lane (else they are streamed
// from global memory every step: static addresses, L2-resident).
// SEGLDS: the segment table is staged in LDS (else read from global memory).  Both are template
// parameters so that every access keeps its own address space: a run-time choice between an LDS
// and a global pointer compiles to flat loads, whose waits serialise the row loads behind them.
template <int SL, int WPC, bool INREG, bool SEGLDS>
__global__ __launch_bounds__(WPC * 64) void lif_kernel(const LifArgs a)
{
    constexpr int NPW = SL * 64;
    constexpr int NPAD = NPW * WPC;
    constexpr int NT = WPC * 64;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                                    // NPAD
    uint32_t *icnt = reinterpret_cast<uint32_t *>(acc + NPAD);                        // NPAD
    uint16_t *wlist = reinterpret_cast<uint16_t *>(icnt + NPAD);                      // 2*NPAD
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(wlist + 2 * NPAD);                  // 2*16
    uint4 *feat = reinterpret_cast<uint4 *>(wcnt + 32);                               // n_out
    uint32_t *bits = reinterpret_cast<uint32_t *>(feat + a.n_out);                    // T*CW
    uint32_t *lrow = bits + a.T * a.CW;                                               // N+1 (if SEGLDS)
    uint16_t *lso = reinterpret_cast<uint16_t *>(lrow + (a.N + 1));                     // N*(WPC+1) (if SEGLDS)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;      // wave-uniform
    const int N = a.N, T = a.T, CW = a.CW;

    // ---- prologue: zero LDS state, stage tables, bit-pack the clip's raster time-major ----
    for (int i = tid; i < NPAD; i += NT) { acc[i] = 0.0f; icnt[i] = 0u; }
    if (tid < 32) wcnt[tid] = 0u;
    for (int i = tid; i < a.n_out; i += NT) feat[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < T * CW; i += NT) bits[i] = 0u;
    if (SEGLDS) {
        for (int i = tid; i < N + 1; i += NT) lrow[i] = a.rowptr[i];
        const uint32_t *so32 = reinterpret_cast<const uint32_t *>(a.segoff);
        uint32_t *lso32 = reinterpret_cast<uint32_t *>(lso);
        for (int i = tid; i < (N * (WPC + 1) + 1) / 2; i += NT) lso32[i] = so32[i];
    }
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
    // input-map entries of this wave: word offset of the channel bit, its mask, target neuron.
    // Padding entries get mask 0 and a target of their own (this lane's first neuron): they add 0,
    // and never pile up on one LDS address (same-address atomics serialise).
    uint32_t in_word[IN_REG_SLOTS], in_mask[IN_REG_SLOTS], in_tgt[IN_REG_SLOTS];
    if (INREG) {
#pragma unroll
        for (int q = 0; q < IN_REG_SLOTS; ++q) {
            const int e = q * 64 + lane;
            const uint32_t x = e < a.EinW ? a.in_ent[(size_t)w * a.EinW + e] : 0xFFFFFFFFu;
            const bool ok = x != 0xFFFFFFFFu;
            const uint32_t c = x >> 16;
            in_word[q] = ok ? (c >> 5) : 0u;
            in_mask[q] = ok ? (1u << (c & 31)) : 0u;
            in_tgt[q] = ok ? (x & 0xFFFFu) : (uint32_t)(w * NPW + lane);
        }
    }
    const float theta = a.theta, w_in = a.w_in;
    const uint32_t *my_ent = a.in_ent + (size_t)w * a.EinW;   // !INREG: streamed from L2 each step
    const bool trace = a.spike_matrix != nullptr || a.v_trace != nullptr;
    uint32_t hf = 0u;                  // bit r: my neuron r fired at least once (stats)
    uint32_t tot_spk = 0u;             // spikes of my wave (stats)
    __syncthreads();

    // input drive of step `ts`: count the active channels feeding each target (integer atomics,
    // unconditional: an inactive or padding entry adds 0)
    auto input_drive = [&](int ts) {
        const uint32_t *row = bits + ts * CW;
        if (INREG) {
            uint32_t wd[IN_REG_SLOTS];
#pragma unroll
            for (int q = 0; q < IN_REG_SLOTS; ++q) wd[q] = row[in_word[q]];
#pragma unroll
            for (int q = 0; q < IN_REG_SLOTS; ++q) {
                atomicAdd(icnt + in_tgt[q], (wd[q] & in_mask[q]) ? 1u : 0u);
            }
        } else {
            for (int e = lane; e < a.EinW; e += 64) {
                const uint32_t x = my_ent[e];
                if (x != 0xFFFFFFFFu) {
                    const uint32_t c = x >> 16;
                    atomicAdd(icnt + (x & 0xFFFFu), (row[c >> 5] >> (c & 31)) & 1u);
                }
            }
        }
    };

#if LSM_STAMP
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
    const unsigned long long st_loop0 = st_last, st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        const uint16_t *list_prev = wlist + (cur ^ 1) * NPAD;
        uint16_t *list_cur = wlist + cur * NPAD + w * NPW;

        // ---- a. spike list of step t-1: prefix of the per-wave counts, lane l <- l-th neuron ----
        const uint32_t cv = wcnt[(cur ^ 1) * 16 + (lane & 15)];
        uint32_t pre[WPC + 1];
        pre[0] = 0u;
#pragma unroll
        for (int q = 0; q < WPC; ++q) pre[q + 1] = pre[q] + __builtin_amdgcn_readlane(cv, q);
        const uint32_t total = pre[WPC];
        bool drove = false;
        STAMP(0);

        for (uint32_t l0 = 0; l0 < total; l0 += 64) {
            const uint32_t l = l0 + lane;
            uint32_t wsel = 0u, pbase = 0u;
#pragma unroll
            for (int q = 1; q < WPC; ++q) {
                const bool ge = l >= pre[q];
                wsel += ge ? 1u : 0u;
                pbase = ge ? pre[q] : pbase;
            }
            uint32_t beg = 0u, end = 0u;
            if (l < total) {
                const int j = list_prev[wsel * NPW + (l - pbase)];
                const int idx = j * WPC + w;
                if (SEGLDS) {
                    const uint32_t rp = lrow[j];
                    const int io = j * (WPC + 1) + w;
                    beg = rp + lso[io];
                    end = rp + lso[io + 1];
                } else {
                    beg = a.seg[idx];
                    end = a.seg[idx + 1];
                }
            }
            // only the listed neurons with at least one synapse onto THIS wave's targets matter here
            unsigned long long todo = __ballot(beg < end);
            while (todo != 0ull) {
                // ---- b. up to 8 rows in flight, ascending list position.  Nested "one more?" tests:
                //         a group costs one taken branch (at its end), not one per slot ----
                const int n8 = min((int)__popcll(todo), SPIKE_GROUP);
                uint32_t gb[SPIKE_GROUP], ge[SPIKE_GROUP];
                uint2 ent[SPIKE_GROUP];
                bool any_long = false;
#define LSM_LD(k)                                                                   \
    {                                                                               \
        const int sk = __builtin_ctzll(todo);                                       \
        todo &= todo - 1ull;                                                        \
        gb[k] = __builtin_amdgcn_readlane(beg, sk);                                 \
        ge[k] = __builtin_amdgcn_readlane(end, sk);                                 \
        if (SL > 1) any_long |= (ge[k] - gb[k]) > 64u;   /* SL == 1: a wave owns 64 targets */ \
        if (gb[k] + lane < ge[k]) ent[k] = a.syn[gb[k] + lane];                     \
    }
                LSM_LD(0)
                if (n8 > 1) { LSM_LD(1)
                if (n8 > 2) { LSM_LD(2)
                if (n8 > 3) { LSM_LD(3)
                if (n8 > 4) { LSM_LD(4)
                if (n8 > 5) { LSM_LD(5)
                if (n8 > 6) { LSM_LD(6)
                if (n8 > 7) { LSM_LD(7) } } } } } } }
#undef LSM_LD
                STAMP(1);
                if (!drove) {                     // ---- c. fill the load latency ----
                    input_drive(t);
                    drove = true;
                }
                STAMP(2);
                // ---- d. ordered read+add+write, one spiking neuron after the other ----
#define LSM_RMW(k)                                                                  \
    if (gb[k] + lane < ge[k]) acc[ent[k].x] = acc[ent[k].x] + __uint_as_float(ent[k].y); \
    __builtin_amdgcn_wave_barrier();
#define LSM_RMW_LONG(k)                                                             \
    if (gb[k] + lane < ge[k]) acc[ent[k].x] = acc[ent[k].x] + __uint_as_float(ent[k].y); \
    for (uint32_t base = gb[k] + 64u; base < ge[k]; base += 64u) {                  \
        __builtin_amdgcn_wave_barrier();                                            \
        if (base + lane < ge[k]) {                                                  \
            const uint2 x = a.syn[base + lane];                                     \
            acc[x.x] = acc[x.x] + __uint_as_float(x.y);                             \
        }                                                                           \
    }                                                                               \
    __builtin_amdgcn_wave_barrier();
                if (SL == 1 || !any_long) {       // common case: every segment fits one instruction
                    LSM_RMW(0)
                    if (n8 > 1) { LSM_RMW(1)
                    if (n8 > 2) { LSM_RMW(2)
                    if (n8 > 3) { LSM_RMW(3)
                    if (n8 > 4) { LSM_RMW(4)
                    if (n8 > 5) { LSM_RMW(5)
                    if (n8 > 6) { LSM_RMW(6)
                    if (n8 > 7) { LSM_RMW(7) } } } } } } }
                } else {
                    LSM_RMW_LONG(0)
                    if (n8 > 1) { LSM_RMW_LONG(1)
                    if (n8 > 2) { LSM_RMW_LONG(2)
                    if (n8 > 3) { LSM_RMW_LONG(3)
                    if (n8 > 4) { LSM_RMW_LONG(4)
                    if (n8 > 5) { LSM_RMW_LONG(5)
                    if (n8 > 6) { LSM_RMW_LONG(6)
                    if (n8 > 7) { LSM_RMW_LONG(7) } } } } } } }
                }
#undef LSM_RMW
#undef LSM_RMW_LONG
                STAMP(3);
            }
        }
        if (!drove) input_drive(t);
        wave_lds_fence();
        STAMP(2);

        // ---- e. neuron update ----
        float cin[SL];
        uint32_t nin[SL];
#pragma unroll
        for (int r = 0; r < SL; ++r) {
            cin[r] = acc[(w * SL + r) * 64 + lane];
            nin[r] = icnt[(w * SL + r) * 64 + lane];
        }
        unsigned long long bal[SL];
        unsigned long long any_fire = 0ull;
#pragma unroll
        for (int r = 0; r < SL; ++r) {
            const int i = (w * SL + r) * 64 + lane;
            acc[i] = 0.0f;
            icnt[i] = 0u;
            cin[r] = cin[r] + w_in * (float)nin[r];      // SPEC.md §3: input term after the recurrent sum
        }
#pragma unroll
        for (int r = 0; r < SL; ++r) {
            const bool held = ref[r] > 0;
            const float m = lam[r] * v[r];
            const float d = v[r] - m;
            const float vn = d + cin[r];
            const bool fire = !held && (vn >= theta);
            v[r] = (held || fire) ? 0.0f : vn;
            ref[r] = held ? ref[r] - 1 : (fire ? a.refractory : 0);
            bal[r] = __ballot(fire);
            any_fire |= bal[r];
        }
        int nspk = 0;
        if (any_fire != 0ull) {                  // one branch per wave and step
#pragma unroll
            for (int r = 0; r < SL; ++r) {
                const bool fire = (bal[r] >> lane) & 1ull;
                if (fire) {
                    list_cur[nspk + lane_rank(bal[r])] = (uint16_t)((w * SL + r) * 64 + lane);
                    hf |= 1u << r;
                    if (os[r] >= 0) {
                        uint4 f = feat[os[r]];
                        uint32_t n = f.x & 0xFFFFu, bursts = f.x >> 16;
                        uint32_t first = f.y & 0xFFFFu, last = f.y >> 16;
                        const uint32_t isi = (uint32_t)t - last;
                        first = n == 0 ? (uint32_t)t : first;
                        f.w += n == 0 ? 0u : isi * isi;
                        bursts += (n != 0 && (int)isi <= a.burst_isi_max) ? 1u : 0u;
                        last = (uint32_t)t;
                        n += 1;
                        f.z += (uint32_t)t;
                        f.x = n | (bursts << 16);
                        f.y = first | (last << 16);
                        feat[os[r]] = f;
                    }
                }
                nspk += __popcll(bal[r]);
            }
        }
        tot_spk += (uint32_t)nspk;
        if (lane == 0) wcnt[cur * 16 + w] = (uint32_t)nspk;
        if (trace) {
#pragma unroll
            for (int r = 0; r < SL; ++r) {
                const int i = (w * SL + r) * 64 + lane;
                if (i < N) {
                    if (a.spike_matrix)
                        a.spike_matrix[((size_t)b * T + t) * N + i] = (uint8_t)((bal[r] >> lane) & 1ull);
                    if (a.v_trace) a.v_trace[((size_t)b * T + t) * N + i] = v[r];
                }
            }
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);
    }
#if LSM_STAMP
    st_sum[6] = st_last - st_loop0;                                   // shader cycles in the step loop
    st_sum[7] = __builtin_amdgcn_s_memrealtime() - st_real0;          // 100 MHz ticks in the step loop
    if (tid == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_lif_stamps[k], st_sum[k]);
#endif

    // ---- epilogue: health statistics (the count array is idle and all zero after the last step), then
    //      SPEC.md §4 features from the integer accumulators (float64, then float32) ----
    if (a.stats) {
        atomicAdd(&icnt[0], (uint32_t)__popc(hf));
        if (lane == 0) atomicAdd(&icnt[1], tot_spk);
        __syncthreads();
        if (tid == 0) {
            a.stats[2 * b] = (int32_t)icnt[0];
            a.stats[2 * b + 1] = (int32_t)icnt[1];
        }
    }
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

template <int SL, bool INREG, bool SEGLDS>
lif_fn_t pick_wpc(int wpc)
{
    switch (wpc) {
    case 1: return lif_kernel<SL, 1, INREG, SEGLDS>;
    case 2: return lif_kernel<SL, 2, INREG, SEGLDS>;
    case 4: return lif_kernel<SL, 4, INREG, SEGLDS>;
    case 8: return lif_kernel<SL, 8, INREG, SEGLDS>;
    case 16: return lif_kernel<SL, 16, INREG, SEGLDS>;
    default: return nullptr;
    }
}

template <bool INREG, bool SEGLDS>
lif_fn_t pick_sl(int sl, int wpc)
{
    switch (sl) {
    case 1: return pick_wpc<1, INREG, SEGLDS>(wpc);
    case 2: return pick_wpc<2, INREG, SEGLDS>(wpc);
    case 4: return pick_wpc<4, INREG, SEGLDS>(wpc);
    case 8: return pick_wpc<8, INREG, SEGLDS>(wpc);
    case 16: return pick_wpc<16, INREG, SEGLDS>(wpc);
    default: return nullptr;
    }
}

// one definition per translation unit lif_variant_<inreg><seglds>.hip
lif_fn_t pick_lif_00(int sl, int wpc);
lif_fn_t pick_lif_01(int sl, int wpc);
lif_fn_t pick_lif_10(int sl, int wpc);
lif_fn_t pick_lif_11(int sl, int wpc);
#if LSM_STAMP
int read_lif_stamps(int unit, unsigned long long *out8, int reset);   // reservoir.hip sums the four units
int read_lif_stamps_00(unsigned long long *o, int r);
int read_lif_stamps_01(unsigned long long *o, int r);
int read_lif_stamps_10(unsigned long long *o, int r);
int read_lif_stamps_11(unsigned long long *o, int r);
#define LSM_DEFINE_STAMP_READER(NAME)                                               \
    int NAME(unsigned long long *o, int r)                                          \
    {                                                                               \
        LSM_CHECK_HIP(hipDeviceSynchronize());                                      \
        LSM_CHECK_HIP(hipMemcpyFromSymbol(o, HIP_SYMBOL(g_lif_stamps), 64));        \
        if (r) {                                                                    \
            unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};                     \
            LSM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lif_stamps), z, 64));      \
        }                                                                           \
        return LSM_OK;                                                              \
    }
#endif

}  // namespace lsm_lif
