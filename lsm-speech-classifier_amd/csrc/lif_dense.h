// lif_dense.h -- LIF reservoir time loop on dense presynaptic rows (N <= 8192), gfx950.
//
// Same contract as lif_kernel.h (SPEC.md §3-§4; replaces reset/set_input_spike_times/simulate/
// extract_features_from_spikes of /root/reference/extract_lsm_features.py:79-83), different data
// structure: W is stored as dense rows by PRESYNAPTIC neuron, Wt[j][i] (0 where there is no synapse),
// N x LD floats: 4 MB at N = 1000 (L2 resident), 64 MB at N = 4000 (Infinity Cache), 262 MB at N = 8000.  For every neuron j that spiked
// at t-1, ascending, each lane loads the weight onto ITS OWN target neuron (one coalesced 256-byte load
// per wave and 64-neuron slot) and adds it to a REGISTER accumulator.  The oracle's per-target sum runs
// over the existing synapses in ascending j; the extra terms here are exact zeros and x + 0 = x in
// float32, so the result is bit-identical.  What this removes compared with the sparse kernel: the LDS
// current accumulators and their ordered read-modify-write chain (a burst of n spikes cost n dependent LDS
// round trips on the few waves owning the targets), the segment tables, the exec masking -- every wave does
// the same S loads + S adds, so bursts no longer unbalance the waves.  What it costs: N*4 bytes per spiking
// neuron and clip from L2/MALL instead of ~8 bytes per synapse (4 KB vs 1.6 KB at N = 1000).
//
// Spike exchange: each producer wave writes its spiking neurons (ballot + mbcnt ranks, ascending) into its
// own LDS list; the first R = 64/WPC also go to the wave's fixed region of a 64-entry step list, so lane l of
// a consumer knows its entry (l) and whether it is filled (l%R < count of producer l/R) without any merge;
// when a producer has more than R spikes (its count says so) the consumers merge the per-wave lists for that
// step instead (scalar prefix over the counts + compare chain).
#pragma once
#include "lif_kernel.h"

#ifndef LSM_ABLATE
#define LSM_ABLATE 0        // diagnostic builds only (results are WRONG): 1 = no recurrent rows, 2 = no input
#endif                      // drive, 4 = no step barrier, 8 = no feature updates, 16 = no loads (adds kept),
                            // 32 = rows folded onto the first 256 (1 MB footprint), 64 = non-temporal raster reads

// Row fetch, same-box A/B at 128 filters / 1000 neurons / 256 clips (profiles/r02_dense_row_chain_ab.txt):
// LSM_DENSE_BUF=1 issues the row loads as MUBUF `buffer_load ... offen` with the row offset as scalar operand
// (no per-row vector address add) - and the launch takes 0.60 ms instead of 0.53: with eight waves of a CU
// issuing dword gathers, buffer loads run at 3/4 of the rate of global loads (exp/ubench_vmem.hip,
// profiles/r02_ubench_global_vs_buffer_loads.txt).  Kept as a switch so that the measurement can be repeated.
// LSM_DENSE_BITSET: clear the consumed list bit with one s_bitset0_b64 instead of the three-instruction
// `todo &= todo - 1`; with the row offsets premultiplied once per 64 entries a row costs 9 instructions per
// wave instead of 12 (0.532 -> 0.519 ms).
#ifndef LSM_DENSE_BUF
#define LSM_DENSE_BUF 0
#endif
#ifndef LSM_DENSE_BITSET
#define LSM_DENSE_BITSET 1
#endif
namespace lsm_lif {

struct DenseArgs {
    int N, C, T, B;
    int n_out, CW, EinW, refractory, burst_isi_max, ld;
    float theta, w_in;
    const uint8_t *raster;     // (B, C, T) uint8
    const float *wt;           // (N, ld) dense rows by presynaptic neuron
    const float *leak;         // (NPAD)
    const int *oslot;          // (NPAD) output slot or -1
    const uint32_t *in_ent;    // (WPC, EinW) (channel << 16) | target, 0xFFFFFFFF = padding
    const uint32_t *inmask;    // (NPAD, 4) input-channel bit mask per neuron (INMODE 2, 3: C <= 128), or null
    const uint8_t *inperm;     // (C) bit position of channel c in the input bit row (INMODE 3), or null (position c)
    int n_keys;
    int key_ids[8];
    float *features;           // (B, n_keys * n_out)
    uint8_t *spike_matrix;     // (B, T, N) or null
    float *v_trace;            // (B, T, N) or null
    int32_t *stats;            // (B, 2) {neurons that fired at least once, spikes of the whole reservoir} or null
    const int32_t *order;      // (B) clip of workgroup g, or null (g): lsm_reservoir_run_ordered starts long clips first
};

// INMODE: how the input drive m_i(t) = #{active channels feeding neuron i} is formed.
//   0  the wave's (channel, target) entries are streamed from global memory, integer LDS atomics count
//   1  the entries sit in registers (<= IN_REG_SLOTS*64 per wave), integer LDS atomics count
//   2  every neuron holds the bit mask of its input channels in registers (C <= 128, SL <= 4) and counts
//      popcount(mask & row) against the step's wave-uniform input bit row: no atomics, no count array,
//      and nothing to wait for between the recurrent rows and the update
//   3  as 2, with the channels' bit positions permuted (DenseArgs::inperm, chosen by the host) so that the channels
//      feeding one neuron differ in their position mod 32: the four masked words of a neuron are disjoint and ONE
//      popcount of their union counts them -- 5 vector instructions per neuron and step instead of 9
// (Ring-like reservoirs whose dense table no longer fits the caches run on lif_ring.h instead: window + list rows.)
// LSM_DENSE_MAX_VGPR: register cap of the kernel (0 = the compiler's choice).  Inside the pipeline a reservoir wave
// shares its SIMD with a front-end wave of 160-168 registers: at <= 112 registers THREE reservoir workgroups fit beside
// it (168 + 3 x 112 <= 512), at the compiler's 113 (allocated in steps of 8: 120) only two.
#ifndef LSM_DENSE_MAX_VGPR
#define LSM_DENSE_MAX_VGPR 0
#endif
#if LSM_DENSE_MAX_VGPR
#define LSM_DENSE_VGPR_ATTR __attribute__((amdgpu_waves_per_eu(512 / LSM_DENSE_MAX_VGPR)))
#else
#define LSM_DENSE_VGPR_ATTR
#endif
// REFM: the refractory countdown of the reference's period (REFRACTORY_PERIOD = 2, extract_lsm_features.py:13) lives in
// SCALAR registers -- two 64-bit lane masks per slot, h2 = "fired at the last step" and h1 = "fired the step before" --
// instead of one vector register per neuron.  The fire condition comes out of v_cmp as a lane mask, `held` = h1 | h2
// and the countdown (h1 <- h2, h2 <- fire) are scalar mask moves, the potential's reset is ONE v_cndmask on the mask,
// and the fire path takes the masks as execution masks: per neuron and step 2 vector instructions for threshold /
// reset / refractory instead of ~9 (three compares, the boolean materialised and compared again for the ballot, two
// selects, decrement, select).  The arithmetic on v is untouched.  Inside the pipeline the chip is bound by
// vector-instruction issue, and scalar instructions issue beside it.  Other periods keep the vector
// countdown (REFM = false).
template <int SL, int WPC, int INMODE, bool REFM>
__global__ __launch_bounds__(WPC * 64) LSM_DENSE_VGPR_ATTR void lif_dense_kernel(const DenseArgs a)
{
    constexpr bool INREG = INMODE == 1;
    constexpr bool INMASK = INMODE >= 2;
    constexpr bool INCOL = INMODE == 3;
    constexpr int NPW = SL * 64;
    constexpr int NPAD = NPW * WPC;
    constexpr int NT = WPC * 64;
    constexpr int R = 64 / WPC;            // fixed-region list entries per producer wave
#ifndef LSM_DENSE_G4
#define LSM_DENSE_G4 8
#endif
    // rows in flight per group (<= 32 registers of weights; LSM_DENSE_G4: the SL = 4 value, see LSM_DENSE_MAX_VGPR's note)
    constexpr int G = SL == 1 ? 16 : (SL == 4 ? LSM_DENSE_G4 : 32 / SL);
    constexpr bool FEATREG = SL <= 4;           // feature accumulators in registers (4 per neuron) instead of LDS

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *icnt = reinterpret_cast<uint32_t *>(smem);                              // NPAD
    uint16_t *wlist = reinterpret_cast<uint16_t *>(icnt + NPAD);                      // 2*NPAD
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(wlist + 2 * NPAD);                  // 2*16 counts
    uint16_t *flist = reinterpret_cast<uint16_t *>(wcnt + 64);                        // 2*64 fixed-region list
    uint4 *feat = reinterpret_cast<uint4 *>(wcnt + 128);                              // n_out
    uint32_t *bits = reinterpret_cast<uint32_t *>(feat + a.n_out);                    // T*CW

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;      // wave-uniform
    const int N = a.N, T = a.T, CW = a.CW;

    // ---- prologue: zero LDS state, bit-pack the clip's raster time-major ----
    for (int i = tid; i < NPAD; i += NT) icnt[i] = 0u;
    if (tid < 64) wcnt[tid] = 0u;
    for (int i = tid; i < a.n_out; i += NT) feat[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < T * CW; i += NT) bits[i] = 0u;
    __syncthreads();
    {
        const uint8_t *clip = a.raster + (size_t)b * a.C * T;
        if ((T & 3) == 0) {
            const uint32_t *clip4 = reinterpret_cast<const uint32_t *>(clip);
            const int nd = a.C * T / 4;
            for (int q = tid; q < nd; q += NT) {
                const uint32_t v = (LSM_ABLATE & 64) ? __builtin_nontemporal_load(clip4 + q) : clip4[q];
                if (v == 0) continue;
                const int c = (q * 4) / T;
                const int t0 = (q * 4) - c * T;
                const int pc = INCOL ? (int)a.inperm[c] : c;        // the channel's place in the bit row
                const uint32_t bit = 1u << (pc & 31);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((v >> (8 * k)) & 0xFFu) atomicOr(&bits[(t0 + k) * CW + (pc >> 5)], bit);
            }
        } else {
            const int nb = a.C * T;
            for (int q = tid; q < nb; q += NT)
                if (clip[q]) {
                    const int c = q / T;
                    const int pc = INCOL ? (int)a.inperm[c] : c;
                    atomicOr(&bits[(q - c * T) * CW + (pc >> 5)], 1u << (pc & 31));
                }
        }
    }

    float v[SL], lam[SL];
    int ref[REFM ? 1 : SL], os[SL];
    unsigned long long h1[REFM ? SL : 1], h2[REFM ? SL : 1];     // REFM: lanes whose countdown stands at 1 / at 2
#pragma unroll
    for (int r = 0; r < SL; ++r) {
        const int i = (w * SL + r) * 64 + lane;
        v[r] = 0.0f;
        if (REFM) { h1[r] = 0ull; h2[r] = 0ull; }
        else ref[r] = 0;
        lam[r] = a.leak[i];
        os[r] = a.oslot[i];
    }

    uint32_t im[SL][4];                 // INMASK: channels 0..127 feeding my neuron r
    if (INMASK) {
#pragma unroll
        for (int r = 0; r < SL; ++r) {
            const uint4 m = reinterpret_cast<const uint4 *>(a.inmask)[(w * SL + r) * 64 + lane];
            im[r][0] = m.x; im[r][1] = m.y; im[r][2] = m.z; im[r][3] = m.w;
        }
    }
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
            in_tgt[q] = ok ? (x & 0xFFFFu) : (uint32_t)(w * NPW + lane);   // padding: own slot, adds 0
        }
    }
    const float theta = a.theta, w_in = a.w_in;
    const uint32_t *my_ent = a.in_ent + (size_t)w * a.EinW;
    const bool trace = a.spike_matrix != nullptr || a.v_trace != nullptr;
    // weight of presynaptic j onto my target r: byte offset j*ld*4 (scalar, 32 bits are enough:
    // N*ld*4 <= 2^28 for N <= 8192) + my lane's byte offset (vector) + r*256 (immediate)
    // (a buffer load takes the row offset as its scalar operand: no per-row vector address arithmetic)
    const char *wt_bytes = reinterpret_cast<const char *>(a.wt);
    const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.wt), 0, (int)((uint32_t)N * (uint32_t)a.ld * 4u), 0x00020000);
    const uint32_t ld_bytes = (uint32_t)a.ld * 4u;
    const uint32_t lane_off = (uint32_t)(w * NPW + lane) * 4u;
    uint4 fr[FEATREG ? SL : 1];        // FEATREG: {n | bursts << 16, first | last << 16, sum t, sum isi^2} per neuron
#pragma unroll
    for (int r = 0; r < (FEATREG ? SL : 1); ++r) fr[r] = make_uint4(0, 0, 0, 0);
    uint32_t hf = 0u;                  // bit r: my neuron r fired at least once (stats)
    uint32_t tot_spk = 0u;             // spikes of my wave (stats)
    __syncthreads();

    // input drive of step `ts`: count the active channels feeding each target (integer atomics)
    auto input_drive = [&](int ts) {
        if (INMASK || (LSM_ABLATE & 2)) return;
        const uint32_t *row = bits + ts * CW;
        if (INREG) {
#pragma unroll
            for (int q = 0; q < IN_REG_SLOTS; q += 2) {
                if (q * 64 < a.EinW) {          // two slots at a time, only as many as the map needs
                    const uint32_t w0 = row[in_word[q]], w1 = row[in_word[q + 1]];
                    atomicAdd(icnt + in_tgt[q], (w0 & in_mask[q]) ? 1u : 0u);
                    if ((q + 1) * 64 < a.EinW) atomicAdd(icnt + in_tgt[q + 1], (w1 & in_mask[q + 1]) ? 1u : 0u);
                }
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

#if LSM_STAMP                         // diagnostic builds only: per-phase s_memtime sums of wave 0 of each clip
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
    const unsigned long long st_loop0 = st_last, st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = 0; t < T; ++t) {
        // The step list read and the row fetch are the latency-critical part of a step: they issue at
        // raised priority so that waves of other kernels sharing the SIMD (the float64 filterbank in the
        // pipeline) do not delay the loads; the update below runs at normal priority in their stall slots.
        // Measured inside the pipeline: launch duration 1.31 -> 1.11 ms at unchanged throughput (raising
        // the priority for the whole step gives 0.67 ms but costs 4 % of the pipeline's throughput).
#ifndef LSM_LIF_NO_PRIO                  // diagnostic builds: same-box A/B of the priority phases
        __builtin_amdgcn_s_setprio(1);
#endif
        const int cur = t & 1, prv = cur ^ 1;
        const uint16_t *list_prev = wlist + prv * NPAD;
        uint16_t *list_cur = wlist + cur * NPAD + w * NPW;

        float cin[SL];
#pragma unroll
        for (int r = 0; r < SL; ++r) cin[r] = 0.0f;
        uint32_t rowbits[4] = {0u, 0u, 0u, 0u};          // INMASK: this step's input bit row (wave-uniform)
        if (INMASK && !(LSM_ABLATE & 2)) {
            if (CW == 4) {
                const uint4 q4 = *reinterpret_cast<const uint4 *>(bits + t * 4);
                rowbits[0] = q4.x; rowbits[1] = q4.y; rowbits[2] = q4.z; rowbits[3] = q4.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) rowbits[q] = q < CW ? bits[t * CW + q] : 0u;
            }
        }

        // `todo`: filled entries of the step list; lane l holds entry l's neuron in `jl`.  Up to G rows
        // are in flight; nested "one more?" tests make a group cost one taken branch.
        bool drove = false;
        auto add_rows = [&](unsigned long long todo, uint32_t jl) {
            if (LSM_ABLATE & 1) todo = 0ull;
            if (LSM_ABLATE & 4) jl = min(jl, (uint32_t)(N - 1));
            // entry -> byte offset of its row, once per 64 entries (N*ld*4 <= 2^28 for N <= 8192)
            const uint32_t jo = ((LSM_ABLATE & 32) ? (jl & 255u) : jl) * ld_bytes;
            while (todo != 0ull) {
                const int n8 = min((int)__builtin_popcountll(todo), G);
                float wv[16][SL];                   // only the first G rows are ever live
#define LSM_LD(k)                                                                   \
    {                                                                               \
        const int sk = __builtin_ctzll(todo);                                       \
        const uint32_t j = __builtin_amdgcn_readlane(jo, sk);                       \
        if (LSM_DENSE_BITSET) asm volatile("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(sk)); \
        else todo &= todo - 1ull;                                                   \
        {                                                                           \
            const int so = (int)j;                                                  \
            _Pragma("unroll") for (int r = 0; r < SL; ++r)                          \
                wv[k][r] = (LSM_ABLATE & 16) ? __uint_as_float(j + r)               \
                         : LSM_DENSE_BUF ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(    \
                               wt_rsrc, (int)lane_off + r * 256, so, 0))            \
                         : *reinterpret_cast<const float *>(wt_bytes + j + lane_off + r * 256); \
        }                                                                           \
    }
                // the load chain tests the remaining-entries mask itself (one scalar 64-bit compare per row)
                LSM_LD(0)
                if (G > 1 && todo) { LSM_LD(1)
                if (G > 2 && todo) { LSM_LD(2)
                if (G > 3 && todo) { LSM_LD(3)
                if (G > 4 && todo) { LSM_LD(4)
                if (G > 5 && todo) { LSM_LD(5)
                if (G > 6 && todo) { LSM_LD(6)
                if (G > 7 && todo) { LSM_LD(7)
                if (G > 8 && todo) { LSM_LD(8)
                if (G > 9 && todo) { LSM_LD(9)
                if (G > 10 && todo) { LSM_LD(10)
                if (G > 11 && todo) { LSM_LD(11)
                if (G > 12 && todo) { LSM_LD(12)
                if (G > 13 && todo) { LSM_LD(13)
                if (G > 14 && todo) { LSM_LD(14)
                if (G > 15 && todo) { LSM_LD(15) } } } } } } } } } } } } } } }
#undef LSM_LD
                STAMP(1);
                if (!drove) {                     // the input counts fill the load latency
                    input_drive(t);
                    drove = true;
                }
#define LSM_ADD(k)                                                                  \
    {                                                                               \
        _Pragma("unroll") for (int r = 0; r < SL; ++r) cin[r] = cin[r] + wv[k][r];  \
    }
                LSM_ADD(0)
                if (n8 > 1) { LSM_ADD(1)
                if (n8 > 2) { LSM_ADD(2)
                if (n8 > 3) { LSM_ADD(3)
                if (n8 > 4) { LSM_ADD(4)
                if (n8 > 5) { LSM_ADD(5)
                if (n8 > 6) { LSM_ADD(6)
                if (n8 > 7) { LSM_ADD(7)
                if (n8 > 8) { LSM_ADD(8)
                if (n8 > 9) { LSM_ADD(9)
                if (n8 > 10) { LSM_ADD(10)
                if (n8 > 11) { LSM_ADD(11)
                if (n8 > 12) { LSM_ADD(12)
                if (n8 > 13) { LSM_ADD(13)
                if (n8 > 14) { LSM_ADD(14)
                if (n8 > 15) { LSM_ADD(15) } } } } } } } } } } } } } } }
#undef LSM_ADD
                STAMP(2);
            }
        };

        // ---- spiking neurons of step t-1 ----
        const uint32_t pcnt = wcnt[prv * 16 + lane / R];            // spikes of producer wave lane/R
        const uint32_t jfix = flist[prv * 64 + lane];               // its (lane%R)-th spiking neuron
        STAMP(0);
        // a producer with more than R spikes does not fit its fixed region: the counts themselves say so
        if (__ballot(pcnt > (uint32_t)R) == 0ull) {
            add_rows(__ballot((uint32_t)(lane % R) < pcnt), jfix);
        } else {
            // general path: merge the per-wave lists (scalar prefix over the counts, compare chain)
            const uint32_t cv = wcnt[prv * 16 + (lane & 15)];
            uint32_t pre[WPC + 1];
            pre[0] = 0u;
#pragma unroll
            for (int q = 0; q < WPC; ++q) pre[q + 1] = pre[q] + __builtin_amdgcn_readlane(cv, q);
            const uint32_t total = pre[WPC];
            for (uint32_t l0 = 0; l0 < total; l0 += 64) {
                const uint32_t l = l0 + lane;
                uint32_t wsel = 0u, pbase = 0u;
#pragma unroll
                for (int q = 1; q < WPC; ++q) {
                    const bool ge = l >= pre[q];
                    wsel += ge ? 1u : 0u;
                    pbase = ge ? pre[q] : pbase;
                }
                uint32_t jl = 0u;
                if (l < total) jl = list_prev[wsel * NPW + (l - pbase)];
                add_rows(__ballot(l < total), jl);
            }
        }
        if (!drove) input_drive(t);
        if (!INMASK) wave_lds_fence();

#if !defined(LSM_LIF_NO_PRIO) && !defined(LSM_LIF_PRIO_WHOLE)   // (PRIO_WHOLE: diagnostic build, raised for the whole step)
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- neuron update ----
        unsigned long long bal[SL];
        unsigned long long any_fire = 0ull;
#pragma unroll
        for (int r = 0; r < SL; ++r) {
            const int i = (w * SL + r) * 64 + lane;
            uint32_t nin;
            if (INCOL) {
                // disjoint by construction of the bit positions: the union's popcount is the sum of the four.
                // (v_and_or_b32 spelled out: the compiler forms four v_and and two v_or/v_or3 from the C expression)
                uint32_t u = im[r][0] & rowbits[0];
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(u) : "v"(im[r][q]), "s"(rowbits[q]));
                nin = __popc(u);
            } else if (INMASK) {
                nin = __popc(im[r][0] & rowbits[0]) + __popc(im[r][1] & rowbits[1]) +
                      __popc(im[r][2] & rowbits[2]) + __popc(im[r][3] & rowbits[3]);
            } else {
                nin = icnt[i];
                icnt[i] = 0u;
            }
            cin[r] = cin[r] + w_in * (float)nin;         // SPEC.md §3: input term after the recurrent sum
            const float m = lam[r] * v[r];
            const float d = v[r] - m;
            const float vn = d + cin[r];
            if (REFM) {
                const unsigned long long held = h1[r] | h2[r];
                const unsigned long long ge = __builtin_amdgcn_fcmpf(vn, theta, 3 /* ordered >= */);
                const unsigned long long fire = ge & ~held;
                v[r] = __builtin_amdgcn_inverse_ballot_w64(ge | held) ? 0.0f : vn;
                h1[r] = h2[r];                  // 2 -> 1 (and 1 -> 0: the old h1 is dropped)
                h2[r] = fire;                   // a firing neuron (never a held one) starts at the period, 2
                bal[r] = fire;
            } else {
                const bool held = ref[r] > 0;
                const bool fire = !held && (vn >= theta);
                v[r] = (held || fire) ? 0.0f : vn;
                ref[r] = held ? ref[r] - 1 : (fire ? a.refractory : 0);
                bal[r] = __ballot(fire);
            }
            any_fire |= bal[r];
        }
        STAMP(3);
        int nspk = 0;
        if (any_fire != 0ull) {                  // one branch per wave and step
#pragma unroll
            for (int r = 0; r < SL; ++r) {
                // (the mask itself becomes the execution mask: no per-lane bit test)
                const bool fire = REFM ? __builtin_amdgcn_inverse_ballot_w64(bal[r]) : (bool)((bal[r] >> lane) & 1ull);
                if (fire) {
                    const int rank = nspk + lane_rank(bal[r]);
                    const uint16_t me = (uint16_t)((w * SL + r) * 64 + lane);
                    list_cur[rank] = me;
                    if (rank < R) flist[cur * 64 + w * R + rank] = me;
                    hf |= 1u << r;
                    if ((FEATREG || os[r] >= 0) && !(LSM_ABLATE & 8)) {
                        // FEATREG: the accumulators of EVERY neuron of the lane sit in registers (no LDS round
                        // trip in the fire path, which all other waves wait for at the barrier); they are parked
                        // in the LDS array once, after the last step
                        uint4 f = FEATREG ? fr[r] : feat[os[r]];
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
                        if (FEATREG) fr[r] = f;
                        else feat[os[r]] = f;
                    }
                }
                nspk += __popcll(bal[r]);
            }
        }
        tot_spk += (uint32_t)nspk;
        if (lane == 0) wcnt[cur * 16 + w] = (uint32_t)nspk;   // > R: next step's consumers merge the lists
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
        if (!(LSM_ABLATE & 4)) __syncthreads();
        STAMP(5);
    }
#if LSM_STAMP
    st_sum[6] = st_last - st_loop0;                                   // shader cycles in the step loop
    st_sum[7] = __builtin_amdgcn_s_memrealtime() - st_real0;          // 100 MHz ticks in the step loop
    if (tid == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_lif_stamps[k], st_sum[k]);
#endif

    if (FEATREG) {
#pragma unroll
        for (int r = 0; r < SL; ++r)
            if (os[r] >= 0) feat[os[r]] = fr[r];
        __syncthreads();
    }
    // ---- epilogue: health statistics (/root/reference/extract_lsm_features.py:119-133 derives them from the
    //      (T, N) spike matrix; here they come from one flag per neuron and one count per wave), then
    //      SPEC.md §4 features from the integer accumulators (float64, then float32) ----
    if (a.stats) {
        atomicAdd(&wcnt[32], (uint32_t)__popc(hf));
        if (lane == 0) atomicAdd(&wcnt[33], tot_spk);
        __syncthreads();
        if (tid == 0) {
            a.stats[2 * b] = (int32_t)wcnt[32];
            a.stats[2 * b + 1] = (int32_t)wcnt[33];
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

typedef void (*dense_fn_t)(const DenseArgs);

// REFM needs 2 x 2 scalar registers per slot: offered up to 4 slots per lane (the layouts of N <= 4096 with >= 4
// waves, i.e. every reservoir the dense rows serve by default); fatter layouts keep the countdown in vector registers.
constexpr int DENSE_REFM_MAX_SL = 4;
constexpr int DENSE_REFM_REFRACTORY = 2;        // the one period the mask form is written for (the reference's)

template <int SL, int INMODE, bool REFM>
dense_fn_t pick_dense_wpc(int wpc)
{
    switch (wpc) {
    case 1: return lif_dense_kernel<SL, 1, INMODE, REFM>;
    case 2: return lif_dense_kernel<SL, 2, INMODE, REFM>;
    case 4: return lif_dense_kernel<SL, 4, INMODE, REFM>;
    case 8: return lif_dense_kernel<SL, 8, INMODE, REFM>;
    case 16: return lif_dense_kernel<SL, 16, INMODE, REFM>;
    default: return nullptr;
    }
}

template <int INMODE>
dense_fn_t pick_dense_sl(int sl, int wpc, bool refm)
{
    if (refm && sl <= DENSE_REFM_MAX_SL) {
        switch (sl) {
        case 1: return pick_dense_wpc<1, INMODE, true>(wpc);
        case 2: return pick_dense_wpc<2, INMODE, true>(wpc);
        case 4: return pick_dense_wpc<4, INMODE, true>(wpc);
        default: return nullptr;
        }
    }
    switch (sl) {
    case 1: return pick_dense_wpc<1, INMODE, false>(wpc);
    case 2: return pick_dense_wpc<2, INMODE, false>(wpc);
    case 4: return pick_dense_wpc<4, INMODE, false>(wpc);
    case 8: if (INMODE >= 2) return nullptr; else return pick_dense_wpc<INMODE >= 2 ? 4 : 8, INMODE, false>(wpc);
    case 16: if (INMODE >= 2) return nullptr; else return pick_dense_wpc<INMODE >= 2 ? 4 : 16, INMODE, false>(wpc);
    default: return nullptr;
    }
}

// refm: the caller checked a.refractory == DENSE_REFM_REFRACTORY (else the countdown stays in vector registers)
dense_fn_t pick_dense_0(int sl, int wpc, bool refm);      // lif_dense_0.hip (INMODE 0: entries from global memory)
dense_fn_t pick_dense_1(int sl, int wpc, bool refm);      // lif_dense_1.hip (INMODE 1: entries in registers)
dense_fn_t pick_dense_2(int sl, int wpc, bool refm);      // lif_dense_2.hip (INMODE 2: channel masks, C <= 128, SL <= 4)
dense_fn_t pick_dense_3(int sl, int wpc, bool refm);      // lif_dense_3.hip (INMODE 3: channel masks at coloured positions)
#if LSM_STAMP
int read_lif_stamps_d2(unsigned long long *o, int r);   // stamps of the INMODE-2 unit
int read_lif_stamps_d3(unsigned long long *o, int r);   // stamps of the INMODE-3 unit (the cfg2 kernel)
#endif

}  // namespace lsm_lif
