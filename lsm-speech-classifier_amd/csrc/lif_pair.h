// lif_pair.h -- LIF reservoir time loop on ring rows, PAIR blocks (small-world reservoirs whose ring window is about as
// wide as the clip's waves times 128 neurons: N = 4000, k = 800), gfx950.
//
// Same contract as lif_kernel.h / lif_dense.h / lif_ring.h (SPEC.md §3-§4; replaces reset / set_input_spike_times /
// simulate / extract_features_from_spikes of /root/reference/extract_lsm_features.py:79-83) and the same ring-row
// table (dense window of row j from its 128-byte-aligned start + a list of the synapses outside it).  What differs
// from lif_ring.h is how a row is shared out and what a row costs a wave (round 5, profiles/r05_ring_issue_ports.txt:
// the quad kernel's row loop is bound by SCALAR-instruction issue -- 24 scalar + 12 vector instructions per row and
// wave, most of them building two buffer descriptors -- and, behind that, by the LDS store path, on which three of the
// eight waves of a clip rewrite a 1 KB accumulator quad that the row's window does not reach):
//   * a lane owns TWO consecutive neurons of a 128-neuron block, a wave owns the blocks w, w+WPC, w+2*WPC, ...; a
//     window of 2H+1 <= (WPC-1)*128 targets then puts exactly one block into EVERY wave: all waves fetch 512 useful
//     bytes per row (8 bytes per lane) and rewrite 512 bytes of accumulators (ds_read_b64 + ds_write_b64: 8 LDS cycles
//     against 17 for a quad, none of them on zeros);
//   * everything a wave needs for row j is ONE 16-byte record built by the host, rec[j*WPC + w] = {byte offset of the
//     row's address, my block's offset in the row | its accumulators' LDS offset, bytes that exist | bytes of
//     my list, my list's address}: the chunk set-up is one 16-byte load per lane (= per row), and a row is six
//     v_readlane + two scalar instructions (the quad kernel derived all of it from j, per row, on the scalar unit);
//   * the merged order of a step's spikes (32 block lists instead of 16 quad lists) comes from two DPP scans and one
//     pass through a 64-word LDS scratch instead of a compare chain over the blocks;
//   * chunks of 64 rows, four in flight, and exactly the chunk's rows are requested: a buffer load whose descriptor
//     has no bytes still holds the texture-address path ~11 cycles (profiles/r05_residency_ablation.txt,
//     r05_pair_rows_in_flight.txt).
// What bounds it (profiles/r05_pair_rmw_ablation.txt): waiting for the rows' data at the start of every step and the
// address-path time of every request -- not the accumulator read-modify-write through LDS, not residency.
// Arithmetic, order of the float32 additions (rows ascending, a target gets a row's weight from the window or from
// the list, never both, the other term is +0.0) and results are those of lif_ring.h: bit-identical to the oracle.
// The kernel exists for the input drive counted from per-neuron channel masks (C <= 128: the reference's 128 filters), with
// one leak coefficient for every neuron (its default) or one per neuron in registers (LEAKV: --leak-variance-divisor);
// other reservoirs keep lif_ring.h.
#pragma once
#include "lif_kernel.h"

namespace lsm_lif {

#ifndef LSM_PAIR_P
#define LSM_PAIR_P 4        // rows in flight (5 registers each; 6 rows: the lone launch 1.5 % faster, the whole path 3.5 % slower:
                            // profiles/r05_pair_rows_in_flight.txt)
#endif
#ifndef LSM_PAIR_PRE
#define LSM_PAIR_PRE 0      // where a step's input counts and leak terms (they do not depend on the recurrent sums) are computed: 0 = in the
#endif                      // update; 1 = while the first chunk's row records are fetched; 2 = while its first rows are fetched (both 1-2 %
                            // SLOWER at cfg4: the other waves of the CU fill those waits already; profiles/r05_pair_rows_in_flight.txt)
#ifndef LSM_PAIR_PRIO
#define LSM_PAIR_PRIO 1     // wave priority of the step loop (as lif_ring.h: profiles/r04_ring_priority.txt)
#endif
#ifndef LSM_PAIR_ABLATE
#define LSM_PAIR_ABLATE 0   // diagnostic builds only (WRONG results): 1 = no window loads, 2 = no accumulator read-modify-write,
#endif                      // 8 = no list loads, 32 = no feature updates, 64 = no accumulator reads, 128 = the window piece summed in a
                            // register (no block switches), 256 = the list entries summed in a register (profiles/r05_pair_rmw_ablation.txt)

#ifndef LSM_PAIR_PHASES
#define LSM_PAIR_PHASES 0   // diagnostic builds only: every wave sums the core-clock cycles of its step phases and writes them OVER
#endif                      // the feature rows (exp/r03_ring_phases.py reads them back), as LSM_RING_PHASES of lif_ring.h
#if LSM_PAIR_PHASES
#define LSM_PAIR_MARK(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); ph_[k] += (uint32_t)(now_ - last_); last_ = now_; }
#else
#define LSM_PAIR_MARK(k)
#endif

// Diagnostic builds only (results stay right): extra work per row and wave, to see which port the row loop is bound by
// (profiles/r05_pair_issue_ports.txt): scalar / vector instructions, LDS stores to the lane's own dump word, buffer loads
// whose descriptor holds zero bytes (they pass through the texture-address unit and touch no memory).
#ifndef LSM_PAIR_DUMMY_SALU
#define LSM_PAIR_DUMMY_SALU 0
#endif
#ifndef LSM_PAIR_DUMMY_VALU
#define LSM_PAIR_DUMMY_VALU 0
#endif
#ifndef LSM_PAIR_DUMMY_LDS
#define LSM_PAIR_DUMMY_LDS 0
#endif
#ifndef LSM_PAIR_DUMMY_VMEM
#define LSM_PAIR_DUMMY_VMEM 0
#endif
#ifndef LSM_PAIR_DUMMY_VMEM_LANES
#define LSM_PAIR_DUMMY_VMEM_LANES 64
#endif
#ifndef LSM_PAIR_OWN_DUMP
#define LSM_PAIR_OWN_DUMP 0     // 1: a lane without a list entry adds its zero to a dump word of its own (lane*4), not to word 0
#endif
// Time-only ablation builds (VERDICT r4 #1a: "predict before building"; exp/r05_residency.py).  REPLAY: the update takes every
// neuron's spike from a RECORDED spike matrix (the `spike_matrix` argument, read instead of written: a correct run's
// output), so the rows of every step are those of the real run whatever else the build leaves out; features are garbage.
// LEAN (with REPLAY): no input masks, no refractory/slot register, no feature records -- the registers and the LDS a
// kernel would have with the input counts precomputed and the features accumulated outside LDS.
#ifndef LSM_PAIR_REPLAY
#define LSM_PAIR_REPLAY 0
#endif
#ifndef LSM_PAIR_LEAN
#define LSM_PAIR_LEAN 0
#endif
#ifndef LSM_PAIR_WAVES_PER_EU
#define LSM_PAIR_WAVES_PER_EU 4
#endif
#ifndef LSM_PAIR_LDS_PAD
#define LSM_PAIR_LDS_PAD 0      // extra LDS bytes per clip (host side): pins the clips per compute unit of an ablation build
#endif
#define LSM_PAIR_DUMMIES (LSM_PAIR_DUMMY_SALU || LSM_PAIR_DUMMY_VALU || LSM_PAIR_DUMMY_LDS || LSM_PAIR_DUMMY_VMEM)

struct PairArgs {
    int N, C, T, B;
    int n_out, CW, refractory, burst_isi_max;
    float theta, w_in, leak_u;
    const uint8_t *raster;     // (B, C, T) uint8
    const float *band;         // ring windows, row j at byte j*pitch (the table of lif_ring.h)
    const uint4 *rec;          // (N*WPC) row records (pair_record), built for the addresses of `band` and `rem`
    const uint2 *rem;          // list entries {LDS byte offset of the target's accumulator, weight bits}, (row, wave) major
    const float *leak;         // LEAKV: (NPAD) leak coefficient per neuron, neuron order (else null: leak_u for every neuron)
    const int *oslot;          // (NPAD) output slot or -1, neuron order
    const uint32_t *inmask;    // (NPAD, 4) input-channel bit mask per neuron
    const uint8_t *inperm;     // coloured positions: (C) bit position of channel c in the input bit row, else null
    int n_keys;
    int key_ids[8];
    float *features;           // (B, n_keys * n_out)
    uint8_t *spike_matrix;     // (B, T, N) or null
    float *v_trace;            // (B, T, N) or null
    int32_t *stats;            // (B, 2) or null
    const int32_t *order;      // (B) clip of workgroup g, or null
};

typedef float pair_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t pair_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t pair_u4 __attribute__((ext_vector_type(4)));
// LDS accesses of the row loop by ABSOLUTE byte address (the kernel has no static LDS: its dynamic LDS starts at 0, checked
// once per launch): `smem + offset` costs a vector add per access that the compiler does not fold.
typedef __attribute__((address_space(3))) pair_f2 pair_lds_f2;
typedef __attribute__((address_space(3))) float pair_lds_f1;
#define LSM_PAIR_LDS_F2(addr) (*reinterpret_cast<pair_lds_f2 *>((uintptr_t)(uint32_t)(addr)))
#define LSM_PAIR_LDS_F1(addr) (*reinterpret_cast<pair_lds_f1 *>((uintptr_t)(uint32_t)(addr)))

constexpr int PAIR_DUMP_BYTES = 256;                // LDS bytes 0..255: word 0 is where a lane without a list entry adds its zero
constexpr int PAIR_MAX_BLOCKS = 64;                 // 8192 neurons
constexpr int PAIR_CHUNK = 64;                      // rows per chunk: lane m holds the record of the chunk's row m
constexpr int PAIR_WCNT_WORDS = 2 * PAIR_MAX_BLOCKS + 8;

// LDS byte offset (from the start of LDS) of neuron i's float32 accumulator
__host__ __device__ inline uint32_t pair_acc_byte(int i) { return (uint32_t)PAIR_DUMP_BYTES + (uint32_t)i * 4u; }
// owner wave / register block of block g
__host__ __device__ inline int pair_wave_of_block(int g, int wpc) { return g % wpc; }
// Row record of (row j, wave w) -- see the head of this file.  row_lo / list_lo: low 32 bits of the DEVICE address of the
// row's window / of the wave's list (the tables do not cross a 4 GB line: the kernel takes the high bits from the table
// pointers, so a buffer descriptor costs no address arithmetic).  so: byte offset of the wave's block inside the stored row
// (negative for the lanes of the first block that lie in front of the row's 128-byte-aligned start: their unsigned
// offsets are out of range); gb: the block; nbytes: bytes of the row that exist; list_first / list_n: the wave's list.
__host__ __device__ inline uint4 pair_record(uint32_t row_lo, int so, int gb, uint32_t nbytes, uint32_t list_lo,
                                             uint32_t list_n)
{
    uint4 r;
    r.x = row_lo;
    r.y = ((uint32_t)so & 0xFFFFu) | ((uint32_t)(gb * 512) << 16);
    r.z = nbytes | ((list_n * 8u) << 16);
    r.w = list_lo;
    return r;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t pair_dpp0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// inclusive scans over the 64 lanes (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast 15 / 31 across them)
__device__ __forceinline__ uint32_t pair_scan_add(uint32_t v)
{
    v += pair_dpp0<0x111, 0xF>(v);
    v += pair_dpp0<0x112, 0xF>(v);
    v += pair_dpp0<0x114, 0xF>(v);
    v += pair_dpp0<0x118, 0xF>(v);
    v += pair_dpp0<0x142, 0xA>(v);
    v += pair_dpp0<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint32_t pair_scan_max(uint32_t v)
{
    v = max(v, pair_dpp0<0x111, 0xF>(v));
    v = max(v, pair_dpp0<0x112, 0xF>(v));
    v = max(v, pair_dpp0<0x114, 0xF>(v));
    v = max(v, pair_dpp0<0x118, 0xF>(v));
    v = max(v, pair_dpp0<0x142, 0xA>(v));
    v = max(v, pair_dpp0<0x143, 0xC>(v));
    return v;
}

// SPEC.md 4: one feature of an output neuron from its exact integers (n spikes, first / last spike time, S1 = sum of the
// spike times, Q = sum of the squared inter-spike intervals, bursts), evaluated in float64 and rounded to float32.
__device__ __forceinline__ float pair_feature_value(int key, int n, int bursts, int first, int last, uint32_t s1, uint32_t q2,
                                                    int T)
{
    double val = 0.0;
    switch (key) {
    case 0: val = (double)n; break;
    case 1: { const double p = (double)n / (double)T; val = p * (1.0 - p); } break;
    case 2: val = n >= 1 ? (double)s1 / (double)n : 0.0; break;
    case 3: val = n >= 1 ? (double)first : 0.0; break;
    case 4: val = n >= 1 ? (double)last : 0.0; break;
    case 5: val = n >= 2 ? (double)(last - first) / (double)(n - 1) : 0.0; break;
    case 6:
        if (n >= 2) {
            const double m = (double)(last - first) / (double)(n - 1);
            val = (double)q2 / (double)(n - 1) - m * m;
        }
        break;
    default: val = (double)bursts; break;
    }
    return (float)val;
}

// spiking input channels of one neuron in this step: its channel masks against the step's input bit row
template <int INMASK>
__device__ __forceinline__ uint32_t pair_input_count(const uint32_t (&im)[4], const uint32_t (&rowbits)[4])
{
    if (INMASK == 2) {
        // disjoint by construction of the bit positions: one popcount of the union (lif_dense.h, INMODE 3)
        uint32_t u = im[0] & rowbits[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(u) : "v"(im[k]), "v"(rowbits[k]));
        return __popc(u);
    }
    return __popc(im[0] & rowbits[0]) + __popc(im[1] & rowbits[1]) + __popc(im[2] & rowbits[2]) + __popc(im[3] & rowbits[3]);
}

// BL: blocks (128 neurons, 2 per lane) per wave; WPC: waves per clip; INMASK: 1 = natural bit positions of the input
// channels, 2 = coloured positions (lif_dense.h, INMODE 3).
// LEAKV: a leak coefficient per neuron in registers (the reference's --leak-variance-divisor, extract_lsm_features.py:174,
// 182-183) instead of one for all (its default).
template <int BL, int WPC, int INMASK, bool LEAKV = false>
__global__ __launch_bounds__(WPC * 64) __attribute__((amdgpu_waves_per_eu(LSM_PAIR_WAVES_PER_EU)))
void lif_pair_kernel(const PairArgs a)
{
    constexpr int SL = 2 * BL;
    constexpr int NBP = BL * WPC;                   // blocks of the padded layout
    constexpr int NPAD = NBP * 128;
    constexpr int NT = WPC * 64;
    constexpr int P = LSM_PAIR_P;
    constexpr int CH = PAIR_CHUNK;
    constexpr uint32_t RSRC_FLAGS = 0x00020000u;    // raw dword buffer, gfx9 family
    static_assert(NBP <= PAIR_MAX_BLOCKS, "at most 8192 neurons");
    static_assert(P >= 2 && P <= 8 && CH <= 64, "rows in flight; a chunk's records live in the 64 lanes");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *marks = reinterpret_cast<uint32_t *>(smem + PAIR_DUMP_BYTES + NPAD * 4);    // 64 words per wave
    uint8_t *wlist = reinterpret_cast<uint8_t *>(marks + WPC * 64);                       // 2*NPAD: 128 per block
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(wlist + 2 * NPAD);                      // 2*64 block counts + stats
    uint4 *feat = reinterpret_cast<uint4 *>(wcnt + PAIR_WCNT_WORDS);                      // n_out (none in a LEAN ablation build)
    uint32_t *bits = reinterpret_cast<uint32_t *>(feat + (LSM_PAIR_LEAN ? 0 : a.n_out)); // T*CW

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;      // wave-uniform
    const int N = a.N, T = a.T, CW = a.CW;
#define LSM_PAIR_GB(q) ((q) * WPC + w)              // global block of my register block q

    // the row loop addresses LDS absolutely (LSM_PAIR_LDS_F2): the dynamic LDS must start at 0
    if ((uintptr_t)((__attribute__((address_space(3))) unsigned char *)smem) != 0) __builtin_trap();

    // ---- prologue: zero LDS state, bit-pack the clip's raster time-major ----
    for (int i = tid; i < (PAIR_DUMP_BYTES + NPAD * 4 + WPC * 256) / 4; i += NT) reinterpret_cast<uint32_t *>(smem)[i] = 0u;
    for (int i = tid; i < PAIR_WCNT_WORDS; i += NT) wcnt[i] = 0u;
    if (!LSM_PAIR_LEAN)
        for (int i = tid; i < a.n_out; i += NT) feat[i] = make_uint4(0, 0, 0, 0);
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
                const int pc = INMASK == 2 ? (int)a.inperm[c] : c;      // the channel's place in the bit row
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
                    const int pc = INMASK == 2 ? (int)a.inperm[c] : c;
                    atomicOr(&bits[(q - c * T) * CW + (pc >> 5)], 1u << (pc & 31));
                }
        }
    }

    // my neurons: register r = 2*q + h  <->  neuron GB(q)*128 + lane*2 + h
    // oref[r] = (output slot + 1) | (refractory countdown << 16), as in lif_ring.h
    float v[SL];
    float lam[LEAKV ? SL : 1];
    uint32_t oref[LSM_PAIR_LEAN ? 1 : SL];
    uint32_t im[LSM_PAIR_LEAN ? 1 : SL][4];         // channels 0..127 feeding my neuron r
    const float lam_u = a.leak_u;
#pragma unroll
    for (int q = 0; q < BL; ++q) {
        const int i0 = LSM_PAIR_GB(q) * 128 + lane * 2;
        const int2 o2 = *reinterpret_cast<const int2 *>(a.oslot + i0);
        const int o[2] = {o2.x, o2.y};
        if (LEAKV) {
            const float2 l2 = *reinterpret_cast<const float2 *>(a.leak + i0);
            lam[2 * q] = l2.x; lam[2 * q + 1] = l2.y;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // padding neurons (i >= N) start with a NaN potential: it stays NaN through every update and never fires
            // (a window that wraps past the ring's end may deliver weights of real blocks to them)
            v[2 * q + h] = (i0 + h) < N ? 0.0f : __builtin_nanf("");
            if (!LSM_PAIR_LEAN) {
                oref[2 * q + h] = (uint32_t)(o[h] + 1);
                const uint4 m = reinterpret_cast<const uint4 *>(a.inmask)[i0 + h];
                im[2 * q + h][0] = m.x; im[2 * q + h][1] = m.y; im[2 * q + h][2] = m.z; im[2 * q + h][3] = m.w;
            }
        }
    }
    uint32_t ref_set_v = (uint32_t)a.refractory << 16;             // (vector registers: a VOP3 select takes no literal)
    uint32_t minus_one_v = 0xFFFF0000u;
    asm volatile("" : "+v"(ref_set_v), "+v"(minus_one_v));
    const float theta = a.theta, w_in = a.w_in;
    const bool trace = !LSM_PAIR_REPLAY && (a.spike_matrix != nullptr || a.v_trace != nullptr);
#if LSM_PAIR_REPLAY
    const uint8_t *rec_row = a.spike_matrix + (size_t)b * a.T * a.N;      // recorded spikes of step t (read, not written)
    unsigned long long never64 = 0ull;
    asm volatile("" : "+s"(never64));
#endif
    // row t of the clip's optional (T, N) outputs: advanced by N per step (the per-block form of the address, a 64-bit
    // product, was computed by every block of every step, traced or not)
    uint8_t *sm_row = (a.spike_matrix && !LSM_PAIR_REPLAY) ? a.spike_matrix + (size_t)b * T * N : nullptr;
    float *vt_row = a.v_trace ? a.v_trace + (size_t)b * T * N : nullptr;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t accl = (uint32_t)PAIR_DUMP_BYTES + lane8;       // my pair of block g: accl + g*512
    // the records hold the LOW 32 address bits of a row / a list; neither table crosses a 4 GB line (host-checked)
    const uint64_t band_hi = reinterpret_cast<uint64_t>(a.band) & 0xFFFFFFFF00000000ull;
    const uint64_t rem_hi = reinterpret_cast<uint64_t>(a.rem) & 0xFFFFFFFF00000000ull;
    const uint4 *my_rec = a.rec + w;                               // record of row j: my_rec[j*WPC]
    uint32_t *mymarks = marks + w * 64;
    uint32_t hf = 0u;                  // bit r: my neuron r fired at least once (stats)
    uint32_t tot_spk = 0u;             // spikes of my wave (stats)
#if LSM_PAIR_PHASES
    uint32_t ph_[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t rows_ = 0u;
    uint64_t last_ = 0;
#endif
    __syncthreads();
#if LSM_PAIR_PHASES
    last_ = __builtin_amdgcn_s_memtime();
#endif

    // The part of a step's update that does not wait for the recurrent sums -- the count of spiking input channels of every
    // neuron and the leak d = v - lambda * v -- is computed while the step's first loads are in the air (LSM_PAIR_PRE); the
    // update then forms sum + w_in * count and d + that: the same operations in the same order (SPEC.md 3).
    uint32_t npk[(LSM_PAIR_PRE && !LSM_PAIR_LEAN) ? (SL + 3) / 4 : 1];     // input counts (<= 128 channels), a byte per neuron
#define LSM_PAIR_PRE_UPDATE                                                                             \
    {                                                                                                   \
        uint32_t rowbits[4];                                    /* this step's input bit row (wave-uniform) */ \
        if (CW == 4) {                                                                                  \
            const uint4 q4 = *reinterpret_cast<const uint4 *>(bits + t * 4);                            \
            rowbits[0] = q4.x; rowbits[1] = q4.y; rowbits[2] = q4.z; rowbits[3] = q4.w;                 \
        } else {                                                                                        \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) rowbits[k] = k < CW ? bits[t * CW + k] : 0u;  \
        }                                                                                               \
        _Pragma("unroll") for (int r = 0; r < SL; ++r) {                                                \
            const uint32_t nn_ = pair_input_count<INMASK>(im[r], rowbits);                              \
            npk[r >> 2] = (r & 3) ? (npk[r >> 2] | (nn_ << (8 * (r & 3)))) : nn_;                       \
            const float m_ = (LEAKV ? lam[r] : lam_u) * v[r];                                           \
            v[r] = v[r] - m_;                                                                           \
        }                                                                                               \
    }

    if (LSM_PAIR_PRIO) __builtin_amdgcn_s_setprio(LSM_PAIR_PRIO);
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        const uint8_t *list_prev = wlist + prv * NPAD;
        uint8_t *list_cur = wlist + cur * NPAD;

        // ---- spiking neurons of step t-1: block g holds cv[g] of them, ascending; lane l <- l-th neuron of the clip ----
        const uint32_t cv = wcnt[prv * PAIR_MAX_BLOCKS + lane];             // (blocks past NBP: never written, zero)
        const uint32_t inc = pair_scan_add(cv);
        const uint32_t exc = inc - cv;
        const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
        LSM_PAIR_MARK(0)               // block counts read, scanned
#if LSM_PAIR_PHASES
        rows_ += total;
#endif

        for (uint32_t l0 = 0; l0 < total; l0 += CH) {
            // Block g's spikes sit at positions exc[g] .. inc[g]-1 of the merged order.  Every non-empty block that reaches
            // into this chunk leaves {g, exc[g]} at its first position inside the chunk; a running maximum over the lanes
            // (both fields ascend with g) then tells every lane its block.
            const bool mk = cv != 0u && inc > l0 && exc < l0 + (uint32_t)CH;
            if (mk) mymarks[max(exc, l0) - l0] = ((uint32_t)lane << 16) | exc;
            wave_lds_fence();
            uint32_t mv = mymarks[lane];
            mymarks[lane] = 0u;                                    // (the read above ran for all lanes first: LDS keeps a wave's order)
            mv = pair_scan_max(mv);
            const uint32_t gsel = mv >> 16, pbase = mv & 0xFFFFu;
            const uint32_t l = l0 + (uint32_t)lane;
            const bool valid = lane < CH && l < total;
            const uint32_t jl = valid ? gsel * 128u + (uint32_t)list_prev[gsel * 128u + (l - pbase)] : 0u;
            const int n = (int)min((uint32_t)CH, total - l0);
            // lane m: the record of the chunk's row m; lanes past the chunk's end keep row 0's record (a valid LDS block
            // of this wave) with zero bytes: their loads are out of range and return zeros without traffic
            // The packed fields are taken apart HERE, once per chunk on the vector unit: a row then costs six v_readlane and
            // no scalar arithmetic (an extra scalar instruction per row costs the launch five times what a vector one does:
            // profiles/r05_ring_issue_ports.txt).
            const uint4 rc = my_rec[(size_t)jl * WPC];
            if (LSM_PAIR_PRE == 1 && !LSM_PAIR_LEAN && l0 == 0) LSM_PAIR_PRE_UPDATE
            const uint32_t rx = rc.x, rw = rc.w;
            const uint32_t r_so = (uint32_t)(int)(int16_t)(rc.y & 0xFFFFu);        // my block's byte offset in the row (signed)
            const uint32_t r_lo = rc.y >> 16;                                      // LDS offset of its accumulators
            const uint32_t r_nb = valid ? (rc.z & 0xFFFFu) : 0u;                   // bytes of the row that exist
            const uint32_t r_ln = valid ? (rc.z >> 16) : 0u;                       // bytes of my list
#if LSM_PAIR_PHASES
            asm volatile("" : : "v"(rx), "v"(r_so), "v"(r_lo), "v"(r_nb), "v"(r_ln), "v"(rw));    // the records have arrived
#endif
            LSM_PAIR_MARK(1)           // chunk set-up: merged order, list read, row records

            pair_f2 wv[P];                       // my 8 bytes of the window of the rows in flight
            pair_u2 re[P];                       // my list entry of those rows
            uint32_t wa[P];                      // LDS byte address (without the dump bytes) of the pair each row's window piece adds to
            // the accumulators under the row being applied are fetched one row ahead: two sets, by the row's parity
            pair_f2 old[2];                      // under the window piece
            float oldl[2];                       // under my list entry
#if LSM_PAIR_DUMMIES
            uint32_t dummy_s_ = 0u, dummy_v_ = 0u;
            pair_u2 dummy_m_[LSM_PAIR_P][LSM_PAIR_DUMMY_VMEM ? LSM_PAIR_DUMMY_VMEM : 1] = {};
#define LSM_PAIR_DUMMY_WORK                                                                                     \
    {                                                                                                           \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_PAIR_DUMMY_SALU; ++d_)                                      \
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy_s_) : : "scc");                                     \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_PAIR_DUMMY_VALU; ++d_)                                      \
            asm volatile("v_add_u32 %0, 1, %0" : "+v"(dummy_v_));                                               \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_PAIR_DUMMY_LDS; ++d_)                                       \
            asm volatile("ds_write_b32 %0, %1" : : "v"(lane8 >> 1), "v"(dummy_v_) : "memory"); /* dump bytes */ \
    }
            // the zero-byte loads take a buffer of the row pipeline like the real ones (issued with the row, consumed when the
            // row is applied): waiting for one at once would drain the pipeline and measure that instead
#define LSM_PAIR_DUMMY_LOADS(p)                                                                                 \
    {                                                                                                           \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_PAIR_DUMMY_VMEM; ++d_) {                                    \
            const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc(                               \
                reinterpret_cast<void *>(band_hi | (uint64_t)sx), 0, 0, RSRC_FLAGS);                            \
            /* LSM_PAIR_DUMMY_VMEM_LANES: only that many lanes take part (exec mask): does a load's cost follow its lanes? */ \
            if (LSM_PAIR_DUMMY_VMEM_LANES >= 64 || lane < LSM_PAIR_DUMMY_VMEM_LANES)                            \
                dummy_m_[p][d_] = __builtin_amdgcn_raw_buffer_load_b64(rd_, (int)lane8, 0, 0);                  \
        }                                                                                                       \
    }
#define LSM_PAIR_DUMMY_USE(p)                                                                                   \
    {                                                                                                           \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_PAIR_DUMMY_VMEM; ++d_) asm volatile("" : : "v"(dummy_m_[p][d_])); \
    }
#else
#define LSM_PAIR_DUMMY_WORK
#define LSM_PAIR_DUMMY_LOADS(p)
#define LSM_PAIR_DUMMY_USE(p)
#endif
            // SEL(m): the six fields of the chunk's row m into scalar registers.  It runs AHEAD of the row's loads, with the
            // previous row's accumulator update between them: a buffer load that reads a scalar register written by
            // v_readlane needs five wait states, which were three s_nop per row when the loads followed at once.
            uint32_t sx, snb, sw, sln, slo, so;
#define LSM_PAIR_SEL(m)                                                                         \
    {                                                                                           \
        const int mm = (m);                                                                     \
        sx = __builtin_amdgcn_readlane(rx, mm);                                                 \
        snb = __builtin_amdgcn_readlane(r_nb, mm);                                              \
        sw = __builtin_amdgcn_readlane(rw, mm);                                                 \
        sln = __builtin_amdgcn_readlane(r_ln, mm);                                              \
        slo = __builtin_amdgcn_readlane(r_lo, mm);                                              \
        so = __builtin_amdgcn_readlane(r_so, mm);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            // ISSUE(p): request the selected row's window piece and list entry into buffer p
#define LSM_PAIR_ISSUE(p)                                                                       \
    {                                                                                           \
        LSM_PAIR_DUMMY_WORK                                                                     \
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(                    \
            reinterpret_cast<void *>(band_hi | (uint64_t)sx), 0, (int)snb, RSRC_FLAGS);         \
        wa[p] = lane8 + slo;                                                                    \
        /* the whole byte offset goes through the VGPR (opaque scalar: nothing is folded into the  */ \
        /* instruction's immediate), so a lane in front of the row is a huge unsigned offset        */ \
        uint32_t so_ = so;                                                                      \
        asm volatile("" : "+s"(so_));                                                           \
        if (LSM_PAIR_ABLATE & 1) {                                                              \
            wv[p] = (pair_f2){0.0f, 0.0f};                                                      \
        } else {                                                                                \
            const pair_u2 x = __builtin_amdgcn_raw_buffer_load_b64(rb, (int)(lane8 + so_), 0, 0); \
            wv[p] = (pair_f2){__uint_as_float(x.x), __uint_as_float(x.y)};                      \
        }                                                                                       \
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(                    \
            reinterpret_cast<void *>(rem_hi | (uint64_t)sw), 0, (int)sln, RSRC_FLAGS);          \
        if (LSM_PAIR_ABLATE & 8) re[p] = (pair_u2){0u, 0u};                                     \
        else re[p] = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)lane8, 0, 0);                \
        LSM_PAIR_DUMMY_LOADS(p)                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            // READ(p): fetch the accumulators the row in buffer p adds to: the pair under its window piece and the word
            // under my list entry (a lane without one got {0, 0} from the bounds check: LDS word 0, the dump word every
            // such lane rewrites with the same value, old + 0 -- same-address lanes of one LDS instruction do not conflict)
#define LSM_PAIR_READ(p)                                                                        \
    {                                                                                           \
        if (LSM_PAIR_ABLATE & (64 | 128)) old[(p) & 1] = (pair_f2){0.0f, 0.0f};                 \
        else old[(p) & 1] = LSM_PAIR_LDS_F2(wa[p] + PAIR_DUMP_BYTES);                           \
        if (LSM_PAIR_OWN_DUMP) re[p].x = max(re[p].x, lane8 >> 1);                              \
        if (LSM_PAIR_ABLATE & (64 | 256)) oldl[(p) & 1] = 0.0f;                                 \
        else oldl[(p) & 1] = LSM_PAIR_LDS_F1(re[p].x);                                          \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            // APPLY(p): add row p and write back -- the window pair first, the list word after it (a list target may sit in
            // a pair the window piece covers with +0.0: its word then holds old + 0 from the pair write and receives
            // old + weight from the list write, in that order).  Rows are applied in ascending j = the order of these calls.
#define LSM_PAIR_APPLY(p)                                                                       \
    {                                                                                           \
        LSM_PAIR_DUMMY_USE(p)                                                                   \
        if (!(LSM_PAIR_ABLATE & 2)) {                                                           \
            /* the list sum first: its read was issued last, so ONE wait covers both reads */   \
            float newl = oldl[(p) & 1] + __uint_as_float(re[p].y);                              \
            asm volatile("" : "+v"(newl));                                                      \
            if (LSM_PAIR_ABLATE & 128) abl_cur_ = abl_cur_ + wv[p];                             \
            else LSM_PAIR_LDS_F2(wa[p] + PAIR_DUMP_BYTES) = old[(p) & 1] + wv[p];               \
            asm volatile("" ::: "memory");                                                      \
            if (LSM_PAIR_ABLATE & 256) abl_l_ += newl;                                          \
            else LSM_PAIR_LDS_F1(re[p].x) = newl;                                               \
            asm volatile("" ::: "memory");                                                      \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            static_assert(P % 2 == 0, "the look-ahead sets alternate by the row's parity");
            pair_f2 abl_cur_ = {0.0f, 0.0f};       // LSM_PAIR_ABLATE & 128 / 256 only
            float abl_l_ = 0.0f;
            // Exactly the chunk's n rows are requested (a buffer load holds the texture-address path about 11 cycles whether
            // or not its descriptor has bytes: profiles/r05_residency_ablation.txt).  n >= P: P rows requested, whole turns
            // of P rows while the P rows requested in a turn all exist, one last turn that requests the n % P rows left, then
            // those rows on their own.  n < P: the n rows requested together, then applied.
            if (n >= P) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    LSM_PAIR_SEL(p)
                    LSM_PAIR_ISSUE(p)
                }
                LSM_PAIR_MARK(2)       // first P rows requested
                if (LSM_PAIR_PRE == 2 && !LSM_PAIR_LEAN && l0 == 0) LSM_PAIR_PRE_UPDATE
                LSM_PAIR_READ(0)
                int m = 0;
                for (; m + 2 * P <= n; m += P) {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        LSM_PAIR_SEL(m + p + P)
                        LSM_PAIR_APPLY(p)
                        LSM_PAIR_READ((p + 1) % P)
                        LSM_PAIR_ISSUE(p)
                    }
                }
                const int rest = n - m - P;         // 0 .. P-1 rows not yet requested (wave-uniform)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (p < rest) LSM_PAIR_SEL(m + p + P)
                    LSM_PAIR_APPLY(p)
                    if (p + 1 < P) {
                        LSM_PAIR_READ(p + 1)
                    } else if (rest > 0) {
                        LSM_PAIR_READ(0)
                    }
                    if (p < rest) LSM_PAIR_ISSUE(p)
                }
#pragma unroll
                for (int p = 0; p < P - 1; ++p)
                    if (p < rest) {
                        if (p > 0) LSM_PAIR_READ(p)
                        LSM_PAIR_APPLY(p)
                    }
            } else {
#pragma unroll
                for (int p = 0; p < P - 1; ++p)
                    if (p < n) {
                        LSM_PAIR_SEL(p)
                        LSM_PAIR_ISSUE(p)
                    }
                LSM_PAIR_MARK(2)
                if (LSM_PAIR_PRE == 2 && !LSM_PAIR_LEAN && l0 == 0) LSM_PAIR_PRE_UPDATE
#pragma unroll
                for (int p = 0; p < P - 1; ++p)
                    if (p < n) {
                        LSM_PAIR_READ(p)
                        LSM_PAIR_APPLY(p)
                    }
            }
            if (LSM_PAIR_ABLATE & (128 | 256)) LSM_PAIR_LDS_F2(lane8 + PAIR_DUMP_BYTES) = abl_cur_ + (pair_f2){abl_l_, 0.0f};
#undef LSM_PAIR_SEL
#undef LSM_PAIR_ISSUE
#undef LSM_PAIR_DUMMY_WORK
#undef LSM_PAIR_DUMMY_LOADS
#undef LSM_PAIR_DUMMY_USE
#undef LSM_PAIR_READ
#undef LSM_PAIR_APPLY
            LSM_PAIR_MARK(3)           // rows applied (waits for the row loads included)
        }
        wave_lds_fence();
#if LSM_PAIR_PRE && !LSM_PAIR_LEAN
        if (total == 0u) LSM_PAIR_PRE_UPDATE                    // a step without rows
#else
        uint32_t rowbits[4];                                    // this step's input bit row (wave-uniform)
        if (CW == 4) {
            const uint4 q4 = *reinterpret_cast<const uint4 *>(bits + t * 4);
            rowbits[0] = q4.x; rowbits[1] = q4.y; rowbits[2] = q4.z; rowbits[3] = q4.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) rowbits[k] = k < CW ? bits[t * CW + k] : 0u;
        }
#endif

        // ---- neuron update, block by block: leak/integrate/threshold by select, then (only if a neuron of the
        //      block fired) its entries of the block's spike list and the feature accumulators ----
        int nspk = 0;
        uint32_t cntv = 0u;
#pragma unroll
        for (int q = 0; q < BL; ++q) {
            const int gb = LSM_PAIR_GB(q);
            // my two recurrent sums of this block, cleared for the next step
            const pair_f2 cq = *reinterpret_cast<const pair_f2 *>(smem + accl + (uint32_t)gb * 512u);
            *reinterpret_cast<pair_f2 *>(smem + accl + (uint32_t)gb * 512u) = (pair_f2){0.0f, 0.0f};
            float ci[2] = {cq.x, cq.y};
            unsigned long long bq[2];
#if LSM_PAIR_REPLAY
            uint32_t rec2 = 0u;                                     // my two neurons' recorded spikes of this step
            if (gb * 128 + lane * 2 + 1 < N) rec2 = *reinterpret_cast<const uint16_t *>(rec_row + gb * 128 + lane * 2);
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 2 * q + h;
#if LSM_PAIR_PRE && !LSM_PAIR_LEAN
                ci[h] = ci[h] + w_in * (float)((npk[r >> 2] >> (8 * (r & 3))) & 0xFFu);    // SPEC.md §3: input term after the recurrent sum
                const float vn = v[r] + ci[h];             // v[r] holds d = v - lambda * v since LSM_PAIR_PRE_UPDATE
#else
                const uint32_t nn = LSM_PAIR_LEAN ? 0u : pair_input_count<INMASK>(im[r], rowbits);
                ci[h] = ci[h] + w_in * (float)nn;          // SPEC.md §3: input term after the recurrent sum
                const float m = (LEAKV ? lam[r] : lam_u) * v[r];
                const float d = v[r] - m;
                const float vn = d + ci[h];
#endif
#if LSM_PAIR_REPLAY
                const unsigned long long held = LSM_PAIR_LEAN ? 0ull : __builtin_amdgcn_uicmp(oref[r], 0x10000u, 35);
                const unsigned long long ge = __builtin_amdgcn_fcmpf(vn, theta, 3 /* ordered >= */);
                // (the membrane arithmetic stays alive through a mask the compiler cannot see is empty)
                const unsigned long long fire = __builtin_amdgcn_uicmp((rec2 >> (8 * h)) & 0xFFu, 0u, 33 /* != */) |
                                                (ge & ~held & never64);
#else
                const unsigned long long held = __builtin_amdgcn_uicmp(oref[r], 0x10000u, 35 /* unsigned >= */);
                const unsigned long long ge = __builtin_amdgcn_fcmpf(vn, theta, 3 /* ordered >= */);
                const unsigned long long fire = ge & ~held;
#endif
                v[r] = __builtin_amdgcn_inverse_ballot_w64(ge | held) ? 0.0f : vn;
                // countdown: -1 while held, set on fire -- two selects on the lane masks and an add (written out: the
                // compiler turned the nested select into an exec-masked region of six instructions)
                if (!LSM_PAIR_LEAN) {
                    uint32_t dlt;
                    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(dlt) : "v"(ref_set_v), "s"(fire));
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(dlt) : "v"(dlt), "v"(minus_one_v), "s"(held));
                    oref[r] += dlt;
                }
                bq[h] = fire;
            }
            int nq = 0;
            if ((bq[0] | bq[1]) != 0ull) {
                // ascending neuron order inside a block: lane, then h
                int rank = lane_rank(bq[0]) + lane_rank(bq[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = 2 * q + h;
                    if (__builtin_amdgcn_inverse_ballot_w64(bq[h])) {
                        list_cur[gb * 128 + rank] = (uint8_t)(lane * 2 + h);
                        rank += 1;
                        hf |= 1u << r;
                        const int osl = LSM_PAIR_LEAN ? -1 : (int)(oref[r] & 0xFFFFu) - 1;
                        if (osl >= 0 && !(LSM_PAIR_ABLATE & 32)) {
                            uint4 f = feat[osl];
                            uint32_t nf = f.x & 0xFFFFu, bursts = f.x >> 16;
                            uint32_t first = f.y & 0xFFFFu, last = f.y >> 16;
                            const uint32_t isi = (uint32_t)t - last;
                            first = nf == 0 ? (uint32_t)t : first;
                            f.w += nf == 0 ? 0u : isi * isi;
                            bursts += (nf != 0 && (int)isi <= a.burst_isi_max) ? 1u : 0u;
                            last = (uint32_t)t;
                            nf += 1;
                            f.z += (uint32_t)t;
                            f.x = nf | (bursts << 16);
                            f.y = first | (last << 16);
                            feat[osl] = f;
                        }
                    }
                }
                nq = __popcll(bq[0]) + __popcll(bq[1]);
            }
            asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(nq), "n"(q));      // lane q: spikes of my block q
            nspk += nq;
            if (trace) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = gb * 128 + lane * 2 + h;
                    if (i < N) {
                        if (sm_row) sm_row[i] = (uint8_t)((bq[h] >> lane) & 1ull);
                        if (vt_row) vt_row[i] = v[2 * q + h];
                    }
                }
            }
        }
        if (lane < BL) wcnt[cur * PAIR_MAX_BLOCKS + lane * WPC + w] = cntv;      // block lane*WPC + w is my block `lane`
#if LSM_PAIR_REPLAY
        rec_row += N;
#endif
        if (trace) {
            if (sm_row) sm_row += N;
            if (vt_row) vt_row += N;
        }
        tot_spk += (uint32_t)nspk;
        LSM_PAIR_MARK(5)               // neuron update, spike lists, feature accumulators
        __syncthreads();
        LSM_PAIR_MARK(6)               // barrier
    }
#undef LSM_PAIR_GB
    if (LSM_PAIR_PRIO) __builtin_amdgcn_s_setprio(0);

    // ---- epilogue: health statistics, then SPEC.md §4 features from the integer accumulators ----
    if (a.stats) {
        atomicAdd(&wcnt[2 * PAIR_MAX_BLOCKS], (uint32_t)__popc(hf));
        if (lane == 0) atomicAdd(&wcnt[2 * PAIR_MAX_BLOCKS + 1], tot_spk);
        __syncthreads();
        if (tid == 0) {
            a.stats[2 * b] = (int32_t)wcnt[2 * PAIR_MAX_BLOCKS];
            a.stats[2 * b + 1] = (int32_t)wcnt[2 * PAIR_MAX_BLOCKS + 1];
        }
    }
    const int nf = a.n_keys * a.n_out;
    for (int idx = tid; idx < (LSM_PAIR_LEAN ? 0 : nf); idx += NT) {
        const int kq = idx / a.n_out;
        const int o = idx - kq * a.n_out;
        const uint4 f = feat[o];
        const int n = (int)(f.x & 0xFFFFu), bursts = (int)(f.x >> 16);
        const int first = (int)(f.y & 0xFFFFu), last = (int)(f.y >> 16);
        a.features[(size_t)b * nf + idx] = pair_feature_value(a.key_ids[kq], n, bursts, first, last, f.z, f.w, T);
    }
#if LSM_PAIR_PHASES
    __syncthreads();
    float out_ = (float)rows_;
#pragma unroll
    for (int k = 0; k < 7; ++k) out_ = lane == k ? (float)ph_[k] : out_;
    if (lane < 8 && nf >= WPC * 8) a.features[(size_t)b * nf + w * 8 + lane] = out_;
#endif
}

typedef void (*pair_fn_t)(const PairArgs);

template <int BL, int INMASK, bool LEAKV>
pair_fn_t pick_pair_wpc(int wpc)
{
    switch (wpc) {
    case 4: return lif_pair_kernel<BL, 4, INMASK, LEAKV>;
    case 8: return lif_pair_kernel<BL, 8, INMASK, LEAKV>;
    case 16: return lif_pair_kernel<BL, 16, INMASK, LEAKV>;
    default: return nullptr;
    }
}
template <int BL>
pair_fn_t pick_pair(int wpc, int inmask, bool leakv)
{
    if (leakv) return inmask == 2 ? pick_pair_wpc<BL, 2, true>(wpc) : pick_pair_wpc<BL, 1, true>(wpc);
    return inmask == 2 ? pick_pair_wpc<BL, 2, false>(wpc) : pick_pair_wpc<BL, 1, false>(wpc);
}

// one definition per translation unit lif_pair_<bl>.hip
pair_fn_t pick_pair_1(int wpc, int inmask, bool leakv);
pair_fn_t pick_pair_2(int wpc, int inmask, bool leakv);
pair_fn_t pick_pair_3(int wpc, int inmask, bool leakv);
pair_fn_t pick_pair_4(int wpc, int inmask, bool leakv);

}  // namespace lsm_lif
