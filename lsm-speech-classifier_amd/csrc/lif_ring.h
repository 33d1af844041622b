// lif_ring.h -- LIF reservoir time loop on RING rows (small-world reservoirs, N up to 8192), gfx950.
//
// Same contract as lif_kernel.h / lif_dense.h (SPEC.md §3-§4; replaces reset / set_input_spike_times /
// simulate / extract_features_from_spikes of /root/reference/extract_lsm_features.py:79-83), third data
// structure.  The dense-row kernel moves N*4 bytes per spiking neuron and clip, 80 % of them zeros; at
// N >= 4000 that is what bounds it (profiles/r01_big_traffic.json: 8-17x the algorithmic bytes).  A
// Watts-Strogatz reservoir is a ring lattice with ~10 % rewired edges, so row j is stored as
//   * its ring WINDOW: the weights onto the targets j-H .. j+H (circular; from the 128-byte-aligned start), dense, in
//     natural target order -- fetched with bounds-checked 16-byte buffer loads: a lane owns FOUR consecutive
//     neurons of a 256-neuron "quad", and a lane or quad the window does not reach is an out-of-range load
//     that returns zeros without touching memory, so every wave runs the same straight-line code for every
//     row (no per-slot index arithmetic, no branches around loads: the counted waits stay exact) and only
//     the window's own bytes are fetched;
//   * a LIST of the synapses outside the window ("rewired"), per (row, wave), unpadded: {LDS byte offset of
//     the target's accumulator word, weight bits}, one lane per entry.
// The float32 accumulators of a step live in LDS, one word per neuron in neuron order, and belong to the wave
// that owns the neuron.  A row is applied by read-modify-write: the wave reads the 1 KB accumulator quad under
// its window load (16 bytes per lane) and the words its list entries point at, adds, writes the quad back and
// then the list words (LDS runs a wave's instructions in order: the next row's reads see these writes; a lane
// without an entry got {0, 0} from the bounds check and adds 0 to its own word of a 64-word dump region, so
// nothing is masked off).  A row costs a wave ~30 instructions whatever the number of neurons it owns; the
// first version of this kernel kept the sums in registers and handed the list weights over through an LDS
// scratch array that every lane read back in full for every row (~60 instructions per row and wave at 16
// neurons per lane, 32 KB of LDS reads per row at N = 8000) -- see DESIGN.md 4b.
// A target receives row j's weight from the window or from the list, never both, the other term is +0.0
// (x + 0 = x in float32 for every x the sum can take), and rows are applied in ascending j: every target's
// float32 sum keeps the oracle's order and is bit-identical.
//
// Quad ownership.  CONTIGUOUS: wave w owns quads w*QL .. w*QL+QL-1; a window (<= 8 quads) then lies in two or
// three waves, which issue QL window loads per row while the others issue out-of-range ones.  STRIDED: wave w
// owns quads w, w+WPC, w+2*WPC, ...; a window of <= WPC quads puts exactly ONE quad into every wave, so every
// wave issues one useful 1 KB load per row (a quarter of the load instructions, and all the waves keep row
// gathers in flight instead of two or three of them).  STRIDED needs NQ % WPC == 0 (the residues must survive
// the ring's wrap) and window quads <= WPC; the host offers it when that holds and prefers it (N = 8000, 512
// clips: 29.0 ms against 42.0 ms contiguous).
//
// Spike exchange: per QUAD lists (ascending inside a quad: lane, then the lane's four neurons) + one count
// per quad; quad order = ascending neuron order under both ownerships, so lane l of a consumer finds the
// l-th spiking neuron of the clip by a prefix over the quad counts.
//
// Rows are pipelined: P rows' loads are in flight while the oldest is applied, and the accumulator reads of row
// m+1 are issued right after the writes of row m, ahead of the scalar work that launches row m+P.
#pragma once
#include "lif_kernel.h"

namespace lsm_lif {

#define LSM_RING_ADD4(a, b) a = a + b;      // four floats: two v_pk_add_f32 (four v_add_f32 measured equal)

#ifndef LSM_RING_PHASES
#define LSM_RING_PHASES 0   // diagnostic builds only: 1 = every wave sums the core-clock cycles of its step phases and writes them
#endif                      // OVER the feature rows (exp/r03_ring_phases.py reads them back); results are not features.
#if LSM_RING_PHASES         // A mark is not free: s_memtime returns through lgkmcnt (it waits for the wave's outstanding LDS
                            // operations) and the compiler moves no memory operation across it -- phases that the product build
                            // overlaps (e.g. the input-map fetch with the last rows) show up serialised here.
#define LSM_RING_MARK(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); ph_[k] += (uint32_t)(now_ - last_); last_ = now_; }
#else
#define LSM_RING_MARK(k)
#endif

#ifndef LSM_RING_PRIO
#define LSM_RING_PRIO 1         // wave priority of the step loop (s_setprio 0..3; profiles/r04_ring_priority.txt)
#endif
#ifndef LSM_RING_DRIVE_AT
#define LSM_RING_DRIVE_AT 0     // 0 = at the top of the step (product: 6.09 ms at cfg4), 1 = behind the first row loads of the step
#endif                          // (6.13 ms and seven spilled registers: profiles/r04_ring_input_drive.txt)

#ifndef LSM_RING_DRIVE_ALL_LANES
#define LSM_RING_DRIVE_ALL_LANES 0
#endif
// Diagnostic builds only (results stay right): extra scalar / vector / LDS instructions per row and wave, to see which
// issue port the row loop is bound by (profiles/r05_ring_issue_ports.txt).
#ifndef LSM_RING_DUMMY_SALU
#define LSM_RING_DUMMY_SALU 0
#endif
#ifndef LSM_RING_DUMMY_VALU
#define LSM_RING_DUMMY_VALU 0
#endif
#ifndef LSM_RING_DUMMY_LDS
#define LSM_RING_DUMMY_LDS 0
#endif
#if LSM_RING_DUMMY_SALU || LSM_RING_DUMMY_VALU || LSM_RING_DUMMY_LDS
#define LSM_RING_DUMMY_WORK                                                                                     \
    {                                                                                                           \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_RING_DUMMY_SALU; ++d_)                                      \
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy_s_) : : "scc");                                     \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_RING_DUMMY_VALU; ++d_)                                      \
            asm volatile("v_add_u32 %0, 1, %0" : "+v"(dummy_v_));                                               \
        _Pragma("unroll") for (int d_ = 0; d_ < LSM_RING_DUMMY_LDS; ++d_)                                       \
            asm volatile("ds_write_b32 %0, %1" : : "v"(lane4), "v"(dummy_v_) : "memory");  /* my dump word */   \
    }
#else
#define LSM_RING_DUMMY_WORK
#endif
#ifndef LSM_RING_ABLATE
#define LSM_RING_ABLATE 0   // diagnostic builds only (1, 2, 8, 16, 32 give WRONG results): 1 = no window loads, 2 = no
#endif                      // accumulator read-modify-write, 8 = no list loads, 16 = no input drive, 32 = no feature updates;
                            // 64 = the input drive issued TWICE, the second pass adding zeros (results stay right: its time is the drive's cost)

struct RingArgs {
    int N, C, T, B;
    int n_out, CW, EinW, refractory, burst_isi_max;
    int H;                     // ring half-width (window = targets j-H .. j+H, circular)
    int NQ;                    // quads holding real neurons: ceil(N / 256)
    uint32_t pitch;            // bytes between consecutive band rows
    float theta, w_in;
    const uint8_t *raster;     // (B, C, T) uint8
    const float *band;         // (N, pitch/4): window of row j from its 32-aligned start, natural target order
    const uint32_t *rem_ptr;   // (N*WPC + 1) first list entry of (row j, wave w)
    const uint2 *rem;          // list entries {LDS byte offset of the target's accumulator, weight bits}
    const float *leak;         // (NPAD), neuron order
    const int *oslot;          // (NPAD) output slot or -1, neuron order
    const uint32_t *in_ent;    // (WPC, EinW) packed entries (ring_pack_entry), EinW a multiple of RING_ENT_BLOCK, padded with dump-word entries
    const uint32_t *inmask;    // INMASK: (NPAD, 4) input-channel bit mask per neuron, neuron order (C <= 128), else null
    const uint8_t *inperm;     // INMASK with coloured positions: (C) bit position of channel c in the input bit row, else null
    float leak_u;              // INMASK: the one leak coefficient of every neuron (the kernel form exists for uniform leaks)
    int n_keys;
    int key_ids[8];
    float *features;           // (B, n_keys * n_out)
    uint8_t *spike_matrix;     // (B, T, N) or null
    float *v_trace;            // (B, T, N) or null
    int32_t *stats;            // (B, 2) {neurons that fired at least once, spikes of the whole reservoir} or null
    const int32_t *order;      // (B) clip of workgroup g, or null (g): lsm_reservoir_run_ordered starts long clips first
};

typedef float ring_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t ring_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t ring_u2 __attribute__((ext_vector_type(2)));

constexpr int RING_DUMP_WORDS = 64;                 // LDS words 0..63 of an array: where idle lanes add their zeros
constexpr int RING_MAX_QUADS = 32;                  // 8192 neurons

// Input-map entries of the ring kernel, one packed word per (channel c -> target neuron i) pair, built by the host:
//   bits  0..14  byte offset of i's 16-bit count from the start of the count array (its dump words included)
//   bit   15     i & 1: the count sits in the upper half of its word
//   bits 16..20  c & 31, bits 21..31  c >> 5: where the channel's bit sits in the step's input bit row
// A wave's entries are padded to whole blocks of RING_ENT_BLOCK: the padding entry at index e adds channel 0's bit to dump
// word e % 64 (= its lane's own dump word, which nobody reads) -- so the drive runs without guards, exec masks or
// branches, and without two padding lanes ever adding to the same word.
constexpr int RING_ENT_BLOCK = 512;                 // entries per block: 8 per lane
constexpr int RING_ENT_REG_BLOCKS = 1;              // blocks a wave keeps resident in registers (INREG): 8 registers
__host__ __device__ inline uint32_t ring_pack_entry(int c, int i)
{
    return (uint32_t)((RING_DUMP_WORDS + (i >> 1)) * 4) | ((uint32_t)(i & 1) << 15) | ((uint32_t)(c & 31) << 16) |
           ((uint32_t)(c >> 5) << 21);
}

// owner of quad g in a layout: (wave, register quad)
__host__ __device__ inline int ring_wave_of_quad(int g, int ql, int wpc, bool strided) { return strided ? g % wpc : g / ql; }
__host__ __device__ inline int ring_slot_of_quad(int g, int ql, int wpc, bool strided) { return strided ? g / wpc : g % ql; }

// accumulator WORD index (from the start of LDS) of neuron i: neuron order behind the dump words, so a lane's four
// neurons are one 16-byte word group and a quad is 1 KB (conflict-free 16-byte accesses of 64 lanes)
__host__ __device__ inline int ring_acc_word(int i) { return RING_DUMP_WORDS + i; }
// input counts: 16 bits per neuron, two neurons per word, behind their own dump words
__host__ __device__ inline int ring_cnt_word(int i) { return RING_DUMP_WORDS + (i >> 1); }

// QL: quads (256 neurons, 4 per lane) per wave; WPC: waves per clip; INREG: the wave's input-map entries (at most
// RING_ENT_REG_BLOCKS blocks of 512, one packed word each) stay in registers, else they stream from global memory
// every step; STRIDED: quad ownership (see above).
// (Up to two quads per wave the strided kernel must keep four waves per SIMD -- 128 registers -- whatever it holds in
// registers: two clips per CU is what its LDS image allows, and that is 16 waves.)
// INMASK (round 4; 0 = off, 1 = natural bit positions, 2 = coloured positions): the input drive as in the dense kernel's INMODE
// 2 / 3 -- every neuron holds the bit mask of its input channels in four registers and counts popcount(mask & row) in the update;
// no input-map entries, no LDS atomics, no count array on the step's critical path (the entry drive cost ~1.0 of cfg4's 6.0 ms,
// profiles/r04_ring_input_drive.txt).  The 4 x SL registers exist only when the leak coefficients do not need SL of their own:
// offered for UNIFORM leaks (the reference's default, leak_variance_divisor = None), C <= 128 and at most two quads per wave.
template <int QL, int WPC, bool INREG, bool STRIDED, int INMASK = 0>
__global__ __launch_bounds__(WPC * 64) __attribute__((amdgpu_waves_per_eu((QL <= 2 && STRIDED) ? 4 : 1)))
void lif_ring_kernel(const RingArgs a)
{
    constexpr int SL = 4 * QL;
    constexpr int NQP = QL * WPC;                   // quads of the padded layout (<= 32)
    constexpr int NPAD = NQP * 256;
    constexpr int NT = WPC * 64;
    constexpr int WL = STRIDED ? 1 : QL;            // window loads per row and wave
#ifndef LSM_RING_P                                  // experiment switches: rows in flight (strided / contiguous)
#define LSM_RING_P 4
#endif
#ifndef LSM_RING_PC
#define LSM_RING_PC (QL >= 4 ? 4 : 8)
#endif
    // rows in flight (WL*4 + 2 registers each), even.  Round 4, after the update's lane masks freed registers: four quads per
    // wave (N = 8000) take 6 -- 142 registers, still three waves per SIMD: cfg5 163.2 -> 160.4 ms --, two quads keep 4
    // (N = 4000 with 6: 6.02 -> 6.23 ms; profiles/r04_ring_masks_and_rows_in_flight.txt)
    constexpr int P = STRIDED ? (QL == 4 && LSM_RING_P == 4 ? 6 : LSM_RING_P) : LSM_RING_PC;
    constexpr uint32_t RSRC_FLAGS = 0x00020000u;    // raw dword buffer, gfx9 family
    static_assert(NQP <= RING_MAX_QUADS, "at most 8192 neurons");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                                     // dump + NPAD accumulators
    uint32_t *cnt = reinterpret_cast<uint32_t *>(acc + RING_DUMP_WORDS + NPAD);       // dump + NPAD/2 input counts (none with INMASK)
    constexpr int CNT_WORDS = INMASK ? 0 : RING_DUMP_WORDS + NPAD / 2;
    uint16_t *wlist = reinterpret_cast<uint16_t *>(cnt + CNT_WORDS);                  // 2*NPAD: 256 per quad
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(wlist + 2 * NPAD);                  // 2*32 quad counts + 2 stats
    uint4 *feat = reinterpret_cast<uint4 *>(wcnt + 128);                              // n_out
    uint32_t *bits = reinterpret_cast<uint32_t *>(feat + a.n_out);                    // T*CW

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;      // wave-uniform
    const int N = a.N, T = a.T, CW = a.CW;
    // global quad of my register quad q
#define LSM_RING_GQ(q) (STRIDED ? (q) * WPC + w : w * QL + (q))

    // ---- prologue: zero LDS state, bit-pack the clip's raster time-major ----
    for (int i = tid; i < RING_DUMP_WORDS + NPAD + CNT_WORDS; i += NT) reinterpret_cast<uint32_t *>(smem)[i] = 0u;
    for (int i = tid; i < 128; i += NT) wcnt[i] = 0u;
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

    // my neurons: register r = 4*q + h  <->  neuron GQ(q)*256 + lane*4 + h
    // oref[r] = (output slot + 1) | (refractory countdown << 16): one register for both, "held" is one
    // unsigned compare, the slot is unpacked only when the neuron fires.
    float v[SL], lam[INMASK ? 1 : SL];
    uint32_t oref[SL];
    uint32_t im[INMASK ? SL : 1][4];              // INMASK: channels 0..127 feeding my neuron r
    const float lam_u = a.leak_u;
#pragma unroll
    for (int q = 0; q < QL; ++q) {
        const int i0 = LSM_RING_GQ(q) * 256 + lane * 4;
        const float4 l4 = INMASK ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(a.leak + i0);
        const int4 o4 = *reinterpret_cast<const int4 *>(a.oslot + i0);
        const float l[4] = {l4.x, l4.y, l4.z, l4.w};
        const int o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            // padding neurons (i >= N) start with a NaN potential: it stays NaN through every update, the
            // threshold test is false for ever -- a window that wraps past the ring's end may deliver weights
            // of real quads to them, and they must never fire
            v[4 * q + h] = (i0 + h) < N ? 0.0f : __builtin_nanf("");
            if (!INMASK) lam[4 * q + h] = l[h];
            oref[4 * q + h] = (uint32_t)(o[h] + 1);
            if (INMASK) {
                const uint4 m = reinterpret_cast<const uint4 *>(a.inmask)[i0 + h];
                im[4 * q + h][0] = m.x; im[4 * q + h][1] = m.y; im[4 * q + h][2] = m.z; im[4 * q + h][3] = m.w;
            }
        }
    }
    const uint32_t ref_set = (uint32_t)a.refractory << 16;
    const float theta = a.theta, w_in = a.w_in;
    const uint32_t *my_ent = a.in_ent + (size_t)w * a.EinW;            // EinW: a multiple of RING_ENT_BLOCK
    constexpr int EPL = RING_ENT_BLOCK / 64;                            // entries per lane and block
    uint32_t ent_reg[(INREG && !INMASK) ? EPL * RING_ENT_REG_BLOCKS : 1];   // INREG: the wave's whole map, resident
    if (INREG && !INMASK) {
#pragma unroll
        for (int u = 0; u < EPL * RING_ENT_REG_BLOCKS; ++u) ent_reg[u] = my_ent[u * 64 + lane];
    }
    const bool trace = a.spike_matrix != nullptr || a.v_trace != nullptr;
    // byte address (from the start of LDS) of my four accumulators of global quad g: acc_b + g*1024
    const uint32_t acc_b = (uint32_t)RING_DUMP_WORDS * 4u + (uint32_t)lane * 16u;
    const uint32_t cnt_b = (uint32_t)(2 * RING_DUMP_WORDS + NPAD) * 4u + (uint32_t)lane * 8u;   // my four counts: + g*512
    const uint32_t lane4 = (uint32_t)lane * 4u, lane8 = (uint32_t)lane * 8u, lane16 = (uint32_t)lane * 16u;
    const uint64_t band_base = reinterpret_cast<uint64_t>(a.band);
    const uint64_t rem_base = reinterpret_cast<uint64_t>(a.rem);
    const int H = a.H, NQ = a.NQ;
#if LSM_RING_PHASES
    uint32_t ph_[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t rows_ = 0u;
    uint64_t last_ = 0;
#endif
#if LSM_RING_DUMMY_SALU || LSM_RING_DUMMY_VALU || LSM_RING_DUMMY_LDS
    uint32_t dummy_s_ = 0u, dummy_v_ = 0u;
#endif
    uint32_t hf = 0u;                  // bit r: my neuron r fired at least once (stats)
    uint32_t tot_spk = 0u;             // spikes of my wave (stats)
    __syncthreads();

    // Input drive of step `ts`: every entry looks its channel up in the step's bit row and adds the bit to its target's
    // 16-bit count (integer LDS atomics: two entries of one instruction may share a target).  One block = 8 entries
    // per lane, straight-line: the 8 row-word reads are in flight together, then the 8 atomics -- the first version
    // read, waited and added entry by entry behind per-entry guards, and a same-box build that issued the drive twice
    // put it at 1.49 of cfg4's 6.44 ms (profiles/r04_ring_input_drive.txt).  A wave's entries only ever touch the counts
    // of its own neurons, which it cleared itself in the previous update: the drive needs no barrier and can run anywhere
    // between two updates of its wave.
    auto drive_block = [&](const uint32_t *row, const uint32_t (&x)[EPL], uint32_t keep) __attribute__((always_inline)) {
        uint32_t rw[EPL];
#pragma unroll
        for (int u = 0; u < EPL; ++u) rw[u] = row[x[u] >> 21];
        // Only the lanes whose channel is active add (about a quarter of them on speech-like rasters): what the drive
        // costs is the LDS pipe's time for the atomics -- 8 per lane and step with all 64 lanes adding took 1.34 of
        // cfg4's 6.17 ms (same-box build that issues the drive twice) --, and that time goes with the lanes that take
        // part and the bank conflicts among them.
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const uint32_t bit = (rw[u] >> ((x[u] >> 16) & 31u)) & 1u;
            const uint32_t inc = bit << ((x[u] >> 11) & 16u);
#if LSM_RING_DRIVE_ALL_LANES                       // diagnostic builds: every lane adds (zeros included)
            atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(cnt) + (x[u] & 0x7FFFu)), inc & keep);
#else
            if (bit & keep)
                atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(cnt) + (x[u] & 0x7FFFu)), inc);
#endif
        }
    };
    auto input_drive = [&](int ts, uint32_t keep = 0xFFFFFFFFu) {
        if (INMASK) return;                       // counted in the update, from the masks
        const uint32_t *row = bits + ts * CW;
        if (INREG) {
#pragma unroll
            for (int q = 0; q < RING_ENT_REG_BLOCKS; ++q) {
                uint32_t x[EPL];
#pragma unroll
                for (int u = 0; u < EPL; ++u) x[u] = ent_reg[q * EPL + u];
                drive_block(row, x, keep);
            }
        } else {
            for (int e0 = 0; e0 < a.EinW; e0 += RING_ENT_BLOCK) {
                uint32_t x[EPL];
#pragma unroll
                for (int u = 0; u < EPL; ++u) x[u] = my_ent[e0 + u * 64 + lane];
                drive_block(row, x, keep);
            }
        }
    };

#if LSM_RING_PHASES
    last_ = __builtin_amdgcn_s_memtime();
#endif
    // Inside the pipeline the ring kernel is the stage that bounds the step (cfg4: 5.1 ms of reservoir against 2.5 ms of front
    // end per 1024 clips), and a float64 filterbank wave shares every SIMD with its waves for 40 % of the time: the step loop runs
    // at raised wave priority so that its latency chain is not stretched by the front end's issue slots.
    if (LSM_RING_PRIO) __builtin_amdgcn_s_setprio(LSM_RING_PRIO);
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        const uint16_t *list_prev = wlist + prv * NPAD;
        uint16_t *list_cur = wlist + cur * NPAD;

        // The input drive is independent of the rows (see input_drive): it is issued once per step, at the top (behind
        // the first row loads of the step -- LSM_RING_DRIVE_AT=1, diagnostic builds -- it measured 0.7 % slower).
        bool drove = false;
        auto drive_once = [&]() __attribute__((always_inline)) {
            if (drove) return;
            drove = true;
            if (!(LSM_RING_ABLATE & 16)) input_drive(t);
            if (LSM_RING_ABLATE & 64) {
                uint32_t zero = 0u;
                asm volatile("" : "+s"(zero));      // opaque: the second pass is not folded away
                input_drive(t, zero);
            }
        };
        if (LSM_RING_DRIVE_AT == 0) drive_once();
        LSM_RING_MARK(4)               // input counts issued (the fetch of streamed input-map entries included)

        // ---- spiking neurons of step t-1: prefix of the per-quad counts, lane l <- l-th neuron ----
        const uint32_t cv = wcnt[prv * 32 + (lane & 31)];
        uint32_t total = 0u;
#pragma unroll
        for (int g = 0; g < NQP; ++g) total += __builtin_amdgcn_readlane(cv, g);

        LSM_RING_MARK(0)               // quad counts read, total known
#if LSM_RING_PHASES
        rows_ += total;
#endif
        for (uint32_t l0 = 0; l0 < total; l0 += 64) {
            const uint32_t l = l0 + lane;
            uint32_t gsel = 0u, pbase = 0u, run = 0u;      // (the prefix is formed again per chunk: it would
#pragma unroll                                             //  otherwise hold NQP scalar registers across the rows)
            for (int g = 1; g < NQP; ++g) {
                run += __builtin_amdgcn_readlane(cv, g - 1);
                const bool ge = l >= run;
                gsel += ge ? 1u : 0u;
                pbase = ge ? run : pbase;
            }
            const bool valid = l < total;
            const int jl = valid ? (int)list_prev[gsel * 256 + (l - pbase)] : 0;
            // ---- lane l: everything a wave needs to fetch row jl, computed for 64 rows at once ----
            // The stored row starts at a4 = the 32-aligned (128-byte) start of jl-H and runs to jl+H along the PADDED ring
            // (NQ*256 positions: neurons N..NQ*256-1 do not exist and hold zeros), so target i sits at byte
            // ((i - a4) mod NQ*256)*4; only those bytes exist (num_records).  The quad at ring position p (global
            // quad (q0 + p) mod NQ, q0 = a4 >> 8) therefore starts at byte (p*256 - (a4 & 255))*4: negative for
            // the lanes of the first quad that lie in front of the window -- a huge unsigned offset, out of range.
            int a0 = jl - H; a0 += a0 < 0 ? N : 0;
            int b0 = jl + H; b0 -= b0 >= N ? N : 0;
            const int a4 = a0 & ~31;        // rows start on a 128-byte line: a wave's 1 KB load touches 8 lines, not 9
            const int q0 = a4 >> 8, lead = a4 & 255;
            int span = b0 - a4; span += span < 0 ? NQ * 256 : 0;
            const uint32_t p_nrec = valid ? (uint32_t)(((span >> 2) + 1) * 16) : 0u;
            uint32_t p_soff;
            if (STRIDED) {
                // the one window quad with residue w (NQ % WPC == 0): position (w - q0) mod WPC, global quad
                // (q0 + position) mod NQ, my register quad = that / WPC -- carried in the offset's low 4 bits
                int ph = (w - q0) % WPC; ph += ph < 0 ? WPC : 0;
                int gh = q0 + ph; gh -= gh >= NQ ? NQ : 0;
                p_soff = (uint32_t)((ph * 256 - lead) * 4) | (uint32_t)(gh / WPC);
            } else {
                // position of my wave's first quad in that row: quads below q0 sit NQ further (wrapped window); a
                // wave never holds both ends of a window (host: window quads + QL <= NQ)
                const int g0 = w * QL;
                const int base = (g0 + QL - 1 < q0) ? g0 - q0 + NQ : g0 - q0;
                p_soff = (uint32_t)((base * 256 - lead) * 4);           // negative: wraps to > any num_records
            }
            // Three packed words per row instead of seven values: the 64-bit addresses are formed per row by the
            // SCALAR unit (the kernel is bound by vector-instruction issue at N = 4000; v_readlane is a vector
            // instruction, s_mul/s_add are not): A = row | bytes/16 << 16, B = offset word, C = list start | count << 24.
            // (lanes past the chunk's end look at row 0: a valid address, and `live` voids their rows below --
            //  no branch around these loads)
            const uint32_t r0 = a.rem_ptr[jl * WPC + w];
            const uint32_t r1 = a.rem_ptr[jl * WPC + w + 1];
            const uint32_t p_a = (uint32_t)jl | ((p_nrec >> 4) << 16);
            const uint32_t p_c = r0 | ((r1 - r0) << 24);                // host: fewer than 2^24 list entries per layout
            const int n = (int)min(64u, total - l0);
#if LSM_RING_PHASES
            asm volatile("" : : "v"(p_a), "v"(p_c), "v"(p_soff));      // the row words (and the pointer loads) have arrived
#endif
            LSM_RING_MARK(1)           // chunk set-up: prefix, list read, geometry, list pointers
            ring_f4 wv[P][WL];
            ring_u2 re[P];
            ring_f4 old[WL];                     // accumulators under the window load of the row being applied
            float oldl;                          // accumulator under my list entry of that row
            uint32_t wa[WL];                     // their LDS byte addresses
            uint32_t pa;
            uint32_t qh[P];                      // STRIDED: register quad the window load of buffer p belongs to
            // LOADW / LOADL(p, m): issue the window / list load of the chunk's row m (a scalar) into buffer p.  Rows >= n do not
            // exist: their num_records is 0, every load is out of range and returns zeros without traffic.
#define LSM_RING_LOADW(p, m)                                                                    \
    {                                                                                           \
        LSM_RING_DUMMY_WORK                                                                     \
        const int mm = (m) & 63;                                                                \
        const uint32_t live = (uint32_t) - (int)((m) < n);                                      \
        const uint32_t ra = __builtin_amdgcn_readlane(p_a, mm);                                 \
        const uint32_t soff = __builtin_amdgcn_readlane(p_soff, mm);                            \
        const uint32_t nrec = ((ra >> 16) << 4) & live;                                         \
        const uint64_t baddr = band_base + (uint64_t)((ra & 0xFFFFu) * a.pitch);                \
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(                    \
            reinterpret_cast<void *>(baddr), 0, (int)nrec, RSRC_FLAGS);                         \
        qh[p] = soff & 15u;                                                                     \
        _Pragma("unroll") for (int q = 0; q < WL; ++q) {                                        \
            /* the whole byte offset goes through the VGPR (opaque scalar: nothing is folded into the    */ \
            /* instruction's immediate), so a quad in front of the window is a huge unsigned offset and  */ \
            /* out of range whatever the address adder does with a carry                                  */ \
            uint32_t so = (STRIDED ? (soff & ~15u) : soff) + (uint32_t)q * 1024u;               \
            asm volatile("" : "+s"(so));                                                        \
            if (LSM_RING_ABLATE & 1) {                                                          \
                wv[p][q] = (ring_f4){0.0f, 0.0f, 0.0f, 0.0f};                                   \
            } else {                                                                            \
                const ring_u4 x = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(lane16 + so), 0, 0); \
                wv[p][q] = (ring_f4){__uint_as_float(x.x), __uint_as_float(x.y),                \
                                     __uint_as_float(x.z), __uint_as_float(x.w)};               \
            }                                                                                   \
        }                                                                                       \
    }
#define LSM_RING_LOADL(p, m)                                                                    \
    {                                                                                           \
        const int mm = (m) & 63;                                                                \
        const uint32_t live = (uint32_t) - (int)((m) < n);                                      \
        const uint32_t rc = __builtin_amdgcn_readlane(p_c, mm);                                 \
        const uint32_t rnrec = ((rc >> 24) << 3) & live;                                        \
        const uint64_t raddr = rem_base + (uint64_t)((rc & 0xFFFFFFu) << 3);                    \
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(                    \
            reinterpret_cast<void *>(raddr), 0, (int)rnrec, RSRC_FLAGS);                        \
        if (LSM_RING_ABLATE & 8) re[p] = (ring_u2){0u, 0u};                                     \
        else re[p] = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)lane8, 0, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            // READ(p): fetch the accumulators the row in buffer p adds to: the quad(s) under its window load and
            // the word under my list entry (a lane without one got {0, 0} from the bounds check: its own dump word).
#define LSM_RING_READ(p)                                                                        \
    {                                                                                           \
        _Pragma("unroll") for (int q = 0; q < WL; ++q) {                                        \
            wa[q] = acc_b + (uint32_t)(STRIDED ? (qh[p] * WPC + w) : (w * QL + q)) * 1024u;     \
            old[q] = *reinterpret_cast<const ring_f4 *>(smem + wa[q]);                          \
        }                                                                                       \
        pa = max(re[p].x, lane4);                                                               \
        oldl = *reinterpret_cast<const float *>(smem + pa);                                     \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            // APPLY(p): add row p and write back -- the window quad first, the list word after it (a list target
            // may sit in a part of that quad the window does not cover: its word then holds old + 0 from the
            // quad write and receives old + weight from the list write, in that order).  Rows are applied in
            // ascending j = the order of these calls; LDS keeps a wave's accesses in program order.
#define LSM_RING_APPLY(p)                                                                       \
    {                                                                                           \
        if (!(LSM_RING_ABLATE & 2)) {                                                           \
            _Pragma("unroll") for (int q = 0; q < WL; ++q) {                                    \
                LSM_RING_ADD4(old[q], wv[p][q])                                                 \
                *reinterpret_cast<ring_f4 *>(smem + wa[q]) = old[q];                            \
            }                                                                                   \
            asm volatile("" ::: "memory");                                                      \
            *reinterpret_cast<float *>(smem + pa) = oldl + __uint_as_float(re[p].y);            \
            asm volatile("" ::: "memory");                                                      \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            // software pipeline: P rows of loads in flight; the accumulator reads of row m+1 are issued right
            // after the writes of row m and their round trip hides behind the scalar work that launches row m+P
#pragma unroll
            for (int p = 0; p < P; ++p) {
                LSM_RING_LOADW(p, p)
                LSM_RING_LOADL(p, p)
            }
            LSM_RING_MARK(2)           // first P rows requested
            if (LSM_RING_DRIVE_AT == 1) drive_once();
            LSM_RING_READ(0)
            for (int m = 0; m < n; m += P) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    LSM_RING_APPLY(p)
                    LSM_RING_READ((p + 1) % P)
                    LSM_RING_LOADW(p, m + p + P)
                    LSM_RING_LOADL(p, m + p + P)
                }
            }
            LSM_RING_MARK(3)           // rows applied (waits for the row loads included)
#undef LSM_RING_LOADW
#undef LSM_RING_LOADL
#undef LSM_RING_READ
#undef LSM_RING_APPLY
        }
        drive_once();                  // a step without reservoir spikes: no rows to hide behind
        wave_lds_fence();
        uint32_t rowbits[4] = {0u, 0u, 0u, 0u};              // INMASK: this step's input bit row (wave-uniform)
        if (INMASK) {
            if (CW == 4) {
                const uint4 q4 = *reinterpret_cast<const uint4 *>(bits + t * 4);
                rowbits[0] = q4.x; rowbits[1] = q4.y; rowbits[2] = q4.z; rowbits[3] = q4.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) rowbits[k] = k < CW ? bits[t * CW + k] : 0u;
            }
        }

        // ---- neuron update, quad by quad: leak/integrate/threshold by select, then (only if a neuron of the
        //      quad fired) its entries of the quad's spike list and the feature accumulators ----
        int nspk = 0;
#pragma unroll
        for (int q = 0; q < QL; ++q) {
            const int gq = LSM_RING_GQ(q);
            // my four recurrent sums and input counts of this quad; both arrays are cleared for the next step
            const ring_f4 cq = *reinterpret_cast<const ring_f4 *>(smem + acc_b + (uint32_t)gq * 1024u);
            *reinterpret_cast<ring_f4 *>(smem + acc_b + (uint32_t)gq * 1024u) = (ring_f4){0.0f, 0.0f, 0.0f, 0.0f};
            uint32_t nn[4];
            if (INMASK) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int r = 4 * q + h;
                    if (INMASK == 2) {
                        // disjoint by construction of the bit positions: one popcount of the union (lif_dense.h, INMODE 3)
                        uint32_t u = im[r][0] & rowbits[0];
#pragma unroll
                        for (int k = 1; k < 4; ++k)
                            asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(u) : "v"(im[r][k]), "v"(rowbits[k]));
                        nn[h] = __popc(u);
                    } else {
                        nn[h] = __popc(im[r][0] & rowbits[0]) + __popc(im[r][1] & rowbits[1]) +
                                __popc(im[r][2] & rowbits[2]) + __popc(im[r][3] & rowbits[3]);
                    }
                }
            } else {
                const ring_u2 nin = *reinterpret_cast<const ring_u2 *>(smem + cnt_b + (uint32_t)gq * 512u);
                *reinterpret_cast<ring_u2 *>(smem + cnt_b + (uint32_t)gq * 512u) = (ring_u2){0u, 0u};
                nn[0] = nin.x & 0xFFFFu; nn[1] = nin.x >> 16; nn[2] = nin.y & 0xFFFFu; nn[3] = nin.y >> 16;
            }
            float ci[4] = {cq.x, cq.y, cq.z, cq.w};
            unsigned long long bq[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int r = 4 * q + h;
                ci[h] = ci[h] + w_in * (float)nn[h];       // SPEC.md §3: input term after the recurrent sum
                const float m = (INMASK ? lam_u : lam[r]) * v[r];
                const float d = v[r] - m;
                const float vn = d + ci[h];
                // lane masks straight from the compares (round 4, as in lif_dense.h): `held`, `fire` and the reset's
                // select are scalar mask arithmetic + one v_cndmask; a ballot of a materialised boolean costs two more
                const unsigned long long held = __builtin_amdgcn_uicmp(oref[r], 0x10000u, 35 /* unsigned >= */);
                const unsigned long long ge = __builtin_amdgcn_fcmpf(vn, theta, 3 /* ordered >= */);
                const unsigned long long fire = ge & ~held;
                v[r] = __builtin_amdgcn_inverse_ballot_w64(ge | held) ? 0.0f : vn;
                oref[r] += __builtin_amdgcn_inverse_ballot_w64(held) ? 0xFFFF0000u
                           : (__builtin_amdgcn_inverse_ballot_w64(fire) ? ref_set : 0u);
                bq[h] = fire;
            }
            int nq = 0;
            if ((bq[0] | bq[1] | bq[2] | bq[3]) != 0ull) {
                // ascending neuron order inside a quad: lane, then h
                int rank = lane_rank(bq[0]) + lane_rank(bq[1]) + lane_rank(bq[2]) + lane_rank(bq[3]);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int r = 4 * q + h;
                    if (__builtin_amdgcn_inverse_ballot_w64(bq[h])) {
                        list_cur[gq * 256 + rank] = (uint16_t)(gq * 256 + lane * 4 + h);
                        rank += 1;
                        hf |= 1u << r;
                        const int osl = (int)(oref[r] & 0xFFFFu) - 1;
                        if (osl >= 0 && !(LSM_RING_ABLATE & 32)) {
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
                nq = __popcll(bq[0]) + __popcll(bq[1]) + __popcll(bq[2]) + __popcll(bq[3]);
            }
            if (lane == 0) wcnt[cur * 32 + gq] = (uint32_t)nq;
            nspk += nq;
            if (trace) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int i = gq * 256 + lane * 4 + h;
                    if (i < N) {
                        if (a.spike_matrix)
                            a.spike_matrix[((size_t)b * T + t) * N + i] = (uint8_t)((bq[h] >> lane) & 1ull);
                        if (a.v_trace) a.v_trace[((size_t)b * T + t) * N + i] = v[4 * q + h];
                    }
                }
            }
        }
        tot_spk += (uint32_t)nspk;
        LSM_RING_MARK(5)               // neuron update, spike lists, feature accumulators
        __syncthreads();
        LSM_RING_MARK(6)               // barrier
    }
#undef LSM_RING_GQ
    if (LSM_RING_PRIO) __builtin_amdgcn_s_setprio(0);

    // ---- epilogue: health statistics, then SPEC.md §4 features from the integer accumulators ----
    if (a.stats) {
        atomicAdd(&wcnt[64], (uint32_t)__popc(hf));
        if (lane == 0) atomicAdd(&wcnt[65], tot_spk);
        __syncthreads();
        if (tid == 0) {
            a.stats[2 * b] = (int32_t)wcnt[64];
            a.stats[2 * b + 1] = (int32_t)wcnt[65];
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
#if LSM_RING_PHASES
    __syncthreads();
    float out_ = (float)rows_;
#pragma unroll
    for (int k = 0; k < 7; ++k) out_ = lane == k ? (float)ph_[k] : out_;
    if (lane < 8 && nf >= WPC * 8) a.features[(size_t)b * nf + w * 8 + lane] = out_;
#endif
}

typedef void (*ring_fn_t)(const RingArgs);

template <int QL, bool INREG, bool STRIDED>
ring_fn_t pick_ring_wpc(int wpc)
{
    switch (wpc) {
    case 2: return lif_ring_kernel<QL, 2, INREG, STRIDED>;
    case 4: return lif_ring_kernel<QL, 4, INREG, STRIDED>;
    case 8: return lif_ring_kernel<QL, 8, INREG, STRIDED>;
    case 16:
        if constexpr (QL * 16 <= RING_MAX_QUADS) return lif_ring_kernel<QL, 16, INREG, STRIDED>;   // N <= 8192 = 16 waves x 2 quads
        else return nullptr;
    default: return nullptr;
    }
}

template <int QL>
ring_fn_t pick_ring(int wpc, bool inreg, bool strided)
{
    if (strided) return inreg ? pick_ring_wpc<QL, true, true>(wpc) : pick_ring_wpc<QL, false, true>(wpc);
    return inreg ? pick_ring_wpc<QL, true, false>(wpc) : pick_ring_wpc<QL, false, false>(wpc);
}

// INMASK forms: strided ownership, at most two quads per wave (lif_ring_1.hip, lif_ring_2.hip); inmask = 1 natural / 2 coloured
template <int QL, int INMASK>
ring_fn_t pick_ring_mask_wpc(int wpc)
{
    switch (wpc) {
    case 2: return lif_ring_kernel<QL, 2, false, true, INMASK>;
    case 4: return lif_ring_kernel<QL, 4, false, true, INMASK>;
    case 8: return lif_ring_kernel<QL, 8, false, true, INMASK>;
    case 16:
        if constexpr (QL * 16 <= RING_MAX_QUADS) return lif_ring_kernel<QL, 16, false, true, INMASK>;
        else return nullptr;
    default: return nullptr;
    }
}
template <int QL>
ring_fn_t pick_ring_mask(int wpc, int inmask)
{
    return inmask == 2 ? pick_ring_mask_wpc<QL, 2>(wpc) : pick_ring_mask_wpc<QL, 1>(wpc);
}

// one definition per translation unit lif_ring_<ql>.hip
ring_fn_t pick_ring_1(int wpc, bool inreg, bool strided);
ring_fn_t pick_ring_2(int wpc, bool inreg, bool strided);
ring_fn_t pick_ring_3(int wpc, bool inreg, bool strided);
ring_fn_t pick_ring_4(int wpc, bool inreg, bool strided);
ring_fn_t pick_ring_mask_1(int wpc, int inmask);
ring_fn_t pick_ring_mask_2(int wpc, int inmask);

}  // namespace lsm_lif
