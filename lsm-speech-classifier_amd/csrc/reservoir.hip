// Host side of the reservoir C ABI (include/lsm_hip.h): handle creation (CSC upload, per-layout
// segment tables), layout choice and kernel launch.  The kernel itself lives in lif_kernel.h and
// is instantiated by the four lif_variant_*.hip translation units.
#include "lif_dense.h"
#include "lif_ring.h"
#include "lif_pair.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>
#include <vector>

using lsm_lif::LifArgs;
using lsm_lif::lif_fn_t;
using lsm_lif::IN_REG_SLOTS;

namespace {

lif_fn_t pick_kernel(int sl, int wpc, bool inreg, bool seglds)
{
    if (inreg) return seglds ? lsm_lif::pick_lif_11(sl, wpc) : lsm_lif::pick_lif_10(sl, wpc);
    return seglds ? lsm_lif::pick_lif_01(sl, wpc) : lsm_lif::pick_lif_00(sl, wpc);
}

#if LSM_STAMP
}  // namespace
namespace lsm_lif {
int read_lif_stamps(int unit, unsigned long long *o, int r)
{
    switch (unit) {
    case 0: return read_lif_stamps_00(o, r);
    case 1: return read_lif_stamps_01(o, r);
    case 2: return read_lif_stamps_10(o, r);
    default: return read_lif_stamps_11(o, r);
    }
}
}  // namespace lsm_lif
namespace {
#endif

struct Variant {            // per waves-per-clip layout
    int wpc = 0, sl = 0, einw = 0;
    uint32_t *seg = nullptr;
    uint16_t *segoff = nullptr;
    float *leak = nullptr;
    int *oslot = nullptr;
    uint32_t *in_ent = nullptr;
    uint32_t *inmask = nullptr;      // (npad, 4) input-channel masks, only when C <= 128 and sl <= 4
    bool incol = false;              // the masks use the coloured bit positions of lsm_reservoir::inperm (INMODE 3)
};

struct RingVariant {        // per waves-per-clip layout of the ring-row kernel (lif_ring.h)
    int wpc = 0, ql = 0, einw = 0;
    bool strided = false;            // quad ownership: wave w owns quads w, w+wpc, ... (else w*ql .. w*ql+ql-1)
    size_t n_rem = 0;                // list entries (synapses outside the ring window)
    uint32_t *rem_ptr = nullptr;     // (N*wpc + 1) first list entry of (row, wave)
    uint2 *rem = nullptr;            // synapses outside the ring window: {LDS byte offset of the accumulator, weight bits}
    float *leak = nullptr;
    int *oslot = nullptr;
    uint32_t *in_ent = nullptr;      // (wpc, einw) packed input-map entries (lsm_lif::ring_pack_entry)
    uint32_t *inmask = nullptr;      // (npad, 4) input-channel masks in neuron order: the INMASK kernel form (C <= 128, uniform
                                     // leak, strided ownership, <= 2 quads per wave), else null
    bool incol = false;              // the masks use the coloured bit positions of lsm_reservoir::inperm
};

struct PairVariant {        // per waves-per-clip layout of the pair-block ring kernel (lif_pair.h)
    int wpc = 0, bl = 0;             // wave w owns the 128-neuron blocks w, w+wpc, ...: bl of them
    size_t n_rem = 0;                // list entries (synapses outside the ring window)
    uint4 *rec = nullptr;            // (N*wpc) row records (lsm_lif::pair_record)
    uint2 *rem = nullptr;            // {LDS byte offset of the accumulator, weight bits}, (row, wave) major
    int *oslot = nullptr;
    float *leak = nullptr;           // (npad) leak coefficients when they differ between neurons, else null
    uint32_t *inmask = nullptr;      // (npad, 4) input-channel masks in neuron order
    bool incol = false;              // the masks use the coloured bit positions of lsm_reservoir::inperm
};

}  // namespace

// Dense presynaptic rows from the device copy of the CSC arrays: block j scatters column j into row j.
__global__ __launch_bounds__(256) void build_dense_rows_kernel(const uint32_t *__restrict__ rowptr,
                                                               const uint2 *__restrict__ syn, int ld,
                                                               float *__restrict__ wt)
{
    const uint32_t j = blockIdx.x;
    for (uint32_t e = rowptr[j] + threadIdx.x; e < rowptr[j + 1]; e += blockDim.x) {
        const uint2 s = syn[e];
        wt[(size_t)j * ld + s.x] = __uint_as_float(s.y);
    }
}

struct lsm_reservoir {
    int N = 0, C = 0, n_out = 0, refractory = 0, burst_isi_max = 0;
    float theta = 0, w_in = 0;
    int device = 0;
    int cus = 0;            // compute units of that device
    size_t nnz = 0;
    uint2 *syn = nullptr;
    uint32_t *rowptr = nullptr;
    float *wt = nullptr;    // dense rows by presynaptic neuron (N, ld), small reservoirs only
    int ld = 0;
    // ring rows (lif_ring.h): per presynaptic neuron the dense ring window from the 256-aligned start of
    // j-H up to j+H, plus per-layout lists of the synapses outside it
    float *band = nullptr;
    uint32_t band_pitch = 0;
    double band_bytes_sum = 0;       // sum over rows of the window bytes that exist (what a spike's window loads can touch)
    int band_h = 0, band_nq = 0, band_wsq = 0;
    // INMODE 3 of the dense kernel: bit position of every input channel in the step's input bit row such that the
    // channels feeding one neuron sit at different positions mod 32 (colour_input_channels); null when no such
    // assignment was found (then the masks keep the natural positions, INMODE 2)
    uint8_t *inperm = nullptr;
    bool leak_uniform = false;       // every neuron has the same leak coefficient (leak_u): the reference's default
    float leak_u = 0.0f;
    int mode = 0;           // 0 auto, 1 sparse (CSC scatter through LDS), 2 dense rows, 3 ring rows,
                            // 4 ring rows with contiguous quad ownership only (tests), 5 ring rows in pair blocks only
                            // (lif_pair.h), 6 ring rows in quads only (lif_ring.h, either ownership)
    Variant var[5];         // wpc = 1, 2, 4, 8, 16 (wpc == 0: not available)
    RingVariant rvar[8];    // wpc = 2, 4, 8, 16, contiguous [0..3] and strided [4..7] quad ownership
    PairVariant pvar[3];    // wpc = 4, 8, 16: pair blocks (lif_pair.h), uniform leak and C <= 128 only
};

static int free_reservoir(lsm_reservoir *h)
{
    if (!h) return LSM_OK;
    if (h->syn) (void)hipFree(h->syn);
    if (h->rowptr) (void)hipFree(h->rowptr);
    if (h->wt) (void)hipFree(h->wt);
    if (h->band) (void)hipFree(h->band);
    if (h->inperm) (void)hipFree(h->inperm);
    for (auto &v : h->var) {
        if (v.seg) (void)hipFree(v.seg);
        if (v.segoff) (void)hipFree(v.segoff);
        if (v.leak) (void)hipFree(v.leak);
        if (v.oslot) (void)hipFree(v.oslot);
        if (v.in_ent) (void)hipFree(v.in_ent);
        if (v.inmask) (void)hipFree(v.inmask);
    }
    for (auto &v : h->rvar) {
        if (v.rem_ptr) (void)hipFree(v.rem_ptr);
        if (v.rem) (void)hipFree(v.rem);
        if (v.inmask) (void)hipFree(v.inmask);
        if (v.leak) (void)hipFree(v.leak);
        if (v.oslot) (void)hipFree(v.oslot);
        if (v.in_ent) (void)hipFree(v.in_ent);
    }
    for (auto &v : h->pvar) {
        if (v.rec) (void)hipFree(v.rec);
        if (v.rem) (void)hipFree(v.rem);
        if (v.oslot) (void)hipFree(v.oslot);
        if (v.leak) (void)hipFree(v.leak);
        if (v.inmask) (void)hipFree(v.inmask);
    }
    delete h;
    return LSM_OK;
}

static size_t ring_lds_bytes(const lsm_reservoir *h, const RingVariant &v, int T);

static bool has_ring(const lsm_reservoir *h)
{
    for (const auto &v : h->rvar)
        if (v.wpc) return true;
    return false;
}
static bool has_pairs(const lsm_reservoir *h)
{
    for (const auto &v : h->pvar)
        if (v.wpc) return true;
    return false;
}

// ring rows: on request, or by default when the dense table no longer fits the XCDs' L2 caches together
// (32 MB) -- from there on the dense rows are bound by the bytes of the row gathers, of which the ring
// format moves a quarter (N = 4000: window 3.6 KB + list 0.6 KB against 16 KB per row)
constexpr size_t RING_AUTO_MIN_DENSE_BYTES = (size_t)32 << 20;

// Allocate and fill the dense row table of a handle that does not have it yet (synchronous: handle set-up, not a
// launch function).
static int ensure_dense_rows(lsm_reservoir *h)
{
    if (h->wt || h->ld <= 0) return LSM_OK;
    int dev_now = -1;
    LSM_CHECK_HIP(hipGetDevice(&dev_now));
    LSM_REQUIRE(dev_now == h->device, "reservoir handle lives on device %d but the current device is %d",
                h->device, dev_now);
    const size_t bytes = (size_t)h->N * (size_t)h->ld * sizeof(float);
    float *wt = nullptr;
    LSM_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&wt), bytes));
    hipError_t e = hipMemset(wt, 0, bytes);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(build_dense_rows_kernel, dim3(h->N), dim3(256), 0, nullptr, h->rowptr, h->syn, h->ld, wt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(wt);
        lsm_set_error("building the dense row table failed: %s", hipGetErrorString(e));
        return LSM_ERR_HIP;
    }
    h->wt = wt;
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

// Bit positions for the input channels of the dense kernel's INMODE 3 (C <= 128): position p = word * 32 + bit with
// word < ceil(C / 32), such that the channels feeding ONE neuron all have different `bit`s.  Then the words
// (mask_w & row_w) of a neuron have no set bit in common and popcount(m0&r0) + ... + popcount(m3&r3) =
// popcount((m0&r0) | (m1&r1) | (m2&r2) | (m3&r3)): one v_and, three v_and_or and one v_bcnt per neuron and step
// instead of four v_and, four v_bcnt and an add.  This is an equitable colouring of the channels' conflict graph
// (two channels conflict when they share a target) with 32 colours of ceil(C/32) places each; greedy by conflict
// degree with a one-move repair finds one for every input map tried (N = 256..4000, C = 32..128).  Returns false
// when it does not: the caller keeps the natural positions (INMODE 2).
static bool colour_input_channels(int N, int C, const int32_t *in_tgt, int in_fanout, std::vector<uint8_t> *perm)
{
    if (C > 128) return false;
    const int cap = (C + 31) / 32;
    std::vector<std::vector<int>> chans_of(N);
    for (int c = 0; c < C; ++c)
        for (int d = 0; d < in_fanout; ++d) chans_of[in_tgt[(size_t)c * in_fanout + d]].push_back(c);
    std::vector<std::vector<char>> adj(C, std::vector<char>(C, 0));
    std::vector<int> deg(C, 0);
    for (const auto &l : chans_of)
        for (int x : l)
            for (int y : l)
                if (x != y && !adj[x][y]) { adj[x][y] = 1; ++deg[x]; }
    std::vector<int> order(C);
    for (int c = 0; c < C; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return deg[x] > deg[y]; });
    std::vector<int> col(C, -1), cnt(32, 0);
    auto used_by_neighbours = [&](int c, int skip) {
        uint32_t used = 0u;
        for (int y = 0; y < C; ++y)
            if (adj[c][y] && col[y] >= 0 && y != skip) used |= 1u << col[y];
        return used;
    };
    for (int c : order) {
        const uint32_t used = used_by_neighbours(c, -1);
        int best = -1;
        for (int k = 0; k < 32; ++k)
            if (!((used >> k) & 1u) && cnt[k] < cap && (best < 0 || cnt[k] < cnt[best])) best = k;
        if (best < 0) {
            // every colour free of c's neighbours is full: move one member of such a colour elsewhere
            for (int k = 0; k < 32 && best < 0; ++k) {
                if ((used >> k) & 1u) continue;
                for (int v = 0; v < C && best < 0; ++v) {
                    if (col[v] != k) continue;
                    const uint32_t uv = used_by_neighbours(v, -1);
                    for (int q = 0; q < 32; ++q)
                        if (q != k && !((uv >> q) & 1u) && cnt[q] < cap && !adj[v][c]) {
                            col[v] = q; ++cnt[q]; --cnt[k];
                            best = k;
                            break;
                        }
                }
            }
            if (best < 0) return false;
        }
        col[c] = best;
        ++cnt[best];
    }
    // place the members of a colour in different words; verify before trusting the result
    perm->assign(C, 0);
    std::vector<int> next(32, 0);
    for (int c = 0; c < C; ++c) (*perm)[c] = (uint8_t)(next[col[c]]++ * 32 + col[c]);
    for (const auto &l : chans_of) {
        uint32_t seen = 0u;
        for (int x : l) {
            const uint32_t bit = 1u << ((*perm)[x] & 31);
            if (seen & bit) return false;
            seen |= bit;
        }
    }
    std::vector<char> taken(cap * 32, 0);
    for (int c = 0; c < C; ++c) {
        if ((*perm)[c] >= cap * 32 || taken[(*perm)[c]]) return false;
        taken[(*perm)[c]] = 1;
    }
    return true;
}

// ---- ring-row geometry and the pair-block tables: host arithmetic only (no HIP call), shared by lsm_reservoir_create and
// lsm_debug_pair_layout, which lets a CPU test apply every row from the tables alone (tests/test_pair_layout.py) ----
struct RingGeometry {
    int H = 0, NQ = 0, NP = 0, wsq = 0;
    uint32_t pitch = 0;
    std::vector<int> a4v, nbytes;       // per row: 32-aligned first target of the window; bytes of the row that exist
    // byte offset of synapse (j -> i) in row j, or -1 when i lies outside the stored window
    long win_off(int j, int i) const
    {
        int p = i - a4v[j]; p += p < 0 ? NP : 0;
        const long off = (long)p * 4;
        return off < nbytes[j] ? off : -1;
    }
};

// Window half-width H = half the mean out-degree (k/2 of a small-world graph).  Row j covers the targets from the
// 32-aligned start of (j-H) mod N up to (j+H) mod N along the ring padded to NQ quads of 256.  False when the reservoir is
// not ring-like enough for the format (the plain window must hold most of the synapses and be a real saving).
static bool ring_geometry(int N, const int32_t *csc_ptr, const int32_t *csc_post, size_t nnz, RingGeometry *g)
{
    const int H = (int)((nnz / (size_t)N + 1) / 2);
    const int wd = 2 * H + 1;
    const int NQ = (N + 255) / 256;
    size_t inside = 0;
    for (int j = 0; j < N; ++j)
        for (int e = csc_ptr[j]; e < csc_ptr[j + 1]; ++e) {
            int q = csc_post[e] - (j - H);
            q = q < 0 ? q + N : (q >= N ? q - N : q);
            inside += q < wd;
        }
    if (!(H >= 1 && 2 * wd <= N && NQ >= 2 && nnz > 0 && inside * 10 >= nnz * 6)) return false;
    // geometry per row: a4 (32-aligned first target of the window: 128 bytes), bytes that exist (up to the window's
    // end along the padded ring of NQ*256 positions, 16-byte granules), quads the window touches
    g->H = H; g->NQ = NQ; g->NP = NQ * 256;
    g->a4v.assign(N, 0); g->nbytes.assign(N, 0);
    int wsq = 0, maxb = 0;
    for (int j = 0; j < N; ++j) {
        int a0 = j - H; a0 += a0 < 0 ? N : 0;
        int b0 = j + H; b0 -= b0 >= N ? N : 0;
        const int a4 = a0 & ~31;
        int span = b0 - a4; span += span < 0 ? g->NP : 0;
        g->a4v[j] = a4;
        g->nbytes[j] = ((span >> 2) + 1) * 16;
        maxb = std::max(maxb, g->nbytes[j]);
        wsq = std::max(wsq, (((a4 & 255) + span) >> 8) + 1);
    }
    g->wsq = wsq;
    g->pitch = ((uint32_t)maxb + 127u) & ~127u;         // every row starts on a 128-byte line
    return wsq < NQ;
}

static void ring_windows(int N, const int32_t *csc_ptr, const int32_t *csc_post, const float *csc_w, const RingGeometry &g,
                         std::vector<float> *band)
{
    band->assign((size_t)N * (g.pitch / 4), 0.0f);
    for (int j = 0; j < N; ++j)
        for (int e = csc_ptr[j]; e < csc_ptr[j + 1]; ++e) {
            const long off = g.win_off(j, csc_post[e]);
            if (off >= 0) (*band)[(size_t)j * (g.pitch / 4) + (size_t)(off / 4)] = csc_w[e];
        }
}

// Pair blocks (lif_pair.h): the windows shared out in 128-neuron blocks, wave w owning the blocks w, w+wpc, ...: the
// residues must survive the ring's wrap (2*NQ % wpc == 0), a window must not touch more than wpc blocks, a wave at most
// four blocks (8 neurons per lane), a (row, wave) list at most 64 entries.  Returns the blocks per wave, 0 = no layout.
static int pair_lists(int N, int wpc, const int32_t *csc_ptr, const int32_t *csc_post, const float *csc_w,
                      const RingGeometry &g, std::vector<uint32_t> *rptr, std::vector<uint2> *rem)
{
    const int NB = 2 * g.NQ;
    int wsb = 0;
    for (int j = 0; j < N; ++j) wsb = std::max(wsb, (((g.a4v[j] & 127) + (g.nbytes[j] / 4 - 1)) >> 7) + 1);
    if (NB % wpc != 0 || wsb > wpc) return 0;
    const int bl = NB / wpc;
    if (bl < 1 || bl > 4) return 0;
    rptr->assign((size_t)N * wpc + 1, 0u);
    int emax = 0;
    for (int j = 0; j < N; ++j)
        for (int e = csc_ptr[j]; e < csc_ptr[j + 1]; ++e)
            if (g.win_off(j, csc_post[e]) < 0)
                emax = std::max(emax, (int)++(*rptr)[(size_t)j * wpc + lsm_lif::pair_wave_of_block(csc_post[e] >> 7, wpc) + 1]);
    if (emax > 64) return 0;                                 // one lane per list entry
    for (size_t q = 1; q < rptr->size(); ++q) (*rptr)[q] += (*rptr)[q - 1];
    if ((uint64_t)rptr->back() * 8u >= (1ull << 32)) return 0;
    rem->assign(std::max<size_t>(1, rptr->back()), make_uint2(0u, 0u));
    std::vector<uint32_t> fill(rptr->begin(), rptr->end() - 1);
    for (int j = 0; j < N; ++j)
        for (int e = csc_ptr[j]; e < csc_ptr[j + 1]; ++e) {
            const int i = csc_post[e];
            if (g.win_off(j, i) >= 0) continue;
            uint32_t bits;
            std::memcpy(&bits, &csc_w[e], 4);
            (*rem)[fill[(size_t)j * wpc + lsm_lif::pair_wave_of_block(i >> 7, wpc)]++] = make_uint2(lsm_lif::pair_acc_byte(i), bits);
        }
    return bl;
}

// The records carry addresses (low halves) of the window table at band_a and of the list table at rem_a.
static void pair_records(int N, int wpc, const RingGeometry &g, const std::vector<uint32_t> &rptr, uint64_t band_a, uint64_t rem_a,
                         std::vector<uint4> *rec)
{
    const int NB = 2 * g.NQ;
    rec->resize((size_t)N * wpc);
    for (int j = 0; j < N; ++j) {
        const int q0 = g.a4v[j] >> 7, lead = g.a4v[j] & 127;
        for (int w = 0; w < wpc; ++w) {
            int ph = (w - q0) % wpc; ph += ph < 0 ? wpc : 0;
            int gb = q0 + ph; gb -= gb >= NB ? NB : 0;
            const size_t q = (size_t)j * wpc + w;
            (*rec)[q] = lsm_lif::pair_record((uint32_t)(band_a + (uint64_t)j * g.pitch), (ph * 128 - lead) * 4, gb,
                                            (uint32_t)g.nbytes[j], (uint32_t)(rem_a + (uint64_t)rptr[q] * 8u), rptr[q + 1] - rptr[q]);
        }
    }
}

// Host-only (no GPU, no HIP call): the ring-window table, the pair-block lists and the row records lsm_reservoir_create would
// build for these CSC arrays and `wpc` waves per clip, with the tables placed at the given (fictitious) addresses.  Sizes first
// (null outputs), then the arrays.  Returns the blocks per wave, 0 when the reservoir has no pair layout with that many waves,
// < 0 on a bad argument.  tests/test_pair_layout.py applies every row from these tables alone and compares with the column.
extern "C" __attribute__((visibility("default")))
int lsm_debug_pair_layout(int num_neurons, const int32_t *csc_ptr, const int32_t *csc_post, const float *csc_w, int wpc,
                          unsigned long long band_addr, unsigned long long rem_addr, long *band_floats, long *n_list_entries,
                          int *pitch_bytes, float *band_out, uint32_t *rem_out, uint32_t *rec_out)
{
    LSM_REQUIRE(num_neurons >= 1 && num_neurons <= 8192 && csc_ptr && csc_post && csc_w, "lsm_debug_pair_layout: bad argument");
    LSM_REQUIRE(wpc == 4 || wpc == 8 || wpc == 16, "lsm_debug_pair_layout: wpc must be 4, 8 or 16");
    const int N = num_neurons;
    RingGeometry g;
    if (!ring_geometry(N, csc_ptr, csc_post, (size_t)csc_ptr[N], &g)) return 0;
    std::vector<uint32_t> rptr;
    std::vector<uint2> rem;
    const int bl = pair_lists(N, wpc, csc_ptr, csc_post, csc_w, g, &rptr, &rem);
    if (!bl) return 0;
    if (band_floats) *band_floats = (long)N * (g.pitch / 4);
    if (n_list_entries) *n_list_entries = (long)rptr.back();
    if (pitch_bytes) *pitch_bytes = (int)g.pitch;
    if (band_out) {
        std::vector<float> band;
        ring_windows(N, csc_ptr, csc_post, csc_w, g, &band);
        std::memcpy(band_out, band.data(), band.size() * sizeof(float));
    }
    if (rem_out) std::memcpy(rem_out, rem.data(), (size_t)rptr.back() * sizeof(uint2));
    if (rec_out) {
        std::vector<uint4> rec;
        pair_records(N, wpc, g, rptr, band_addr, rem_addr, &rec);
        std::memcpy(rec_out, rec.data(), rec.size() * sizeof(uint4));
    }
    return bl;
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
    // 8192: beyond that the per-clip LDS image (counts, spike lists, features, input bits) of neither kernel
    // fits the 160 KB of a CU
    LSM_REQUIRE(N >= 1 && N <= 8192, "num_neurons=%d outside [1, 8192]", N);
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
    if (hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) {
        (void)hipGetLastError();
        h->cus = 0;
    }

    std::vector<uint2> syn(nnz);
    for (size_t e = 0; e < nnz; ++e) {
        uint32_t bits;
        std::memcpy(&bits, &csc_w[e], 4);
        syn[e] = make_uint2((uint32_t)csc_post[e], bits);
    }
    int rc = upload(&h->syn, syn);
    if (rc) { free_reservoir(h); return rc; }
    std::vector<uint32_t> rowptr(csc_ptr, csc_ptr + N + 1);
    if ((rc = upload(&h->rowptr, rowptr))) { free_reservoir(h); return rc; }

    // coloured bit positions for the input masks of the dense kernel (one assignment per reservoir)
    std::vector<uint8_t> inperm;
#if LSM_EXPERIMENT_HOOKS
    static const bool no_incol = [] { const char *e = getenv("LSM_DENSE_NO_INCOL"); return e && atoi(e) != 0; }();
#else
    constexpr bool no_incol = false;
#endif
    const bool coloured = !no_incol && colour_input_channels(N, C, in_tgt, in_fanout, &inperm);
    if (coloured && (rc = upload(&h->inperm, inperm))) { free_reservoir(h); return rc; }
    h->leak_uniform = true;
    h->leak_u = leak[0];
    for (int i = 1; i < N; ++i)
        if (std::memcmp(&leak[i], &leak[0], sizeof(float)) != 0) { h->leak_uniform = false; break; }

    const int wpcs[5] = {1, 2, 4, 8, 16};
    for (int vi = 0; vi < 5; ++vi) {
        const int wpc = wpcs[vi];
        const int need = ((N + wpc - 1) / wpc + 63) / 64;      // 64-neuron slots per wave
        int sl = 1;
        while (sl < need) sl <<= 1;
        if (sl > 16) continue;
        const int npw = sl * 64, npad = npw * wpc;
        Variant &v = h->var[vi];
        // segment table: entry (j, w) = first synapse of column j whose target belongs to wave w;
        // the segment ends where entry (j, w+1) -- or column j+1 -- begins.
        std::vector<uint32_t> seg((size_t)N * wpc + 1);
        for (int j = 0; j < N; ++j) {
            int e = csc_ptr[j];
            for (int w = 0; w < wpc; ++w) {
                seg[(size_t)j * wpc + w] = (uint32_t)e;
                const int hi = (w + 1) * npw;
                while (e < csc_ptr[j + 1] && csc_post[e] < hi) ++e;
            }
        }
        seg[(size_t)N * wpc] = (uint32_t)nnz;
        // packed form for LDS: u16 offsets relative to the row start, WPC + 1 per neuron
        std::vector<uint16_t> segoff((size_t)N * (wpc + 1) + 2, 0);
        for (int j = 0; j < N; ++j) {
            for (int w = 0; w < wpc; ++w)
                segoff[(size_t)j * (wpc + 1) + w] = (uint16_t)(seg[(size_t)j * wpc + w] - (uint32_t)csc_ptr[j]);
            segoff[(size_t)j * (wpc + 1) + wpc] = (uint16_t)(csc_ptr[j + 1] - csc_ptr[j]);
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
        if ((rc = upload(&v.seg, seg)) || (rc = upload(&v.segoff, segoff)) || (rc = upload(&v.leak, lk)) || (rc = upload(&v.oslot, os)) ||
            (rc = upload(&v.in_ent, ent))) {
            free_reservoir(h);
            return rc;
        }
        if (C <= 128 && sl <= 4) {
            // INMODE 2 of the dense kernel: bit c of neuron i's mask = channel c feeds neuron i (each
            // (channel, neuron) pair occurs at most once: SPEC.md 2.4 draws targets without replacement)
            std::vector<uint32_t> im((size_t)npad * 4, 0u);
            bool distinct = true;
            for (int c = 0; c < C; ++c)
                for (int d = 0; d < in_fanout; ++d) {
                    const int tgt = in_tgt[(size_t)c * in_fanout + d];
                    const int pc = coloured ? inperm[c] : c;          // bit position of channel c in the input bit row
                    uint32_t &word = im[(size_t)tgt * 4 + (pc >> 5)];
                    distinct = distinct && !(word & (1u << (pc & 31)));
                    word |= 1u << (pc & 31);
                }
            if (distinct && (rc = upload(&v.inmask, im))) {
                free_reservoir(h);
                return rc;
            }
            v.incol = distinct && coloured;
        }
        v.wpc = wpc; v.sl = sl; v.einw = einw;
    }
    // dense presynaptic rows for the register-accumulating kernel (lif_dense.h): N x ld floats (4 MB at N = 1000:
    // L2; 64 MB at N = 4000 and 262 MB at N = 8000).  Built further down, and only when auto mode can run them
    // (ensure_dense_rows); a reservoir that auto mode serves with ring rows gets them on lsm_reservoir_set_kernel(2).
    for (const auto &v : h->var)
        if (v.wpc) h->ld = std::max(h->ld, v.sl * 64 * v.wpc);
    // Ring rows for ring-like graphs (lif_ring.h, lif_pair.h): ring_geometry / ring_windows above.
    {
        // (the ring kernel counts a neuron's active input entries of a step in 16 bits)
        std::vector<uint32_t> fanin(N, 0u);
        uint32_t max_fanin = 0;
        for (size_t e = 0; e < (size_t)C * in_fanout; ++e) max_fanin = std::max(max_fanin, ++fanin[in_tgt[e]]);
        RingGeometry geo;
        if (max_fanin <= 65535u && ring_geometry(N, csc_ptr, csc_post, nnz, &geo)) {
            {
                const int H = geo.H, NQ = geo.NQ, wsq = geo.wsq;
                const uint32_t pitch = geo.pitch;
                const std::vector<int> &nbytes = geo.nbytes;
                auto win_off = [&](int j, int i) -> long { return geo.win_off(j, i); };
                std::vector<float> band;
                ring_windows(N, csc_ptr, csc_post, csc_w, geo, &band);
                if ((rc = upload(&h->band, band))) { free_reservoir(h); return rc; }
                h->band_pitch = pitch; h->band_h = H; h->band_nq = NQ; h->band_wsq = wsq;
                for (int j = 0; j < N; ++j) h->band_bytes_sum += nbytes[j];
                const int rwpcs[4] = {2, 4, 8, 16};
                for (int vi = 0; vi < 8; ++vi) {
                    const int wpc = rwpcs[vi & 3];
                    const bool strided = vi >= 4;
                    int ql = 0;
                    if (strided) {
                        // every wave gets exactly one window quad per row: the residues mod wpc must survive the
                        // ring's wrap (NQ % wpc == 0) and a window must not be wider than wpc quads
                        if (NQ % wpc != 0 || wsq > wpc) continue;
                        ql = NQ / wpc;
                        if (ql < 1 || ql > 4) continue;
                    } else {
                        for (int cand : {1, 2, 3, 4})
                            if (!ql && wpc * cand * 256 >= N) ql = cand;
                        // a wave must never hold both ends of a (wrapped) window: window quads + QL <= NQ
                        if (!ql || wsq + ql > NQ) continue;
                    }
                    const int npad = wpc * ql * 256;
                    auto wave_of = [&](int i) { return lsm_lif::ring_wave_of_quad(i >> 8, ql, wpc, strided); };
                    std::vector<uint32_t> rptr((size_t)N * wpc + 1, 0u);
                    int emax = 0;
                    for (int j = 0; j < N; ++j)
                        for (int e = csc_ptr[j]; e < csc_ptr[j + 1]; ++e)
                            if (win_off(j, csc_post[e]) < 0)
                                emax = std::max(emax, (int)++rptr[(size_t)j * wpc + wave_of(csc_post[e]) + 1]);
                    if (emax > 64) continue;                             // one lane per list entry
                    for (size_t q = 1; q < rptr.size(); ++q) rptr[q] += rptr[q - 1];
                    if (rptr.back() >= (1u << 24)) continue;             // the kernel packs a list's start into 24 bits
                    std::vector<uint2> rem(std::max<size_t>(1, rptr.back()));
                    std::vector<uint32_t> fill(rptr.begin(), rptr.end() - 1);
                    for (int j = 0; j < N; ++j)
                        for (int e = csc_ptr[j]; e < csc_ptr[j + 1]; ++e) {
                            const int i = csc_post[e];
                            if (win_off(j, i) >= 0) continue;
                            uint32_t bits;
                            std::memcpy(&bits, &csc_w[e], 4);
                            rem[fill[(size_t)j * wpc + wave_of(i)]++] =
                                make_uint2((uint32_t)lsm_lif::ring_acc_word(i) * 4u, bits);
                        }
                    std::vector<float> lk(npad, 0.0f);
                    std::vector<int> os(npad, -1);
                    for (int i = 0; i < N; ++i) lk[i] = leak[i];
                    for (int o = 0; o < n_out; ++o) os[out_idx[o]] = o;
                    // input map per owner wave, packed (lsm_lif::ring_pack_entry), padded with zero entries to whole
                    // blocks of RING_ENT_BLOCK so that the kernel's drive needs no guards
                    std::vector<std::vector<uint32_t>> per(wpc);
                    for (int c = 0; c < C; ++c)
                        for (int d = 0; d < in_fanout; ++d) {
                            const int tgt = in_tgt[(size_t)c * in_fanout + d];
                            per[wave_of(tgt)].push_back(lsm_lif::ring_pack_entry(c, tgt));
                        }
                    size_t mx = 1;
                    for (auto &pp : per) mx = std::max(mx, pp.size());
                    const int blk = lsm_lif::RING_ENT_BLOCK;
                    const int einw = (int)((mx + blk - 1) / blk * blk);
                    // padding entry at index e: channel 0 into dump word e % 64 -- every lane its OWN dump word (64 lanes
                    // adding to one LDS word serialise: zero-padding cost cfg4 5 ms, profiles/r04_ring_input_drive.txt)
                    std::vector<uint32_t> ent((size_t)wpc * einw);
                    for (size_t e = 0; e < ent.size(); ++e) ent[e] = (uint32_t)((e % 64) * 4);
                    for (int w = 0; w < wpc; ++w)
                        std::copy(per[w].begin(), per[w].end(), ent.begin() + (size_t)w * einw);
                    RingVariant &v = h->rvar[vi];
                    if ((rc = upload(&v.rem_ptr, rptr)) || (rc = upload(&v.rem, rem)) || (rc = upload(&v.leak, lk)) ||
                        (rc = upload(&v.oslot, os)) || (rc = upload(&v.in_ent, ent))) {
                        free_reservoir(h);
                        return rc;
                    }
                    // INMASK form of the kernel (lif_ring.h): per-neuron channel masks in registers instead of the entry drive
#if LSM_EXPERIMENT_HOOKS
                    static const bool no_ring_mask = [] { const char *e = getenv("LSM_RING_NO_INMASK"); return e && atoi(e) != 0; }();
#else
                    constexpr bool no_ring_mask = false;
#endif
                    if (!no_ring_mask && C <= 128 && h->leak_uniform && strided && ql <= 2) {
                        std::vector<uint32_t> im((size_t)npad * 4, 0u);
                        bool distinct = true;
                        for (int c = 0; c < C; ++c)
                            for (int d = 0; d < in_fanout; ++d) {
                                const int tgt = in_tgt[(size_t)c * in_fanout + d];
                                const int pc = coloured ? inperm[c] : c;
                                uint32_t &word = im[(size_t)tgt * 4 + (pc >> 5)];
                                distinct = distinct && !(word & (1u << (pc & 31)));
                                word |= 1u << (pc & 31);
                            }
                        if (distinct) {
                            if ((rc = upload(&v.inmask, im))) { free_reservoir(h); return rc; }
                            v.incol = coloured;
                        }
                    }
                    v.wpc = wpc; v.ql = ql; v.einw = einw; v.strided = strided; v.n_rem = rptr.back();
                }
                // Pair blocks (lif_pair.h): the same windows shared out in 128-neuron blocks, wave w owning the blocks
                // w, w+wpc, ...: the residues must survive the ring's wrap (2*NQ % wpc == 0), a window must not touch more
                // than wpc blocks, a wave at most four blocks (8 neurons per lane).  The kernel counts the input drive from
                // per-neuron channel masks only: C <= 128.
#if LSM_EXPERIMENT_HOOKS
                static const bool no_pairs = [] { const char *e = getenv("LSM_RING_NO_PAIRS"); return e && atoi(e) != 0; }();
#else
                constexpr bool no_pairs = false;
#endif
                const int NB = 2 * NQ;
                const int pwpcs[3] = {4, 8, 16};
                for (int vi = 0; vi < 3 && !no_pairs && C <= 128; ++vi) {
                    const int wpc = pwpcs[vi];
                    std::vector<uint32_t> rptr;
                    std::vector<uint2> rem;
                    const int bl = pair_lists(N, wpc, csc_ptr, csc_post, csc_w, geo, &rptr, &rem);
                    if (!bl) continue;
                    const int npad = NB * 128;
                    // the records carry device addresses (low halves): upload the lists first; a table that crosses a 4 GB
                    // line (the kernel takes the high address bits from the table pointer) cannot serve this layout
                    PairVariant &v = h->pvar[vi];
                    if ((rc = upload(&v.rem, rem))) { free_reservoir(h); return rc; }
                    const uint64_t band_a = reinterpret_cast<uint64_t>(h->band), rem_a = reinterpret_cast<uint64_t>(v.rem);
                    if ((band_a >> 32) != ((band_a + (uint64_t)N * pitch) >> 32) ||
                        (rem_a >> 32) != ((rem_a + (uint64_t)rem.size() * 8u) >> 32)) {
                        (void)hipFree(v.rem);
                        v.rem = nullptr;
                        continue;
                    }
                    std::vector<uint4> rec;
                    pair_records(N, wpc, geo, rptr, band_a, rem_a, &rec);
                    std::vector<int> os(npad, -1);
                    for (int o = 0; o < n_out; ++o) os[out_idx[o]] = o;
                    std::vector<uint32_t> im((size_t)npad * 4, 0u);
                    bool distinct = true;
                    for (int c = 0; c < C; ++c)
                        for (int d = 0; d < in_fanout; ++d) {
                            const int tgt = in_tgt[(size_t)c * in_fanout + d];
                            const int pc = coloured ? inperm[c] : c;
                            uint32_t &word = im[(size_t)tgt * 4 + (pc >> 5)];
                            distinct = distinct && !(word & (1u << (pc & 31)));
                            word |= 1u << (pc & 31);
                        }
                    if (!distinct) { (void)hipFree(v.rem); v.rem = nullptr; continue; }
                    if ((rc = upload(&v.rec, rec)) || (rc = upload(&v.oslot, os)) || (rc = upload(&v.inmask, im))) {
                        free_reservoir(h);
                        return rc;
                    }
                    if (!h->leak_uniform) {                              // a coefficient per neuron (LEAKV form of the kernel)
                        std::vector<float> lk(npad, 0.0f);
                        for (int i = 0; i < N; ++i) lk[i] = leak[i];
                        if ((rc = upload(&v.leak, lk))) { free_reservoir(h); return rc; }
                    }
                    v.incol = coloured;
                    v.wpc = wpc; v.bl = bl; v.n_rem = rptr.back();
                }
            }
        }
    }
    // dense rows now, unless auto mode will serve this reservoir with ring rows: the table exceeds the L2 caches, ring
    // rows exist AND one of their layouts fits a CU's LDS with room to spare (probed at 1024 time steps; the
    // reference runs 400).  Then the 64-262 MB table and its upload would be dead weight next to the ring table
    // (ADVICE r2); lsm_reservoir_set_kernel(2) still builds it on request.  When no ring layout fits (e.g. N = 8000
    // with 5000 output neurons) the dense kernel is auto mode's fallback and needs its table from the start.
    {
        bool ring_serves = false;
        if (has_ring(h) && (size_t)N * (size_t)h->ld * 4 > RING_AUTO_MIN_DENSE_BYTES)
            for (const auto &v : h->rvar)
                if (v.wpc && ring_lds_bytes(h, v, 1024) <= 160 * 1024) ring_serves = true;
        // (a pair-block layout implies strided quad layouts of the same table exist; the quads decide)
        if (!ring_serves && (rc = ensure_dense_rows(h))) { free_reservoir(h); return rc; }
    }
    *out = h;
    return LSM_OK;
}

// 0 = choose, 1 = sparse CSC kernel, 2 = dense-row kernel, 3 = ring-row kernel.
extern "C" __attribute__((visibility("default")))
int lsm_reservoir_set_kernel(lsm_reservoir *h, int mode)
{
    LSM_REQUIRE(h != nullptr, "lsm_reservoir_set_kernel: null handle");
    LSM_REQUIRE(mode >= 0 && mode <= 6, "mode must be 0 (auto), 1 (sparse), 2 (dense), 3 (ring), 4 (ring, contiguous quads), "
                "5 (ring, pair blocks) or 6 (ring, quads)");
    LSM_REQUIRE(mode < 3 || has_ring(h), "this reservoir has no ring-row format (not ring-like, or too small)");
    LSM_REQUIRE(mode != 5 || has_pairs(h), "this reservoir has no pair-block ring layout (needs at most 128 channels, a block "
                "count that is a multiple of 4, 8 or 16 waves and a window of at most that many blocks)");
    if (mode == 2) {                       // an explicit request builds the table a ring-served reservoir deferred
        const int rc = ensure_dense_rows(h);
        if (rc) return rc;
        LSM_REQUIRE(h->wt != nullptr, "this reservoir has no dense row table");
    }
    h->mode = mode;
    return LSM_OK;
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_destroy(lsm_reservoir *h) { return free_reservoir(h); }

static bool lif_inreg(const Variant &v) { return v.einw <= IN_REG_SLOTS * 64; }

static bool ring_inreg(const RingVariant &rv)
{
    // the wave's input map stays in registers when it is one block (8 registers) AND the layout has few neurons per
    // lane: with three or four quads per wave the extra registers would cost a wave per SIMD
    bool inreg = rv.einw <= lsm_lif::RING_ENT_BLOCK * lsm_lif::RING_ENT_REG_BLOCKS && rv.ql <= 2 && rv.strided;
#if LSM_EXPERIMENT_HOOKS
    static const bool no_inreg = [] { const char *e = getenv("LSM_RING_NO_INREG"); return e && atoi(e) != 0; }();
    if (no_inreg) inreg = false;
#endif
    return inreg;
}


static size_t lif_lds_core(const lsm_reservoir *h, const Variant &v, int T)
{
    const size_t npad = (size_t)v.sl * 64 * v.wpc;
    const size_t cw = (size_t)(h->C + 31) / 32;
    return npad * 4 + npad * 4 + 2 * npad * 2 + 128 + (size_t)h->n_out * 16 + (size_t)T * cw * 4;
}

static size_t lif_seg_bytes(const lsm_reservoir *h, const Variant &v)
{
    // packed LDS form: (N+1) u32 row pointers + N*(WPC+1) u16 offsets (rounded up to a whole dword)
    return ((size_t)h->N + 1) * 4 + (((size_t)h->N * (v.wpc + 1) + 1) / 2) * 4;
}

// The segment table rides in LDS when that still leaves room for >= 2 workgroups per CU's 160 KB.
static bool lif_seg_in_lds(const lsm_reservoir *h, const Variant &v, int T)
{
    return lif_lds_core(h, v, T) + lif_seg_bytes(h, v) <= 80 * 1024;
}

static size_t lif_lds_bytes(const lsm_reservoir *h, const Variant &v, int T)
{
    return lif_lds_core(h, v, T) + (lif_seg_in_lds(h, v, T) ? lif_seg_bytes(h, v) : 0);
}

static size_t ring_lds_bytes(const lsm_reservoir *h, const RingVariant &v, int T)
{
    const size_t npad = (size_t)v.ql * 256 * v.wpc;
    const size_t cw = (size_t)(h->C + 31) / 32;
    // float32 accumulators + 16-bit input counts (each behind 64 dump words; no counts when the kernel counts from
    // per-neuron masks, INMASK), two step lists, quad counts, feature accumulators, the clip's input bits
    const size_t cnt_words = v.inmask ? 0 : (size_t)lsm_lif::RING_DUMP_WORDS + npad / 2;
    return ((size_t)lsm_lif::RING_DUMP_WORDS + npad + cnt_words) * 4 +
           2 * npad * 2 + 512 + (size_t)h->n_out * 16 + (size_t)T * cw * 4;
}

static size_t pair_lds_bytes(const lsm_reservoir *h, const PairVariant &v, int T)
{
    const size_t npad = (size_t)v.bl * 128 * v.wpc;
    const size_t cw = (size_t)(h->C + 31) / 32;
    // dump words, float32 accumulators, 64 scratch words per wave, two step lists of bytes, block counts, feature
    // accumulators, the clip's input bits
    return (size_t)lsm_lif::PAIR_DUMP_BYTES + npad * 4 + (size_t)v.wpc * 256 + 2 * npad +
           (size_t)lsm_lif::PAIR_WCNT_WORDS * 4 + (LSM_PAIR_LEAN ? 0 : (size_t)h->n_out * 16) + (size_t)T * cw * 4 +
           (size_t)LSM_PAIR_LDS_PAD;
}

// Pair-block layout for a batch: the requested waves per clip, else the fewest waves (every wave repeats the per-row work).
static const PairVariant *choose_pair(const lsm_reservoir *h, int T, int requested)
{
    if (h->mode != 0 && h->mode != 3 && h->mode != 5) return nullptr;
    for (const auto &v : h->pvar) {
        if (!v.wpc || pair_lds_bytes(h, v, T) > 160 * 1024) continue;
        if (requested > 0 && v.wpc != requested) continue;
        // Unless pair blocks are asked for by name, they are taken where they pay: with at least three blocks per wave.
        // Same box, 512 clips, k = 0.2 N (profiles/r05_midsize_pairs_vs_quads.txt): N = 1536 (3 blocks, 4 waves) 1.33 ms
        // against 2.95 in quads, N = 4000 (4 blocks) 4.17 against 5.10, N = 3072 (3 blocks, 8 waves) 3.47 against 3.45 --
        // but N = 2048 (2 blocks, 8 waves) 2.46 against 2.19: with few neurons per lane the per-wave work per row dominates
        // and the quads' four fat waves win.
        if (h->mode != 5 && v.bl < 3 && has_ring(h)) continue;
        return &v;
    }
    return nullptr;
}

// Ring layout for a batch: the requested waves per clip, else a strided layout when the reservoir has one (every
// wave gets one useful 1 KB window load per row: N=8000, 8 waves: 29.0 ms against 42.0 ms contiguous; N=4000:
// 8 waves strided 9.7 ms against 4 fat waves contiguous 10.2 ms), and within a kind the layout with the fewest
// waves that still gives a clip at least 4 (every wave repeats the per-row work -- list hand-off, row
// parameters: N=8000 strided, 8 waves 29.0 ms, 16 waves 36.8 ms; N=4000 contiguous, 4 waves 10.2 ms, 8 waves 13.9 ms).
static const RingVariant *choose_ring(const lsm_reservoir *h, int T, int requested)
{
    const RingVariant *best = nullptr;
    auto better = [](const RingVariant &v, const RingVariant &b) {
        if (v.strided != b.strided) return v.strided;
        const bool v4 = v.wpc >= 4, b4 = b.wpc >= 4;
        if (v4 != b4) return v4;
        return v4 ? v.wpc < b.wpc : v.wpc > b.wpc;
    };
    if (h->mode == 5) return nullptr;             // pair blocks only
    for (const auto &v : h->rvar) {
        if (!v.wpc || ring_lds_bytes(h, v, T) > 160 * 1024) continue;
        if (v.strided && h->mode == 4) continue;
        if (requested > 0 && v.wpc != requested) continue;
        if (!best || better(v, *best)) best = &v;
    }
    return best;
}

static bool want_ring(const lsm_reservoir *h)
{
    if (!has_ring(h)) return false;
    if (h->mode >= 3) return true;
    return h->mode == 0 && (size_t)h->N * (size_t)h->ld * 4 > RING_AUTO_MIN_DENSE_BYTES;
}
static bool use_dense(const lsm_reservoir *h) { return h->wt != nullptr && h->mode != 1; }

static size_t dense_lds_bytes(const lsm_reservoir *h, const Variant &v, int T)
{
    const size_t npad = (size_t)v.sl * 64 * v.wpc;
    const size_t cw = (size_t)(h->C + 31) / 32;
    return npad * 4 + 2 * npad * 2 + 512 + (size_t)h->n_out * 16 + (size_t)T * cw * 4;
}

// Pick the waves-per-clip layout.  Measured on MI355X at N=1000 (profiles/): the best layout has
// about 4096 wavefronts in flight (B=256 -> 16, B=512 -> 8) and never fewer than 4 waves per clip
// (B=4096: 4 waves 9.1 ms, 2 waves 10.6 ms, 1 wave 17.6 ms).  Among the layouts this reservoir
// supports and whose LDS image fits one CU, take the smallest one at or above that target.
// requested == -1: the launch runs inside an overlapped pipeline (other kernels share the CUs): the chip is
// then bound by vector-ALU issue and fewer, fatter waves spend fewer instructions on per-wave overheads
// (N=1000, B=256: 4 waves per clip 3.5 % faster for the whole pipeline than 8; a lone launch prefers 8).
static const Variant *choose_variant(const lsm_reservoir *h, int B, int T, int requested)
{
    // LDS image of the kernel that will run (the dense-row kernel needs less than the sparse one)
    auto lds_of = [&](const Variant &v) {
        return use_dense(h) ? dense_lds_bytes(h, v, T) : lif_lds_bytes(h, v, T);
    };
    if (requested > 0) {
        for (const auto &v : h->var)
            if (v.wpc == requested && lds_of(v) <= 160 * 1024) return &v;
        return nullptr;
    }
    int target = 4;
    // dense-row kernel (measured at N=1000, B=256: 4 waves 0.82 ms, 8 waves 0.75 ms, 16 waves 0.81 ms):
    // about 2048 wavefronts; sparse kernel: about 4096
    const long want = use_dense(h) ? 2048 : 4096;
    if (requested == 0)
        while (target < 16 && (long)B * target < want) target <<= 1;
    // large reservoirs: more waves per clip keep the per-lane neuron slots (registers, update work per
    // wave) small -- N=4000, B=1024: 4 waves 68 ms, 8 waves 43 ms, 16 waves 26 ms
    const Variant *best = nullptr;
    for (const auto &v : h->var) {
        if (!v.wpc || lds_of(v) > 160 * 1024) continue;
        if (!best || best->wpc < target || best->sl > 4) best = &v;
    }
    return best;
}

extern "C" __attribute__((visibility("default")))
long lsm_reservoir_order_workspace(int n_clips)
{
    return n_clips > 0 ? 2L * n_clips * (long)sizeof(int32_t) : 0L;     // spike count and start order of every clip
}

// ---- longest clips first (lsm_reservoir_run_ordered) --------------------------------------------------------------
// A clip's time in the LIF kernels grows with its activity (rows per step), and activity follows the input: at
// N = 4000 the input spike count of a clip predicts its reservoir spikes with a rank correlation of 0.99, and clips of
// one batch differ by a factor of 9.  A launch of several rounds (more clips than the chip holds at once) is then as
// long as the clip that happens to start last: 8.8 ms for 1024 clips whose work fills the chip for 5.1 ms.  Workgroups
// are dispatched in index order, so handing workgroup g the clip with the g-th most input spikes is the classic
// longest-processing-time-first schedule.  Results do not depend on it: every clip is simulated on its own.
__global__ __launch_bounds__(256) void clip_keys_kernel(const uint8_t *raster, long bytes_per_clip, int n_clips,
                                                        int32_t *keys)
{
    const int b = blockIdx.x;
    const uint8_t *clip = raster + (size_t)b * (size_t)bytes_per_clip;
    uint32_t sum = 0;
    long i = threadIdx.x * 16L;
    if ((reinterpret_cast<uintptr_t>(clip) & 15) == 0) {
        for (; i + 16 <= bytes_per_clip; i += 256 * 16L) {
            const uint4 v = *reinterpret_cast<const uint4 *>(clip + i);
            // bytes are 0 or 1 (anything else counts as a spike in the LIF kernels too: count non-zero bytes)
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t nz = w4[k] | (w4[k] >> 4);
                nz |= nz >> 2; nz |= nz >> 1;
                sum += __popc(nz & 0x01010101u);
            }
        }
        // the bytes behind the last full 16-byte group of the whole clip
        if (threadIdx.x == 0)
            for (long t = bytes_per_clip & ~15L; t < bytes_per_clip; ++t) sum += clip[t] != 0;
    } else {
        for (long t = threadIdx.x; t < bytes_per_clip; t += 256) sum += clip[t] != 0;
    }
    __shared__ uint32_t part[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor((int)sum, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) keys[b] = (int32_t)(part[0] + part[1] + part[2] + part[3]);
}

constexpr int ORDER_WINDOW = 4096;      // clips ranked against each other (bounds the quadratic ranking)
static_assert(ORDER_WINDOW % 256 == 0, "a ranking workgroup stays inside one window");

// order[first + rank] = clip, rank = position of the clip in its window by (key descending, index ascending).
// A workgroup ranks 256 clips of one window; the window's keys pass through LDS in tiles of 1024, read back 16 bytes
// at a time at the same address in every lane (first version: every lane walked the keys in global memory, 63 us for
// 1024 clips -- 1 % of the cfg4 launch it schedules; now 28 us and less).
__global__ __launch_bounds__(256) void clip_rank_kernel(const int32_t *keys, int n_clips, int32_t *order)
{
    __shared__ __attribute__((aligned(16))) int32_t tile[1024];
    const int i = blockIdx.x * 256 + threadIdx.x;           // ORDER_WINDOW is a multiple of 256: a workgroup never straddles two windows
    const int first = (blockIdx.x * 256) / ORDER_WINDOW * ORDER_WINDOW;
    const int last = min(n_clips, first + ORDER_WINDOW);
    const bool mine = i < n_clips;
    const int32_t ki = mine ? keys[i] : 0;
    int rank = 0;
    for (int t0 = first; t0 < last; t0 += 1024) {
        __syncthreads();
        for (int q = threadIdx.x; q < 1024; q += 256) tile[q] = t0 + q < last ? keys[t0 + q] : INT32_MIN;
        __syncthreads();
        // INT32_MIN pads the tile: no key (a count >= 0) is smaller or equal, so padding never adds to a rank
        const int n4 = (min(1024, last - t0) + 3) / 4;
#pragma unroll 8
        for (int q = 0; q < n4; ++q) {
            const int4 k4 = reinterpret_cast<const int4 *>(tile)[q];
            const int j = t0 + 4 * q;
            rank += (k4.x > ki) || (k4.x == ki && j < i);
            rank += (k4.y > ki) || (k4.y == ki && j + 1 < i);
            rank += (k4.z > ki) || (k4.z == ki && j + 2 < i);
            rank += (k4.w > ki) || (k4.w == ki && j + 3 < i);
        }
    }
    if (mine) order[first + rank] = i;
}

// ONE decision for lsm_reservoir_run, lsm_reservoir_layout, lsm_reservoir_plan and lsm_reservoir_kernel_in_use:
// which kernel and which layout serve (handle, batch, steps, waves_per_clip).  Auto mode prefers ring rows for
// large ring-like reservoirs but FALLS THROUGH to the dense (else sparse) kernel when no ring layout fits -- the
// per-clip LDS image exceeds 160 KB (e.g. N = 8000 with 5000 output neurons: 175.7 KB as ring rows, 158.8 KB
// for the dense/sparse layouts) or the reservoir's ring layouts lack the requested waves per clip.
// Only an explicit ring request (modes 3, 4) is a hard error then.
struct RunPlan {
    int kernel = 0;                   // 1 sparse, 2 dense rows, 3 ring rows
    const PairVariant *pv = nullptr;  // ring rows in pair blocks (lif_pair.h) when set, else
    const RingVariant *rv = nullptr;
    const Variant *v = nullptr;
};

static int make_plan(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip, RunPlan *p)
{
    if (want_ring(h)) {
        // pair blocks where the reservoir has them (cfg4: 5.1 -> see profiles/r05_*), else quads
        p->pv = choose_pair(h, n_steps, waves_per_clip > 0 ? waves_per_clip : 0);
        if (p->pv) { p->kernel = 3; return LSM_OK; }
        p->rv = choose_ring(h, n_steps, waves_per_clip > 0 ? waves_per_clip : 0);
        if (p->rv) { p->kernel = 3; return LSM_OK; }
        LSM_REQUIRE(h->mode < 3, "no ring-row layout for waves_per_clip=%d (N=%d, n_out=%d, T=%d): none fits a CU's "
                    "160 KB of LDS or has that many waves", waves_per_clip, h->N, h->n_out, n_steps);
    }
    p->v = choose_variant(h, n_clips, n_steps, waves_per_clip);
    LSM_REQUIRE(p->v != nullptr, "no reservoir layout for waves_per_clip=%d (N=%d, n_out=%d, T=%d)",
                waves_per_clip, h->N, h->n_out, n_steps);
    p->kernel = use_dense(h) ? 2 : 1;
    return LSM_OK;
}

static int reservoir_run(const lsm_reservoir *h, const uint8_t *spikes_u8, int n_clips, int n_steps,
                         const int32_t *key_ids, int n_keys, float *features_out,
                         uint8_t *spike_matrix_out, float *v_trace_out, int32_t *stats_out,
                         int waves_per_clip, void *workspace, long workspace_bytes, void *stream)
{
    LSM_REQUIRE(h != nullptr, "lsm_reservoir_run: null handle");
    LSM_REQUIRE(n_clips >= 0 && n_steps >= 1 && n_steps <= 65535, "bad n_clips/n_steps");
    LSM_REQUIRE(n_keys >= 1 && n_keys <= 8 && key_ids, "n_keys must be in [1, 8]");
    LSM_REQUIRE(waves_per_clip >= -1 && waves_per_clip <= 16, "waves_per_clip must be -1 (pipelined), 0 (choose) or 1..16");
    if (n_clips == 0) return LSM_OK;            // empty batch: nothing to read or write
    LSM_REQUIRE(spikes_u8 && features_out, "null buffer");
    int dev_now = -1;
    LSM_CHECK_HIP(hipGetDevice(&dev_now));
    LSM_REQUIRE(dev_now == h->device, "reservoir handle lives on device %d but the current device is %d",
                h->device, dev_now);
    for (int k = 0; k < n_keys; ++k)
        LSM_REQUIRE(key_ids[k] >= 0 && key_ids[k] < 8, "key id %d out of range", key_ids[k]);
    const size_t cw = (size_t)(h->C + 31) / 32;

    RunPlan plan;
    {
        const int prc = make_plan(h, n_clips, n_steps, waves_per_clip, &plan);
        if (prc) return prc;
    }
    // longest clips first, when the launch has more workgroups than the chip has CUs (else every clip starts at once)
    const int32_t *order = nullptr;
    if (workspace) {
        LSM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 3) == 0, "workspace must be 4-byte aligned");
        LSM_REQUIRE(workspace_bytes >= lsm_reservoir_order_workspace(n_clips),
                    "workspace of %ld bytes, need %ld (lsm_reservoir_order_workspace)", workspace_bytes,
                    lsm_reservoir_order_workspace(n_clips));
        if (h->cus > 0 && n_clips > h->cus) {
            int32_t *keys = static_cast<int32_t *>(workspace);
            int32_t *ord = keys + n_clips;
            hipLaunchKernelGGL(clip_keys_kernel, dim3(n_clips), dim3(256), 0, (hipStream_t)stream, spikes_u8,
                               (long)h->C * n_steps, n_clips, keys);
            hipLaunchKernelGGL(clip_rank_kernel, dim3((n_clips + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                               keys, n_clips, ord);
            LSM_CHECK_HIP(hipGetLastError());
            order = ord;
        }
    }
    if (plan.kernel == 3 && plan.pv) {
        const PairVariant *pv = plan.pv;
        const int inmask = pv->incol ? 2 : 1;
        const bool leakv = pv->leak != nullptr;
        lsm_lif::pair_fn_t pfn = pv->bl == 1 ? lsm_lif::pick_pair_1(pv->wpc, inmask, leakv)
                                 : pv->bl == 2 ? lsm_lif::pick_pair_2(pv->wpc, inmask, leakv)
                                 : pv->bl == 3 ? lsm_lif::pick_pair_3(pv->wpc, inmask, leakv)
                                               : lsm_lif::pick_pair_4(pv->wpc, inmask, leakv);
        LSM_REQUIRE(pfn != nullptr, "no pair-block ring kernel for BL=%d WPC=%d", pv->bl, pv->wpc);
        lsm_lif::PairArgs r;
        r.N = h->N; r.C = h->C; r.T = n_steps; r.B = n_clips;
        r.n_out = h->n_out; r.CW = (int)cw;
        r.refractory = h->refractory; r.burst_isi_max = h->burst_isi_max;
        r.theta = h->theta; r.w_in = h->w_in; r.leak_u = h->leak_u;
        r.raster = spikes_u8; r.band = h->band; r.rec = pv->rec; r.rem = pv->rem;
        r.oslot = pv->oslot; r.leak = pv->leak; r.inmask = pv->inmask; r.inperm = pv->incol ? h->inperm : nullptr;
        r.n_keys = n_keys;
        for (int k = 0; k < 8; ++k) r.key_ids[k] = k < n_keys ? key_ids[k] : 0;
        r.features = features_out; r.spike_matrix = spike_matrix_out; r.v_trace = v_trace_out;
        r.stats = stats_out; r.order = order;
        const size_t lds = pair_lds_bytes(h, *pv, n_steps);
        if (lds > 64 * 1024) lsm_allow_big_lds(reinterpret_cast<const void *>(pfn));
        hipLaunchKernelGGL(pfn, dim3(n_clips), dim3(pv->wpc * 64), lds, (hipStream_t)stream, r);
        LSM_CHECK_HIP(hipGetLastError());
        return LSM_OK;
    }
    if (plan.kernel == 3) {
        const RingVariant *rv = plan.rv;
        const bool inreg = ring_inreg(*rv);
        const bool inmask = rv->inmask != nullptr;
        lsm_lif::ring_fn_t rfn = inmask ? (rv->ql == 1 ? lsm_lif::pick_ring_mask_1(rv->wpc, rv->incol ? 2 : 1)
                                                       : lsm_lif::pick_ring_mask_2(rv->wpc, rv->incol ? 2 : 1))
                                 : rv->ql == 1 ? lsm_lif::pick_ring_1(rv->wpc, inreg, rv->strided)
                                 : rv->ql == 2 ? lsm_lif::pick_ring_2(rv->wpc, inreg, rv->strided)
                                 : rv->ql == 3 ? lsm_lif::pick_ring_3(rv->wpc, inreg, rv->strided)
                                               : lsm_lif::pick_ring_4(rv->wpc, inreg, rv->strided);
        LSM_REQUIRE(rfn != nullptr, "no ring kernel for QL=%d WPC=%d", rv->ql, rv->wpc);
        lsm_lif::RingArgs r;
        r.N = h->N; r.C = h->C; r.T = n_steps; r.B = n_clips;
        r.n_out = h->n_out; r.CW = (int)cw; r.EinW = rv->einw;
        r.refractory = h->refractory; r.burst_isi_max = h->burst_isi_max;
        r.H = h->band_h; r.NQ = h->band_nq; r.pitch = h->band_pitch;
        r.theta = h->theta; r.w_in = h->w_in;
        r.raster = spikes_u8; r.band = h->band; r.rem_ptr = rv->rem_ptr; r.rem = rv->rem;
        r.leak = rv->leak; r.oslot = rv->oslot; r.in_ent = rv->in_ent;
        r.inmask = rv->inmask; r.inperm = (inmask && rv->incol) ? h->inperm : nullptr; r.leak_u = h->leak_u;
        r.n_keys = n_keys;
        for (int k = 0; k < 8; ++k) r.key_ids[k] = k < n_keys ? key_ids[k] : 0;
        r.features = features_out; r.spike_matrix = spike_matrix_out; r.v_trace = v_trace_out;
        r.stats = stats_out; r.order = order;
        const size_t lds = ring_lds_bytes(h, *rv, n_steps);
        if (lds > 64 * 1024) lsm_allow_big_lds(reinterpret_cast<const void *>(rfn));
        hipLaunchKernelGGL(rfn, dim3(n_clips), dim3(rv->wpc * 64), lds, (hipStream_t)stream, r);
        LSM_CHECK_HIP(hipGetLastError());
        return LSM_OK;
    }
    const Variant *v = plan.v;
    if (plan.kernel == 2) {
        // the reference's refractory period (2 steps) counts down in scalar lane masks (lif_dense.h, REFM);
        // LSM_DENSE_NO_REFM (diagnostic builds) keeps the vector-register countdown for same-box A/B runs
        bool refm = h->refractory == lsm_lif::DENSE_REFM_REFRACTORY;
#if LSM_EXPERIMENT_HOOKS
        static const bool no_refm = [] { const char *e = getenv("LSM_DENSE_NO_REFM"); return e && atoi(e) != 0; }();
        if (no_refm) refm = false;
#endif
        lsm_lif::dense_fn_t dfn = v->inmask       ? (v->incol ? lsm_lif::pick_dense_3(v->sl, v->wpc, refm)
                                                              : lsm_lif::pick_dense_2(v->sl, v->wpc, refm))
                                  : lif_inreg(*v) ? lsm_lif::pick_dense_1(v->sl, v->wpc, refm)
                                                  : lsm_lif::pick_dense_0(v->sl, v->wpc, refm);
        LSM_REQUIRE(dfn != nullptr, "no dense kernel for SL=%d WPC=%d", v->sl, v->wpc);
        lsm_lif::DenseArgs d;
        d.N = h->N; d.C = h->C; d.T = n_steps; d.B = n_clips;
        d.n_out = h->n_out; d.CW = (int)cw; d.EinW = v->einw;
        d.refractory = h->refractory; d.burst_isi_max = h->burst_isi_max; d.ld = h->ld;
        d.theta = h->theta; d.w_in = h->w_in;
        d.raster = spikes_u8; d.wt = h->wt; d.leak = v->leak; d.oslot = v->oslot; d.in_ent = v->in_ent;
        d.inmask = v->inmask;
        d.inperm = v->incol ? h->inperm : nullptr;
        d.n_keys = n_keys;
        for (int k = 0; k < 8; ++k) d.key_ids[k] = k < n_keys ? key_ids[k] : 0;
        d.features = features_out; d.spike_matrix = spike_matrix_out; d.v_trace = v_trace_out;
        d.stats = stats_out; d.order = order;
        const size_t dlds = dense_lds_bytes(h, *v, n_steps);
        LSM_REQUIRE(dlds <= 160 * 1024, "dense layout needs %zu bytes of LDS", dlds);
        if (dlds > 64 * 1024) lsm_allow_big_lds(reinterpret_cast<const void *>(dfn));
        hipLaunchKernelGGL(dfn, dim3(n_clips), dim3(v->wpc * 64), dlds, (hipStream_t)stream, d);
        LSM_CHECK_HIP(hipGetLastError());
        return LSM_OK;
    }
    lif_fn_t fn = pick_kernel(v->sl, v->wpc, lif_inreg(*v), lif_seg_in_lds(h, *v, n_steps));
    LSM_REQUIRE(fn != nullptr, "no kernel for SL=%d WPC=%d", v->sl, v->wpc);

    LifArgs a;
    a.N = h->N; a.C = h->C; a.T = n_steps; a.B = n_clips;
    a.n_out = h->n_out; a.CW = (int)cw; a.EinW = v->einw;
    a.refractory = h->refractory; a.burst_isi_max = h->burst_isi_max;
    a.theta = h->theta; a.w_in = h->w_in;
    a.rowptr = h->rowptr; a.segoff = v->segoff;
    a.raster = spikes_u8; a.seg = v->seg; a.syn = h->syn; a.leak = v->leak; a.oslot = v->oslot;
    a.in_ent = v->in_ent;
    a.n_keys = n_keys;
    for (int k = 0; k < 8; ++k) a.key_ids[k] = k < n_keys ? key_ids[k] : 0;
    a.features = features_out; a.spike_matrix = spike_matrix_out; a.v_trace = v_trace_out;
    a.stats = stats_out; a.order = order;

    const size_t lds = lif_lds_bytes(h, *v, n_steps);
    if (lds > 64 * 1024) lsm_allow_big_lds(reinterpret_cast<const void *>(fn));
    hipLaunchKernelGGL(fn, dim3(n_clips), dim3(v->wpc * 64), lds, (hipStream_t)stream, a);
    LSM_CHECK_HIP(hipGetLastError());
    return LSM_OK;
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_run(const lsm_reservoir *h, const uint8_t *spikes_u8, int n_clips, int n_steps,
                      const int32_t *key_ids, int n_keys, float *features_out,
                      uint8_t *spike_matrix_out, float *v_trace_out, int32_t *stats_out,
                      int waves_per_clip, void *stream)
{
    return reservoir_run(h, spikes_u8, n_clips, n_steps, key_ids, n_keys, features_out, spike_matrix_out, v_trace_out,
                         stats_out, waves_per_clip, nullptr, 0, stream);
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_run_ordered(const lsm_reservoir *h, const uint8_t *spikes_u8, int n_clips, int n_steps,
                              const int32_t *key_ids, int n_keys, float *features_out,
                              uint8_t *spike_matrix_out, float *v_trace_out, int32_t *stats_out,
                              int waves_per_clip, void *workspace, long workspace_bytes, void *stream)
{
    LSM_REQUIRE(workspace != nullptr || n_clips == 0, "lsm_reservoir_run_ordered: null workspace");
    return reservoir_run(h, spikes_u8, n_clips, n_steps, key_ids, n_keys, features_out, spike_matrix_out, v_trace_out,
                         stats_out, waves_per_clip, workspace, workspace_bytes, stream);
}

// Introspection for tests and the bench: kernel and layout chosen for a batch, LDS bytes per workgroup, bytes of the
// weight table the kernel gathers its rows from.
extern "C" __attribute__((visibility("default")))
int lsm_reservoir_plan(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip, int *kernel_out,
                       int *wpc_out, int *slots_out, int *lds_bytes_out, long *table_bytes_out)
{
    LSM_REQUIRE(h != nullptr, "lsm_reservoir_plan: null handle");
    LSM_REQUIRE(n_clips >= 0 && n_steps >= 1 && n_steps <= 65535, "bad n_clips/n_steps");
    LSM_REQUIRE(waves_per_clip >= -1 && waves_per_clip <= 16, "waves_per_clip must be -1 (pipelined), 0 (choose) or 1..16");
    RunPlan p;
    const int rc = make_plan(h, n_clips, n_steps, waves_per_clip, &p);
    if (rc) return rc;
    if (kernel_out) *kernel_out = p.kernel;
    if (p.kernel == 3 && p.pv) {
        if (wpc_out) *wpc_out = p.pv->wpc;
        if (slots_out) *slots_out = p.pv->bl * 2;
        if (lds_bytes_out) *lds_bytes_out = (int)pair_lds_bytes(h, *p.pv, n_steps);
        if (table_bytes_out)    // ring windows + this layout's lists + its row records
            *table_bytes_out = (long)((size_t)h->N * h->band_pitch + p.pv->n_rem * 8 + (size_t)h->N * p.pv->wpc * 16);
        return LSM_OK;
    }
    if (p.kernel == 3) {
        if (wpc_out) *wpc_out = p.rv->wpc;
        if (slots_out) *slots_out = p.rv->ql * 4;
        if (lds_bytes_out) *lds_bytes_out = (int)ring_lds_bytes(h, *p.rv, n_steps);
        if (table_bytes_out) {
            // ring windows + this layout's lists of the synapses outside them + their row pointers
            *table_bytes_out = (long)((size_t)h->N * h->band_pitch + p.rv->n_rem * 8 + ((size_t)h->N * p.rv->wpc + 1) * 4);
        }
        return LSM_OK;
    }
    if (wpc_out) *wpc_out = p.v->wpc;
    if (slots_out) *slots_out = p.v->sl;
    if (lds_bytes_out)
        *lds_bytes_out = (int)(p.kernel == 2 ? dense_lds_bytes(h, *p.v, n_steps) : lif_lds_bytes(h, *p.v, n_steps));
    if (table_bytes_out)
        *table_bytes_out = p.kernel == 2 ? (long)((size_t)h->N * h->ld * 4) : (long)(h->nnz * 8 + ((size_t)h->N + 1) * 4);
    return LSM_OK;
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_input_mode(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip)
{
    if (h == nullptr) return LSM_ERR_ARG;
    RunPlan p;
    if (make_plan(h, n_clips, n_steps, waves_per_clip, &p) != LSM_OK) return LSM_ERR_UNSUPPORTED;
    if (p.kernel == 3 && p.pv) return p.pv->incol ? 15 : 14;
    if (p.kernel == 3) return p.rv->inmask ? (p.rv->incol ? 13 : 12) : (ring_inreg(*p.rv) ? 11 : 10);
    if (p.kernel == 1) return 20;
    return p.v->inmask ? (p.v->incol ? 3 : 2) : (lif_inreg(*p.v) ? 1 : 0);
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_row_request_bytes(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip,
                                    double *mean_bytes_out)
{
    LSM_REQUIRE(h != nullptr && mean_bytes_out != nullptr, "lsm_reservoir_row_request_bytes: null argument");
    LSM_REQUIRE(n_clips >= 0 && n_steps >= 1 && n_steps <= 65535, "bad n_clips/n_steps");
    LSM_REQUIRE(waves_per_clip >= -1 && waves_per_clip <= 16, "waves_per_clip must be -1 (pipelined), 0 (choose) or 1..16");
    RunPlan p;
    const int rc = make_plan(h, n_clips, n_steps, waves_per_clip, &p);
    if (rc) return rc;
    const double n = (double)h->N;
    if (p.kernel == 3 && p.pv)
        *mean_bytes_out = (h->band_bytes_sum + (double)p.pv->n_rem * 8.0) / n + p.pv->wpc * 16.0;
    else if (p.kernel == 3)
        *mean_bytes_out = (h->band_bytes_sum + (double)p.rv->n_rem * 8.0) / n + (p.rv->wpc + 1) * 4.0;
    else if (p.kernel == 2)
        *mean_bytes_out = (double)h->ld * 4.0;
    else
        *mean_bytes_out = (double)h->nnz * 8.0 / n + 8.0;
    return LSM_OK;
}

extern "C" __attribute__((visibility("default")))
int lsm_reservoir_layout(const lsm_reservoir *h, int n_clips, int n_steps, int waves_per_clip,
                         int *wpc_out, int *slots_out, int *lds_bytes_out)
{
    return lsm_reservoir_plan(h, n_clips, n_steps, waves_per_clip, nullptr, wpc_out, slots_out, lds_bytes_out, nullptr);
}

// Which kernel lsm_reservoir_run would launch for this handle: 1 sparse, 2 dense rows, 3 ring rows -- for the
// reference's 400 time steps and waves_per_clip = 0 (lsm_reservoir_plan answers for any launch).
extern "C" __attribute__((visibility("default")))
int lsm_reservoir_kernel_in_use(const lsm_reservoir *h)
{
    if (h == nullptr) return LSM_ERR_ARG;
    RunPlan p;
    if (make_plan(h, 1, 400, 0, &p) != LSM_OK) return want_ring(h) ? 3 : (use_dense(h) ? 2 : 1);
    return p.kernel;
}

// Diagnostic builds (-DLSM_STAMP=1) only: per-phase cycle sums of the LIF kernel; zeros otherwise.
extern "C" __attribute__((visibility("default")))
int lsm_debug_lif_stamps(unsigned long long *out8, int reset)
{
    for (int k = 0; k < 8; ++k) out8[k] = 0;
#if LSM_STAMP
    for (int q = 0; q < 6; ++q) {                 // the four sparse-kernel units + the dense kernel's INMODE-2 and -3 units
        unsigned long long part[8];
        int rc = q < 4 ? lsm_lif::read_lif_stamps(q, part, reset)
                       : (q == 4 ? lsm_lif::read_lif_stamps_d2(part, reset) : lsm_lif::read_lif_stamps_d3(part, reset));
        if (rc) return rc;
        for (int k = 0; k < 8; ++k) out8[k] += part[k];
    }
#else
    (void)reset;
#endif
    return LSM_OK;
}
