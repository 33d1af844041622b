"""Index arithmetic of the ring-row format (csrc/lif_ring.h + the table builder in csrc/reservoir.hip),
restated in NumPy and checked exhaustively on the CPU: for every presynaptic row j and every neuron slot of
every wave, the bounds-checked window fetch plus the (row, wave) list must deliver exactly W[j -> i] — once,
from the window or from the list, never both — including wrapped windows, a partly filled last quad and
padding neurons.  (The GPU parity tests then check the kernel that implements this arithmetic.)"""
import numpy as np
import pytest

from lsm_speech_classifier_amd import reservoir as R


def ring_tables(n, csc_ptr, csc_post, csc_w):
    """Host side: H, NQ, per-row geometry, band table (row -> float array of `pitch` bytes), list membership."""
    nnz = int(csc_ptr[-1])
    H = (nnz // n + 1) // 2
    NQ = (n + 255) // 256
    NP = NQ * 256                                       # the ring padded to whole quads
    j = np.arange(n)
    a4 = ((j - H) % n) & ~31                            # 32-aligned (128-byte) first target of the stored row
    b0 = (j + H) % n
    span = (b0 - a4) % NP
    nbytes = ((span >> 2) + 1) * 16
    wsq = int(((((a4 & 255) + span) >> 8) + 1).max())
    pitch = (int(nbytes.max()) + 127) & ~127
    band = np.zeros((n, pitch // 4), dtype=np.float32)
    in_list = np.zeros(nnz, dtype=bool)
    for jj in range(n):
        e0, e1 = csc_ptr[jj], csc_ptr[jj + 1]
        i = csc_post[e0:e1]
        off = ((i - a4[jj]) % NP) * 4
        win = off < nbytes[jj]
        band[jj, off[win] // 4] = csc_w[e0:e1][win]
        in_list[e0:e1] = ~win
    return H, NQ, a4, nbytes, wsq, band, in_list


def fetch_row(jj, wpc, ql, n, NQ, a4, nbytes, band, strided=False):
    """Device side: what the bounds-checked 16-byte loads of every wave return for row jj, as an array over
    the padded neuron index (wpc*ql*256).  Contiguous ownership: QL loads per wave; strided: one."""
    npad = wpc * ql * 256
    out = np.zeros(npad, dtype=np.float32)
    lane16 = np.arange(64, dtype=np.uint32) * 16
    q0, lead = int(a4[jj]) >> 8, int(a4[jj]) & 255

    def load(ring_pos, dest_quad):                        # the quad at ring position p starts at byte (p*256-lead)*4
        byte_off = (ring_pos * 256 - lead) * 4
        voff = (lane16 + np.uint32(byte_off & 0xFFFFFFFF)) & np.uint32(0xFFFFFFFF)
        for h in range(4):
            o = voff.astype(np.uint64) + 4 * h
            ok = o + 4 <= nbytes[jj]                          # per-dword range check of a raw buffer load
            vals = np.where(ok, band[jj, np.minimum(o // 4, band.shape[1] - 1).astype(np.int64)], 0.0)
            out[dest_quad * 256 + np.arange(64) * 4 + h] = vals

    for w in range(wpc):
        if strided:
            ph = (w - q0) % wpc
            gh = (q0 + ph) % NQ
            assert gh % wpc == w                              # the residue survives the wrap (NQ % wpc == 0)
            load(ph, gh)                                      # register quad gh // wpc of wave w = global quad gh
        else:
            g0 = w * ql
            base = g0 - q0 + NQ if g0 + ql - 1 < q0 else g0 - q0
            for q in range(ql):
                load(base + q, g0 + q)
    return out


@pytest.mark.parametrize("n,k,layouts", [
    (1000, 200, [(4, 1), (2, 2)]),
    (777, 100, [(4, 1)]),                  # last quad holds 9 neurons
    (1300, 300, [(8, 1), (4, 2)]),
    (2000, 400, [(8, 1), (4, 2), (2, 4)]),
    (4000, 800, [(16, 1), (8, 2), (4, 4)]),
])
def test_window_plus_list_reproduce_every_row(n, k, layouts):
    p = R.SimulationParams(num_neurons=n, num_output_neurons=n // 2, small_world_graph_k=k, mean_weight=0.01)
    res = R.build_reservoir(p, 16)
    H, NQ, a4, nbytes, wsq, band, in_list = ring_tables(n, res.csc_ptr, res.csc_post, res.csc_w)
    assert H == k // 2 and 2 * (2 * H + 1) <= n and wsq < NQ
    assert in_list.mean() < 0.12                                     # ~10 % of a small-world graph is rewired
    dense = np.zeros((n, n), dtype=np.float32)
    pre = np.repeat(np.arange(n), np.diff(res.csc_ptr))
    dense[pre, res.csc_post] = res.csc_w
    rows = np.unique(np.concatenate([np.arange(0, n, 37), np.arange(0, 260), np.arange(n - 260, n),
                                     np.arange(H - 3, H + 260), np.arange(n - H - 260, n - H + 3)])) % n
    for wpc, ql in layouts:
        for strided in (False, True):
            if strided:
                if NQ % wpc or wsq > wpc or NQ // wpc not in (1, 2, 4):
                    continue                                          # the builder does not offer it
                ql_, owner = NQ // wpc, (lambda i: (i >> 8) % wpc)
            else:
                assert wpc * ql * 256 >= n and wsq + ql <= NQ         # the constraint the builder enforces
                ql_, owner = ql, (lambda i: (i >> 8) // ql)
            for jj in rows:
                got = fetch_row(jj, wpc, ql_, n, NQ, a4, nbytes, band, strided)
                e0, e1 = res.csc_ptr[jj], res.csc_ptr[jj + 1]
                lst = in_list[e0:e1]
                want = dense[jj].copy()
                want[res.csc_post[e0:e1][lst]] = 0.0                 # those arrive through the list instead
                # real neurons get exactly the window weights; a fetched value is never a synapse of the list
                np.testing.assert_array_equal(got[:n], want, err_msg=f"row {jj} layout {(wpc, ql_, strided)}")
                # list entries belong to the wave that owns the target; a (row, wave) list fits one wavefront
                owners = owner(res.csc_post[e0:e1][lst])
                assert np.all(owners < wpc) and np.bincount(owners, minlength=wpc).max() <= 64


def test_accumulator_layout():
    """ring_acc_word / ring_cnt_word (csrc/lif_ring.h): every neuron has its own float32 accumulator word behind the
    64 dump words, a lane's four neurons form one aligned 16-byte group and a quad is 1 KB (so the 16-byte accesses
    of a wave are 64 consecutive groups: conflict-free); the input counts are 16 bits per neuron, two per word,
    and a lane's four counts are one aligned 8-byte group.  List entries carry acc byte offsets < 64 KB."""
    for npad in (512, 1024, 4096, 8192):
        i = np.arange(npad)
        acc = 64 + i
        assert len(np.unique(acc)) == npad and acc.min() >= 64 and acc.max() * 4 < 65536
        lane_groups = acc.reshape(-1, 4)
        assert np.all(lane_groups[:, 0] % 4 == 0) and np.all(np.diff(lane_groups, axis=1) == 1)
        assert np.all(acc.reshape(-1, 256)[:, 0] * 4 == 256 + np.arange(npad // 256) * 1024)
        cnt_word, cnt_half = 64 + (i >> 1), i & 1
        assert len(np.unique(cnt_word * 2 + cnt_half)) == npad
        per_lane = (cnt_word * 2 + cnt_half).reshape(-1, 4)
        assert np.all(per_lane[:, 0] % 4 == 0) and np.all(np.diff(per_lane, axis=1) == 1)


@pytest.mark.parametrize("n,k,wpc", [(1000, 200, 4), (4000, 800, 8)])
def test_read_modify_write_order_of_a_row(n, k, wpc):
    """The kernel applies a row to LDS accumulators: read the 1 KB quad under the window load and the words under
    the list entries, add, write the quad, THEN the list words (csrc/lif_ring.h, APPLY).  Restated here for the
    strided ownership: a list target often lies in a part of the window quad the window does not cover (the two
    reads return the same old value; the quad write stores old + 0 there and the list write must come second).
    Checks that such targets exist in the reservoirs the GPU parity tests use, that the order 'quad, then list'
    gives W[j -> .] exactly and that the opposite order would not."""
    p = R.SimulationParams(num_neurons=n, num_output_neurons=n // 2, small_world_graph_k=k, mean_weight=0.01)
    res = R.build_reservoir(p, 16)
    H, NQ, a4, nbytes, wsq, band, in_list = ring_tables(n, res.csc_ptr, res.csc_post, res.csc_w)
    assert NQ % wpc == 0 and wsq <= wpc
    ql = NQ // wpc
    npad = NQ * 256
    overlaps = wrong_if_swapped = 0
    for jj in list(range(0, n, 41)) + [0, 1, n - 1, H, n - H - 1]:
        e0, e1 = res.csc_ptr[jj], res.csc_ptr[jj + 1]
        post, wts = res.csc_post[e0:e1], res.csc_w[e0:e1]
        lst = in_list[e0:e1]
        window = fetch_row(jj, wpc, ql, n, NQ, a4, nbytes, band, strided=True)      # zeros where nothing was fetched
        q0 = int(a4[jj]) >> 8
        acc = np.arange(npad, dtype=np.float32) * 0.5 + 1.0                         # some old sums
        want = acc.copy()
        want[post] += wts
        got, swapped = acc.copy(), acc.copy()
        for w in range(wpc):
            gh = (q0 + (w - q0) % wpc) % NQ                                         # the quad wave w reads and writes
            quad = slice(gh * 256, gh * 256 + 256)
            mine = lst & ((post >> 8) % wpc == w)                                   # list entries wave w owns
            old_quad, old_words = acc[quad].copy(), acc[post[mine]].copy()          # both reads precede both writes
            overlaps += int(np.sum((post[mine] >> 8) == gh))
            got[quad] = old_quad + window[quad]
            got[post[mine]] = old_words + wts[mine]
            swapped[post[mine]] = old_words + wts[mine]
            swapped[quad] = old_quad + window[quad]
        np.testing.assert_array_equal(got[:n], want[:n])
        wrong_if_swapped += int(np.any(swapped[:n] != want[:n]))
    assert overlaps > 0 and wrong_if_swapped > 0
