"""CPU check of the pair-block ring tables (csrc/lif_pair.h, csrc/reservoir.hip): every row is applied on the host from the
tables alone -- the 16-byte record of each (row, wave), the window table, the lists -- exactly as a wave of the kernel reads
them (8 bytes per lane at `so + lane*8`, in range iff 0 <= offset < bytes that exist; list entry -> LDS byte offset of the
target's accumulator), and the result must be the row's column of W: every synapse of the row lands on its target exactly once,
in the wave that owns the target's block, nothing lands anywhere else.  `lsm_debug_pair_layout` is host arithmetic only (the
functions lsm_reservoir_create uses), so this runs without a GPU.  Reference: the row is presynaptic neuron j's outgoing weights
(SPEC.md 2.1-2.2; the sum of SPEC.md 3, `acc = acc + w_ij * s_j`, visits exactly these)."""
import ctypes as C

import numpy as np
import pytest

from lsm_speech_classifier_amd import _lib, reservoir as R
from lsm_speech_classifier_amd.csrc_consts import PAIR_DUMP_BYTES

BAND_ADDR, REM_ADDR = 0x7F0012340000, 0x7F00F0000100        # fictitious device addresses (not crossing a 4 GB line)


def _tables(res, wpc):
    lib = _lib.load()
    n = res.num_neurons
    ptr = np.ascontiguousarray(res.csc_ptr, dtype=np.int32)
    post = np.ascontiguousarray(res.csc_post, dtype=np.int32)
    w = np.ascontiguousarray(res.csc_w, dtype=np.float32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    nb, nl, pitch = C.c_long(), C.c_long(), C.c_int()
    bl = lib.lsm_debug_pair_layout(n, p(ptr), p(post), p(w), wpc, BAND_ADDR, REM_ADDR, C.byref(nb), C.byref(nl), C.byref(pitch),
                                   None, None, None)
    assert bl >= 0, lib.lsm_last_error()
    if bl == 0:
        return None
    band = np.zeros(nb.value, dtype=np.float32)
    rem = np.zeros((max(nl.value, 1), 2), dtype=np.uint32)
    rec = np.zeros((n * wpc, 4), dtype=np.uint32)
    assert lib.lsm_debug_pair_layout(n, p(ptr), p(post), p(w), wpc, BAND_ADDR, REM_ADDR, C.byref(nb), C.byref(nl),
                                     C.byref(pitch), p(band), p(rem), p(rec)) == bl
    return bl, pitch.value, band, rem[:nl.value], rec.reshape(n, wpc, 4)


def _expected_wave_counts(res):
    """Which wave counts have a pair layout, restated from the rules (csrc/reservoir.hip: pair_lists): 2*ceil(N/256) blocks a
    multiple of the waves, at most four blocks per wave, no window wider than the waves' blocks (the stored window starts at
    the 32-aligned first target of j-H and runs to j+H along the padded ring)."""
    n = res.num_neurons
    nnz = int(res.csc_ptr[-1])
    h = (nnz // n + 1) // 2
    nq = (n + 255) // 256
    j = np.arange(n)
    a4 = ((j - h) % n) & ~31
    span = (((j + h) % n) - a4) % (nq * 256)
    last = (span >> 2) * 4 + 3                                   # last float of the last 16-byte granule
    wsb = int((((a4 & 127) + last) >> 7).max()) + 1
    return tuple(w for w in (4, 8, 16) if (2 * nq) % w == 0 and 1 <= 2 * nq // w <= 4 and wsb <= w)


@pytest.mark.parametrize("n,k", [(1024, 120), (1000, 200), (1536, 300), (2048, 300), (2048, 409), (3072, 614), (4000, 800),
                                 (3900, 400), (6144, 500)])
def test_every_row_applied_from_the_pair_tables_is_its_column_of_w(n, k):
    p = R.SimulationParams(num_neurons=n, num_output_neurons=max(1, n // 3), small_world_graph_k=k, mean_weight=0.004)
    res = R.build_reservoir(p, 16)
    nq = (n + 255) // 256
    npad = nq * 256
    found = []
    rs = np.random.RandomState(n)
    for wpc in (4, 8, 16):
        t = _tables(res, wpc)
        if t is None:
            continue
        found.append(wpc)
        bl, pitch, band, rem, rec = t
        assert bl * wpc == 2 * nq and 1 <= bl <= 4 and pitch % 128 == 0
        lane8 = np.arange(64) * 8
        rows = np.unique(np.concatenate([[0, 1, n // 2, n - 1], rs.randint(0, n, size=60)]))
        for j in rows:
            acc = np.zeros(npad, dtype=np.float64)
            hits = np.zeros(npad, dtype=np.int32)
            for w in range(wpc):
                row_lo, y, z, list_lo = (int(v) for v in rec[j, w])
                so = ((y & 0xFFFF) ^ 0x8000) - 0x8000                           # signed 16 bits
                lds_off = y >> 16
                nbytes, list_bytes = z & 0xFFFF, z >> 16
                assert row_lo == (BAND_ADDR + j * pitch) & 0xFFFFFFFF and lds_off % 512 == 0
                gb = lds_off // 512
                assert gb % wpc == w and gb < 2 * nq                             # the wave's own block
                off = lane8 + so                                                 # what the kernel puts into the load's offset
                inr = (off >= 0) & (off < nbytes)
                for lane in np.nonzero(inr)[0]:
                    q = (j * pitch + off[lane]) // 4
                    for h in range(2):
                        tgt = gb * 128 + lane * 2 + h
                        acc[tgt] += band[q + h]
                        hits[tgt] += band[q + h] != 0
                assert list_bytes % 8 == 0 and list_bytes <= 512
                first = (list_lo - (REM_ADDR & 0xFFFFFFFF)) & 0xFFFFFFFF
                assert first % 8 == 0
                for e in range(first // 8, first // 8 + list_bytes // 8):
                    tgt = (int(rem[e, 0]) - PAIR_DUMP_BYTES) // 4
                    assert (int(rem[e, 0]) - PAIR_DUMP_BYTES) % 4 == 0 and 0 <= tgt < n and (tgt >> 7) % wpc == w
                    acc[tgt] += rem[e:e + 1, 1].view(np.float32)[0]
                    hits[tgt] += 1
            col = np.zeros(npad, dtype=np.float64)
            a, b = res.csc_ptr[j], res.csc_ptr[j + 1]
            col[res.csc_post[a:b]] = res.csc_w[a:b]
            np.testing.assert_array_equal(acc.astype(np.float32), col.astype(np.float32), err_msg=f"row {j} wpc {wpc}")
            assert hits.max() <= 1 and hits.sum() == np.count_nonzero(res.csc_w[a:b])    # each synapse exactly once
            assert not acc[n:].any()
    assert tuple(found) == _expected_wave_counts(res), (found, _expected_wave_counts(res))
    if (n, k) == (4000, 800):
        assert found == [8, 16]                                  # BASELINE configs[3]: 8 waves x 4 blocks is what runs
    if n == 3900:
        assert found == [8, 16]                                  # the ring is padded to 16 quads = 32 blocks: 196 padding neurons
