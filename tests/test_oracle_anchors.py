"""Independent anchors for the parts of the oracle that NOTHING in the reference pins (gammatone==1.0.3,
librosa==0.11.0 and snn_reservoir_py==2.0.0 are absent: SURVEY.md §8c).  They cannot be pinned; these tests
make sure they do not merely agree with themselves:

* the gammatone coefficient table is checked against its DEFINITION with scipy.signal.freqz (unit gain at
  the centre frequency, ERB spacing, ERB-proportional bandwidth), not against the second copy of the table,
  and against scipy.signal.gammatone -- SciPy's own implementation of the same published filter -- coefficient by
  coefficient;
* the STFT and the mel filterbank against scipy.signal / torch.stft and closed-form triangle areas, and the whole
  mel front end against transformers.audio_utils (an independent implementation modelled on librosa);
* the hand-rolled Watts-Strogatz wiring against networkx.watts_strogatz_graph statistics over 20 seeds;
* (GPU, tests/test_gpu_parity.py::test_oracle_on_a_reservoir_it_did_not_build) the LIF oracle and the
  kernels on a random CSR that reservoir.build_reservoir never produced.
"""
import numpy as np
import pytest
from scipy import signal

from oracle import ref_numpy as O

FS = 16000


# --------------------------------------------------------------------------- gammatone ----
@pytest.mark.parametrize("channels", [8, 40, 128, 256])
def test_gammatone_table_has_unit_gain_at_its_centre_frequencies(channels):
    cfs = O.erb_centre_freqs(FS, channels, 50)[::-1]                 # ascending, like the table
    tab = O.gammatone_coefs(FS, channels, 50)
    from lsm_speech_classifier_amd import frontend
    prod = frontend.gammatone_filter_table(FS, channels, 50)
    for name, t in (("oracle", tab), ("product", prod)):
        assert t.shape == (channels, 10)
        for ch in range(0, channels, max(1, channels // 32)):
            A0, A11, A12, A13, A14, A2, B0, B1, B2, gain = t[ch]
            w = 2 * np.pi * cfs[ch] / FS
            h = 1.0 + 0j
            for a1 in (A11, A12, A13, A14):
                _, hk = signal.freqz([A0, a1, A2], [B0, B1, B2], worN=[w])
                h *= hk[0]
            # the cascade divided by `gain` passes its own centre frequency with |H| = 1
            assert abs(abs(h) / gain - 1.0) < 1e-9, (name, ch)


def test_gammatone_centre_frequencies_are_erb_spaced():
    for channels in (40, 128):
        cfs = O.erb_centre_freqs(FS, channels, 50)
        assert np.all(np.diff(cfs) < 0)                              # descending from just under fs/2
        assert abs(cfs[-1] - 50.0) < 1e-9 and cfs[0] < FS / 2
        # equal steps on the ERB-rate scale  E(f) = EarQ * ln(1 + f / (EarQ * minBW))  (Glasberg & Moore)
        c = O.EAR_Q * O.MIN_BW
        e = O.EAR_Q * np.log(1 + cfs / c)
        steps = np.diff(e)
        np.testing.assert_allclose(steps, steps.mean(), rtol=1e-10)


def test_gammatone_bandwidth_follows_the_erb():
    """-3 dB bandwidth of a 4th-order gammatone = 2*sqrt(2^(1/4)-1) * b with b = 1.019 * 2*pi*ERB(cf) (in rad/s):
    about 0.887 * 1.019 * ERB in Hz.  Read off the filters' frequency responses (not off the table's formula)."""
    channels = 32
    cfs = O.erb_centre_freqs(FS, channels, 50)[::-1]
    tab = O.gammatone_coefs(FS, channels, 50)
    for ch in (4, 12, 20, 27):
        A0, A11, A12, A13, A14, A2, B0, B1, B2, gain = tab[ch]
        f = np.linspace(max(1.0, cfs[ch] * 0.5), min(FS / 2 - 1, cfs[ch] * 1.6), 20001)
        h = np.ones_like(f, dtype=complex)
        for a1 in (A11, A12, A13, A14):
            h *= signal.freqz([A0, a1, A2], [B0, B1, B2], worN=2 * np.pi * f / FS)[1]
        mag = np.abs(h) / gain
        above = f[mag >= 1 / np.sqrt(2)]
        bw = above[-1] - above[0]
        erb = cfs[ch] / O.EAR_Q + O.MIN_BW
        want = 2 * np.sqrt(2 ** 0.25 - 1) * 1.019 * erb
        assert abs(bw / want - 1) < 0.03, (ch, bw, want)


@pytest.mark.parametrize("channels", [40, 128, 256])
def test_gammatone_table_equals_scipy_signal_gammatone(channels):
    """An independent implementation of the same published filter: scipy.signal.gammatone(cf, 'iir', fs) (SciPy's own
    code after Slaney 1993: 4th order, bandwidth 1.019 ERB, unit gain at cf) returns the 8th-order transfer function
    as two polynomials.  The four sections of every table row, multiplied out and divided by `gain`, must be those
    polynomials -- filter design AND gain normalisation, for the oracle's table and the product's."""
    from lsm_speech_classifier_amd import frontend
    cfs = O.erb_centre_freqs(FS, channels, 50)[::-1]
    for name, tab in (("oracle", O.gammatone_coefs(FS, channels, 50)),
                      ("product", frontend.gammatone_filter_table(FS, channels, 50))):
        for ch in range(channels):
            A0, A11, A12, A13, A14, A2, B0, B1, B2, gain = tab[ch]
            assert A2 == 0.0 and B0 == 1.0
            num, den = np.array([1.0]), np.array([1.0])
            for a1 in (A11, A12, A13, A14):
                num = np.convolve(num, [A0, a1])
                den = np.convolve(den, [B0, B1, B2])
            b, a = signal.gammatone(cfs[ch], "iir", fs=FS)
            assert len(b) == 5 and len(a) == 9
            np.testing.assert_allclose(num / gain, b, rtol=0, atol=2e-11 * np.abs(b).max(), err_msg=f"{name} ch {ch}")
            np.testing.assert_allclose(den, a, rtol=0, atol=1e-13 * np.abs(a).max(), err_msg=f"{name} ch {ch}")


def test_gammatone_filterbank_output_against_scipy_signal_gammatone():
    """Time domain: erb_filterbank (the oracle's four-section cascade, / gain) against scipy.signal.lfilter over
    SciPy's own 8th-order polynomials of the same filter.  The direct 8th-order form loses precision as the poles
    approach z = 1 (1e-2 at 50 Hz, 5e-5 at 154 Hz -- the reason the cascade is the form everybody evaluates), so the
    bound tightens with the centre frequency."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal(4000)
    cfs = O.erb_centre_freqs(FS, 32, 50)[::-1]
    y = O.erb_filterbank(x, O.gammatone_coefs(FS, 32, 50))
    checked = 0
    for ch in range(32):
        if cfs[ch] < 450:
            continue
        b, a = signal.gammatone(cfs[ch], "iir", fs=FS)
        ref = signal.lfilter(b, a, x)
        tol = 2e-6 if cfs[ch] < 1000 else 1e-8
        assert np.max(np.abs(y[ch] - ref)) <= tol * np.max(np.abs(ref)), (ch, cfs[ch])
        checked += 1
    assert checked >= 20


def test_gammatone_filterbank_equals_an_independent_sos_cascade():
    """erb_filterbank (four scipy.signal.lfilter calls) against one scipy.signal.sosfilt cascade built from the
    same coefficients: same filter, different evaluation (direct form II transposed per section either way)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(4000)
    tab = O.gammatone_coefs(FS, 16, 50)
    y = O.erb_filterbank(x, tab)
    for ch in (0, 7, 15):
        A0, A11, A12, A13, A14, A2, B0, B1, B2, gain = tab[ch]
        sos = np.array([[A0, a1, A2, B0, B1, B2] for a1 in (A11, A12, A13, A14)])
        ref = signal.sosfilt(sos, x) / gain
        np.testing.assert_allclose(y[ch], ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


# --------------------------------------------------------------------------------- mel ----
def test_stft_power_against_scipy_and_torch():
    import torch
    rng = np.random.default_rng(5)
    y = (rng.standard_normal(FS) * 0.1).astype(np.float32)
    S = O.stft_power(y, 2048, 160)                                   # (1025, 101) float32
    assert S.shape == (1025, 101) and S.dtype == np.float32
    # scipy: same window, zero padding at both ends, no scaling
    _, _, Z = signal.stft(np.concatenate([np.zeros(1024), y.astype(np.float64), np.zeros(1024)]), fs=FS,
                          window=signal.get_window("hann", 2048, fftbins=True), nperseg=2048, noverlap=2048 - 160,
                          nfft=2048, boundary=None, padded=False, return_onesided=True, scaling="spectrum")
    Z = Z * signal.get_window("hann", 2048, fftbins=True).sum()      # undo scipy's 'spectrum' scaling
    ref = np.abs(Z) ** 2
    assert ref.shape == S.shape
    np.testing.assert_allclose(S, ref, rtol=2e-4, atol=1e-6 * ref.max())
    # torch: center=True with constant (zero) padding
    T = torch.stft(torch.from_numpy(y).double(), 2048, hop_length=160, win_length=2048,
                   window=torch.hann_window(2048, periodic=True, dtype=torch.float64), center=True,
                   pad_mode="constant", return_complex=True)
    ref_t = (T.abs() ** 2).numpy()
    np.testing.assert_allclose(S, ref_t, rtol=2e-4, atol=1e-6 * ref_t.max())


def test_mel_front_end_against_transformers_audio_utils():
    """An independent implementation modelled on librosa: Hugging Face's transformers.audio_utils (mel_filter_bank with
    Slaney scale and normalisation, spectrogram with a periodic hann window, centred frames, zero padding, power 2,
    power_to_db with ref = max, amin 1e-10, 80 dB range).  The oracle's mel filterbank, mel power spectrogram and dB
    spectrogram (create_dataset.py:43-48 through librosa 0.11's defaults) must be its numbers."""
    au = pytest.importorskip("transformers.audio_utils")
    for n_mels in (13, 40, 128):
        fb = au.mel_filter_bank(num_frequency_bins=O.MEL_N_FFT // 2 + 1, num_mel_filters=n_mels, min_frequency=0.0,
                                max_frequency=FS / 2, sampling_rate=FS, norm="slaney", mel_scale="slaney")
        mine = O.mel_filterbank(FS, O.MEL_N_FFT, n_mels)
        assert np.max(np.abs(fb.T - mine)) <= 2e-7 * np.max(np.abs(mine)), n_mels
    rng = np.random.default_rng(0)
    t = np.arange(FS) / FS
    for k, x in enumerate([(rng.standard_normal(FS) * 0.1).astype(np.float32),
                           (0.5 * np.sin(2 * np.pi * (200 + 900 * t) * t) * np.hanning(FS)).astype(np.float32)]):
        win = au.window_function(O.MEL_N_FFT, "hann", periodic=True)
        fb = au.mel_filter_bank(O.MEL_N_FFT // 2 + 1, 40, 0.0, FS / 2, FS, norm="slaney", mel_scale="slaney")
        ref = au.spectrogram(x, win, frame_length=O.MEL_N_FFT, hop_length=160, fft_length=O.MEL_N_FFT, power=2.0,
                             center=True, pad_mode="constant", mel_filters=fb)
        mine = O.mel_power(x, 40, FS, 160)
        assert ref.shape == mine.shape == (40, 101)
        assert np.max(np.abs(ref - mine)) <= 2e-6 * np.max(np.abs(mine)), k
        ref_db = au.power_to_db(ref, reference=float(ref.max()), min_value=1e-10, db_range=80.0)
        assert np.max(np.abs(ref_db - O.power_to_db(mine))) <= 5e-5, k


@pytest.mark.parametrize("n_mels", [40, 128])
def test_mel_filterbank_triangles_in_closed_form(n_mels):
    B = O.mel_filterbank(FS, 2048, n_mels).astype(np.float64)
    assert B.shape == (n_mels, 1025) and np.all(B >= 0)
    edges = O._mel_to_hz(np.linspace(O._hz_to_mel(0.0), O._hz_to_mel(FS / 2), n_mels + 2))
    # Slaney scale: linear below 1 kHz (200/3 Hz per mel), logarithmic above (log(6.4)/27 per mel)
    assert abs(O._hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(O._mel_to_hz(15.0) - 1000.0) < 1e-9
    assert abs(O._mel_to_hz(15.0 + 27.0) - 6400.0) < 1e-6
    freqs = np.fft.rfftfreq(2048, 1 / FS)
    df = freqs[1]
    for m in range(n_mels):
        lo, c, hi = edges[m], edges[m + 1], edges[m + 2]
        nz = np.nonzero(B[m])[0]
        assert freqs[nz[0]] > lo - df and freqs[nz[-1]] < hi + df     # support = (lo, hi)
        peak = freqs[np.argmax(B[m])]
        assert abs(peak - c) <= df                                    # apex at the centre edge
        # Slaney normalisation: a triangle of height 2/(hi-lo) has unit area; the sampled one too, up to
        # the bin width
        area = B[m].sum() * df
        assert abs(area - 1.0) < 2.5 * df / (hi - lo) + 1e-6, (m, area)
        assert B[m].max() <= 2.0 / (hi - lo) * (1 + 1e-6)             # float32 storage of the weights


# ------------------------------------------------------------------------------- wiring ----
def test_small_world_wiring_against_networkx_statistics():
    import networkx as nx
    from lsm_speech_classifier_amd import reservoir as R
    n, k, p = 400, 40, 0.1
    ours, theirs = [], []
    for seed in range(20):
        adj = R.small_world_edges(n, k, p, np.random.RandomState(seed))
        assert adj.dtype == bool and np.array_equal(adj, adj.T) and not adj.diagonal().any()
        g = nx.watts_strogatz_graph(n, k, p, seed=seed)
        a_nx = nx.to_numpy_array(g, dtype=bool)
        for name, a, acc in (("ours", adj, ours), ("networkx", a_nx, theirs)):
            deg = a.sum(axis=1)
            assert a.sum() == n * k                                  # n*k/2 undirected edges, none lost or doubled
            i, j = np.nonzero(np.triu(a))
            ring = np.minimum((j - i) % n, (i - j) % n) <= k // 2
            acc.append((deg.mean(), deg.var(), ring.mean(),
                        nx.average_clustering(nx.from_numpy_array(a)) if seed < 3 else np.nan))
    ours, theirs = np.array(ours), np.array(theirs)
    assert np.all(ours[:, 0] == k) and np.all(theirs[:, 0] == k)     # mean degree exactly k
    # a rewired edge must go to a node that is not yet adjacent, and the ring neighbours are: it lands outside
    # the ring window, so the share of edges inside the window is 1 - p (plus the few slots rewiring freed);
    # the two generators must agree with that and with each other
    want = 1 - p
    assert abs(ours[:, 2].mean() - want) < 0.004 and abs(theirs[:, 2].mean() - want) < 0.004
    assert abs(ours[:, 2].mean() - theirs[:, 2].mean()) < 0.004
    # degree variance of Watts-Strogatz rewiring (one end of each rewired edge moves): same law in both
    assert abs(ours[:, 1].mean() / theirs[:, 1].mean() - 1) < 0.1
    # clustering coefficient (three seeds): the small-world signature, far above a random graph's k/n
    assert abs(np.nanmean(ours[:, 3]) - np.nanmean(theirs[:, 3])) < 0.01 and np.nanmean(ours[:, 3]) > 3 * k / n


def test_reverse_cuthill_mckee_cannot_narrow_the_band():
    """VERDICT r1 asked whether a neuron permutation could put >= 98 % of the synapses into the ring window.
    It cannot: ~10 % of a Watts-Strogatz graph's edges have a uniformly random far end, and RCM (scipy) leaves
    the share of synapses within +-k/2 of the diagonal BELOW the identity ordering's 90 %."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    from lsm_speech_classifier_amd import reservoir as R
    n, k = 1000, 200
    adj = R.small_world_edges(n, k, 0.1, np.random.RandomState(42))
    i, j = np.nonzero(adj)

    def inside(pi, pj):
        d = np.abs(pi - pj)
        return (np.minimum(d, n - d) <= k // 2).mean()

    ident = inside(i, j)
    perm = reverse_cuthill_mckee(csr_matrix(adj), symmetric_mode=True)
    pos = np.empty(n, dtype=np.int64)
    pos[perm] = np.arange(n)
    rcm = inside(pos[i], pos[j])
    assert 0.89 < ident < 0.93
    assert rcm < ident                      # the lattice order is already the best band; 98 % is out of reach
