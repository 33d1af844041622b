"""Error behaviour of the C ABI on the GPU box: bad arguments return a negative status with a
message (surfaced as LsmHipError), nothing is launched, and the library stays usable afterwards."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bad_arguments_raise_and_library_survives():
    import torch
    from lsm_speech_classifier_amd import _lib, frontend, reservoir as R, snn, synth
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    audio = torch.zeros((2, 16000), dtype=torch.float32, device="cuda")
    fe = frontend.SpikeFrontEnd(8, "gammatone")
    db = torch.empty((2, 8, 98), dtype=torch.float64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())

    # window longer than 4 hops, columns beyond the clip, both outputs null
    for args in ((p(audio), 2, 16000, p(fe.coefs), 8, 400, 50, 98, None, p(db), 3, stream),
                 (p(audio), 2, 16000, p(fe.coefs), 8, 400, 160, 200, None, p(db), 3, stream),
                 (p(audio), 2, 16000, p(fe.coefs), 8, 400, 160, 98, None, None, 3, stream)):
        rc = lib.lsm_gammatone_spec_f64(*args)
        assert rc < 0 and lib.lsm_last_error()
        with pytest.raises(_lib.LsmHipError):
            _lib.check(rc, "lsm_gammatone_spec_f64")
    # more than 8 thresholds
    thr = np.linspace(0.1, 0.9, 9)
    with pytest.raises(_lib.LsmHipError, match="n_thr"):
        frontend.convert_spectrogram_to_spikes_hysteresis(np.zeros((2, 10)), list(thr), 0.05)
    # wrong number of samples / unknown filterbank are host-side errors
    with pytest.raises(ValueError):
        fe.encode(np.zeros((1, 100), dtype=np.float32))
    with pytest.raises(ValueError):
        frontend.SpikeFrontEnd(8, "bark")

    rasters = synth.bernoulli_raster(2, 8, 40, 0.3, seed=1)
    prm = R.SimulationParams(num_neurons=100, num_output_neurons=40, small_world_graph_k=10, mean_weight=0.05)
    net = snn.SNN(prm, n_channels=8)
    with pytest.raises(ValueError):                      # channel count mismatch
        net.run_batch(np.zeros((1, 9, 40), dtype=np.uint8))
    with pytest.raises(_lib.LsmHipError, match="layout"):
        net.run_batch(rasters, waves_per_clip=3)         # not a supported layout
    key_ids = np.array([9], dtype=np.int32)
    feats = torch.empty((2, 40), dtype=torch.float32, device="cuda")
    r = torch.from_numpy(rasters).cuda()
    rc = lib.lsm_reservoir_run(net._handle, p(r), 2, 40, C.c_void_p(key_ids.ctypes.data), 1, p(feats),
                               None, None, None, 0, stream)
    assert rc < 0 and b"key id" in lib.lsm_last_error()
    rc = lib.lsm_reservoir_run(None, p(r), 2, 40, C.c_void_p(key_ids.ctypes.data), 1, p(feats), None, None, None, 0, stream)
    assert rc < 0
    # invalid wiring is rejected at create time
    bad = R.build_reservoir(prm, 8)
    bad.out_idx = bad.out_idx[::-1].copy()               # not ascending
    with pytest.raises(_lib.LsmHipError, match="out_idx"):
        snn.SNN(None, reservoir=bad)
    with pytest.raises(_lib.LsmHipError, match="membrane_threshold"):
        snn.SNN(R.SimulationParams(num_neurons=100, small_world_graph_k=10, membrane_threshold=0.0), n_channels=8)
    # bit packing: bad shape, null buffer, misaligned raster for rows that move as 64-bit words
    ras = torch.zeros((4, 16), dtype=torch.uint8, device="cuda")
    pk = torch.zeros((4, 2), dtype=torch.uint8, device="cuda")
    for fn, a, b in ((lib.lsm_raster_pack_bits, ras, pk), (lib.lsm_raster_unpack_bits, pk, ras)):
        assert fn(p(a), 4, 0, p(b), stream) < 0 and b"bad shape" in lib.lsm_last_error()
        assert fn(p(a), -1, 16, p(b), stream) < 0
        assert fn(None, 4, 16, p(b), stream) < 0 and b"null" in lib.lsm_last_error()
        assert fn(p(a), 0, 16, None, stream) == 0            # nothing to do: no buffers needed
    odd = torch.zeros(4 * 16 + 1, dtype=torch.uint8, device="cuda")[1:].view(4, 16)
    assert lib.lsm_raster_pack_bits(p(odd), 4, 16, p(pk), stream) < 0 and b"aligned" in lib.lsm_last_error()
    odd15 = torch.zeros(4 * 15 + 1, dtype=torch.uint8, device="cuda")[1:].view(4, 15)
    assert lib.lsm_raster_pack_bits(p(odd15), 4, 15, p(pk), stream) == 0      # byte path: any alignment
    with pytest.raises(ValueError):                       # packed width does not match the step count
        net.run_batch(np.zeros((1, 8, 5), dtype=np.uint8), packed_time_steps=400)
    # ... and the library still works
    f, _, _ = net.run_batch(rasters)
    assert f.shape == (2, 8 * 40) and torch.isfinite(f).all()
