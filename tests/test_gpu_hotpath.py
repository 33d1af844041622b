"""pipeline.HotPath (the overlapped audio -> features path the bench times and the scripts use): identical
rows whatever the stream rotation, host or device inputs, in order; the in-memory route of the drop-in scripts
writes the same File 2 as the npz route; bench.py starts its own ranks on a shared GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'mean_isi', 'isi_variances']


def test_hotpath_rows_equal_the_serial_path_and_the_oracle(oracle_c):
    import torch
    from lsm_speech_classifier_amd import frontend, pipeline, reservoir as R, snn, synth
    from oracle import ref_numpy as O
    assert int(os.environ["GPU_MAX_HW_QUEUES"]) >= 8               # set by the package before HIP initialised
    audio = synth.class_chirps(np.arange(7 * 40) % 12, seed=77)
    fe = frontend.SpikeFrontEnd(64, "gammatone")
    rasters = fe.encode(audio)
    p = R.SimulationParams(num_neurons=1000, num_output_neurons=400, small_world_graph_k=200,
                           mean_weight=O.w_critico(200, 2.0, 2, rasters.cpu().numpy()) * 0.6)
    net = snn.SNN(None, reservoir=R.build_reservoir(p, 64))
    serial, _, _ = net.run_batch(rasters, KEYS)
    batches = [audio[lo:lo + 40] for lo in range(0, len(audio), 40)]
    # serial, the rotation (a step keeps to one stream), and front ends on streams of their own (the default)
    for streams, fes in ((1, None), (3, 0), (6, 0), (3, 2), (6, None)):
        hp = pipeline.HotPath(fe, net, KEYS, streams=streams, fe_streams=fes)
        assert hp.waves_per_clip == (-1 if streams > 1 else 0)
        assert hp.n_fe_streams == (0 if streams == 1 else pipeline.DEFAULT_FE_STREAMS if fes is None else fes)
        for src in (batches, [torch.from_numpy(b).cuda() for b in batches]):       # host and device inputs
            got = hp.run(src)
            assert torch.equal(got, serial), (streams, fes)
    # the layout the library picks inside a pipeline: fewer, fatter waves than for a lone launch of 256 clips
    assert net.layout(256, 400, -1)["waves_per_clip"] == 4 and net.layout(256, 400, 0)["waves_per_clip"] == 8
    ref = oracle_c.lif_run_batch(net.reservoir, rasters[:4].cpu().numpy(), KEYS, n_threads=4)
    np.testing.assert_array_equal(serial[:4].cpu().numpy(), ref)
    host = pipeline.features_from_audio(audio, fe, net, KEYS, batch=64)
    np.testing.assert_array_equal(host, serial.cpu().numpy())


def test_run_gives_the_last_batches_of_a_finite_list_the_low_latency_layouts():
    """Round 4: HotPath.run() submits the last TAIL_STEPS batches with tail=True -- front end in the one-chain layout, the
    reservoir launch laid out as a lone launch -- so that the chip drains evenly at the end of a burst (bench.py does the
    same for its last timed steps).  Layout hints only: the rows are those of the serial path."""
    import torch
    from lsm_speech_classifier_amd import frontend, pipeline, reservoir as R, snn, synth
    audio = synth.class_chirps(np.arange(6 * 32) % 12, seed=5)
    fe = frontend.SpikeFrontEnd(128, "gammatone")
    net = snn.SNN(R.SimulationParams(num_neurons=1000, num_output_neurons=400, small_world_graph_k=200, mean_weight=0.006),
                  n_channels=128)
    serial, _, _ = net.run_batch(fe.encode(audio), KEYS)
    batches = [torch.from_numpy(audio[lo:lo + 32]).cuda() for lo in range(0, len(audio), 32)]
    hp = pipeline.HotPath(fe, net, KEYS)
    seen_fe, seen_wpc = [], []
    real_encode, real_run = fe.encode, net.run_batch
    fe.encode = lambda x, **kw: (seen_fe.append(bool(kw.get("low_latency"))), real_encode(x, **kw))[1]
    net.run_batch = lambda r, keys=None, **kw: (seen_wpc.append(kw.get("waves_per_clip")), real_run(r, keys, **kw))[1]
    try:
        got = hp.run(batches)
    finally:
        fe.encode, net.run_batch = real_encode, real_run
    assert torch.equal(got, serial)
    n = pipeline.TAIL_STEPS
    assert n >= 1 and seen_fe[-n:] == [True] * n and seen_wpc[-n:] == [0] * n      # the tail
    assert seen_fe[0] is True and seen_wpc[0] == -1                                # first step: idle GPU, pipeline layout
    assert seen_fe[1:-n].count(True) <= 1 and set(seen_wpc[:-n]) == {-1}


def test_in_memory_route_writes_the_same_file_2(tmp_path, monkeypatch):
    import create_dataset as cd
    import extract_lsm_features as ex
    monkeypatch.chdir(tmp_path)
    words = ["yes", "no", "up"]
    cd.create_dataset(64, "gammatone", commands=words, synthetic_per_class=16)
    ex.main("original", 0.6)
    with np.load(ex.FEATURE_FILE, allow_pickle=True) as d:
        want = {k: d[k] for k in d.files}
    os.remove(ex.FEATURE_FILE)
    os.remove(cd.OUTPUT_FILE)
    audio, labels = cd.collect_audio(commands=words, synthetic_per_class=16)
    ex.main_from_audio(audio, labels, 64, "gammatone", "original", 0.6)
    assert not os.path.exists(cd.OUTPUT_FILE)                      # no File 1 round trip
    with np.load(ex.FEATURE_FILE, allow_pickle=True) as d:
        assert sorted(d.files) == sorted(want)
        for k in ("X_train_features", "X_test_features", "y_train", "y_test"):
            np.testing.assert_array_equal(d[k], want[k], err_msg=k)
        assert str(d["feature_set"]) == "original"


def test_in_memory_route_keeps_the_features_on_the_device_for_a_torch_readout(tmp_path, monkeypatch, capsys):
    """SURVEY.md 8f-2 / VERDICT r3 #6: with a PyTorch readout the gathered rows go device -> readout.StandardScaler ->
    ridge -> predictions without a host copy (the reference's round trip: extract_lsm_features.py:199-212 ->
    train_classifier.py:27-45); File 2 is still written.  cfg4-shaped corpus (35 classes, N = 4000, ring-row kernel):
    the predictions equal scikit-learn's RidgeClassifier on the host route's File 2."""
    import torch
    import create_dataset as cd
    import extract_lsm_features as ex
    import train_classifier as tc
    from sklearn.linear_model import RidgeClassifier
    from lsm_speech_classifier_amd import readout as ro
    monkeypatch.chdir(tmp_path)
    words = [f"word{i:02d}" for i in range(35)]
    audio, labels = cd.collect_audio(commands=words, synthetic_per_class=5)

    seen = {}
    real_fit, real_report = ro.StandardScaler.fit, tc.report_results

    def spy_fit(self, X):
        assert torch.is_tensor(X) and X.is_cuda and X.dtype == torch.float32     # no .cpu() before the scaler
        seen["scaler_rows"] = tuple(X.shape)
        return real_fit(self, X)

    def spy_report(y_train, y_test, y_pred, class_names=None):
        seen["y_pred"], seen["y_test"] = np.asarray(y_pred), np.asarray(y_test)
        return real_report(y_train, y_test, y_pred, class_names)

    monkeypatch.setattr(ro.StandardScaler, "fit", spy_fit)
    monkeypatch.setattr(tc, "report_results", spy_report)
    acc = ex.main_from_audio(audio, labels, 128, "gammatone", "original", 0.6, num_neurons=4000,
                             readout="torch-ridge", class_names=words)
    out = capsys.readouterr().out
    assert "Test Accuracy" in out and "word34" in out and "ridge classifier on the device" in out
    assert seen["scaler_rows"] == (140, 5 * 1600) and 0.0 <= acc <= 1.0
    with np.load(ex.FEATURE_FILE, allow_pickle=True) as d:
        dev_file = {k: d[k] for k in d.files}
    os.remove(ex.FEATURE_FILE)

    # the host route on the same clips: sklearn's scaler (the reference's), sklearn's ridge
    monkeypatch.setattr(ro.StandardScaler, "fit", real_fit)
    assert ex.main_from_audio(audio, labels, 128, "gammatone", "original", 0.6, num_neurons=4000) is None
    with np.load(ex.FEATURE_FILE, allow_pickle=True) as d:
        host_file = {k: d[k] for k in d.files}
    assert sorted(dev_file) == sorted(host_file)
    for k in ("y_train", "y_test"):
        np.testing.assert_array_equal(dev_file[k], host_file[k])
    for k in ("X_train_features", "X_test_features"):
        assert dev_file[k].dtype == host_file[k].dtype == np.float32
        np.testing.assert_allclose(dev_file[k], host_file[k], rtol=2e-5, atol=2e-6)
    clf = RidgeClassifier(alpha=1.0).fit(host_file["X_train_features"], host_file["y_train"])
    np.testing.assert_array_equal(seen["y_test"], host_file["y_test"])
    np.testing.assert_array_equal(seen["y_pred"], clf.predict(host_file["X_test_features"]))


def _bench_two_ranks(exchange, *extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(LSM_BENCH_SHARE_GPU="1", LSM_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--batch", "64", "--no-cpu-baseline", "--exchange", exchange, *extra], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["clips_per_gpu"] == 64 and "all-gather" in d["config"]["sharding"]
    assert abs(d["value"] - 2 * 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    return d


def test_bench_spawns_two_ranks_on_a_shared_gpu():
    """`bench.py --gpus 2` starts its own ranks.  The exchange modes: rows gathered in chunks on a stream of their own
    while the next steps run (the default; a chunk size that does not divide the steps leaves a tail chunk), ONE
    all-gather after the last step (what the product does per split), an all-gather behind every step.  Chunked and
    once deliver the SAME rows: the digest over the gathered block in (rank, step, clip) order is equal, and every
    rank found its own rows back unchanged (asserted inside bench.py)."""
    chunked = _bench_two_ranks("chunked", "--exchange-chunk", "4")          # chunks [0, 4) and the tail [4, 6)
    once = _bench_two_ranks("once")
    for d, mode, chunk in ((chunked, "chunked", 4), (once, "once", 6)):
        x = d["exchange"]
        assert x["mode"] == mode and x["chunk_steps"] == chunk
        assert x["exchange_bytes"] == 2 * 6 * 64 * 2000 * 4 and x["bytes_sent_per_rank"] == 6 * 64 * 2000 * 4
        assert x["exchange_ms"] is not None and 0.0 <= x["exchange_ms"] < d["ms_per_step"] * 6
    assert chunked["exchange"]["digest"] == once["exchange"]["digest"] != 0
    assert "every 4 steps" in chunked["config"]["sharding"] and "after the last step" in once["config"]["sharding"]
    assert _bench_two_ranks("chunked", "--exchange-chunk", "3")["exchange"]["digest"] == once["exchange"]["digest"]
    per_step = _bench_two_ranks("per-step")
    assert "behind every step" in per_step["config"]["sharding"] and "exchange" not in per_step


def test_bench_rccl_code_path_with_one_rank():
    """The calls the 8-GPU run makes -- process group over RCCL (backend "nccl" on ROCm), broadcast of w_critico,
    all_gather_into_tensor of the feature rows on its own stream, barrier + max-over-ranks timing -- executed for
    real on this box's one GPU (LSM_BENCH_FORCE_DIST=1: world size 1)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               LSM_BENCH_FORCE_DIST="1")
    env.pop("LSM_BENCH_BACKEND", None)
    env.pop("GPU_MAX_HW_QUEUES", None)        # this process's package import set 12; a fresh process under a launcher chooses
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--exchange", "chunked"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 1e4 and d["steps"] == 6
    # the chunked exchange on RCCL (opt-in since round 5; the default is the product's single exchange, whose one-rank
    # run is tests/test_gpu_bench_contract.py::test_a_rank_started_by_the_launcher_inherits_what_rccl_needs): chunks of
    # 5 steps on the exchange stream + the tail chunk, 16 hardware queues
    assert d["exchange"]["mode"] == "chunked" and d["exchange"]["chunk_steps"] == 5 and d["config"]["hw_queues"] == 16
    assert 0.0 <= d["exchange"]["exchange_ms"] < 5.0 and d["exchange"]["digest"] != 0
    # enqueueing a chunk's collective never waits for the GPU (the pipeline must keep running under it)
    assert all(ms < 2.0 for ms in d["exchange"]["host_enqueue_ms_per_collective"])
    # RCCL's first barrier (its one-off set-up: 5-16 ms of idle GPU) was paid before the warm-up, not in the fence in
    # front of the timed region: that fence only waits for the two warm-up steps (profiles/r04_one_rank_rccl_fences.txt)
    f = d["fences"]
    assert f["start_barrier_ms"] + f["start_synchronize_ms"] < 6.0
    assert f["end_barrier_ms"] + f["end_synchronize_ms"] <= d["ms_per_step"] * d["steps"] * 1.001


@pytest.mark.parametrize("world,n", [(2, 9), (3, 2)])          # uneven shards; 3 ranks for 2 clips: an empty shard
def test_extract_all_features_two_ranks_real_reservoir(tmp_path, world, n):
    """extract_all_features under a launcher with the REAL reservoir: the ranks share cuda:0, the
    feature rows travel through gloo, every rank ends with all rows in dataset order = the single-process run."""
    import socket
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dist_worker
    import extract_lsm_features as ex
    rs = np.random.RandomState(1)
    clips = (rs.rand(n, 4, 120) < 0.4).astype(np.uint8)
    single = ex.extract_all_features(_dist_worker.real_lsm(), clips, ["spike_counts", "mean_isi"], "")
    assert single.shape == (n, 200) and single.sum() > 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path),
                                       str(n), "gpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for rank in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"rank{rank}.npz")["feats"], single)
