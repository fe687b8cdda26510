"""GPU checks of the multi-GPU code paths that one GPU can exercise.

  * data-parallel training path (gradient GEMM -> all-reduce -> shared step, Nnet::SetDataParallel)
    with the loopback communicator: n replicas that hold the same bunch X must train exactly like one
    process on n stacked copies of X with gradient averaging over n*|X| frames.
  * utterance-sharded nnet-forward: two rank processes (no collective on this path) share the GPU;
    the union of their outputs equals the single-process output bit for bit.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import nnet1_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


@pytest.fixture(scope="module")
def nn(hip):
    import kaldi_amd.nnet as n
    return n


def _model_bytes(rs, dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(O.BiasedLinearity((0.2 * rs.standard_normal((dims[i + 1], dims[i]))).astype(f32),
                                        (0.1 * rs.standard_normal(dims[i + 1])).astype(f32)))
        layers.append(O.Sigmoid(dims[i + 1]) if i + 2 < len(dims) else O.Softmax(dims[i + 1]))
    return O.write_nnet(layers, True), layers


@pytest.mark.parametrize("mom,avg", [(0.0, True), (0.5, True), (0.0, False)])
@pytest.mark.parametrize("exchange", ["sum", "gather", "shard"])
def test_data_parallel_path_equals_stacked_bunch(hip, nn, tmp_path, monkeypatch, mom, avg, exchange):
    rs = np.random.RandomState(11)
    dims = [40, 96, 64, 33]
    data, _ = _model_bytes(rs, dims)
    model = tmp_path / "m.nnet"
    model.write_bytes(data)
    N, R, steps, lr = 64, 2, 3, 0.05
    X = rs.standard_normal((steps, N, dims[0])).astype(f32)
    T = rs.randint(0, dims[-1], (steps, N)).astype(np.int32)

    hip.call("hipK_comm_init_loopback", R)
    try:
        dp = nn.Trainer(str(model))
        dp.set_train_opts(lr, mom, 0.0, 0.0, 0.0, avg, True)
        # both exchanges: summed gradients (all-reduce) and gathered rows (all-gather + the fused update)
        monkeypatch.setenv("KALDI_AMD_DP_EXCHANGE", exchange)
        dp.set_data_parallel(R)
        for s in range(steps):
            x = hip.DevMatrix.from_numpy(X[s], pitched=(mom == 0.0))   # one case with tightly packed rows
            t = hip.DevMatrix.from_numpy(T[s], pitched=False)
            dp.train_bunch(x.ptr, x.stride, N, dims[0], t.ptr)
        dp_params = [dp.layer_params(i, dims[i // 2 + 1], dims[i // 2]) for i in (0, 2, 4)]
        dp.close()
    finally:
        hip.call("hipK_comm_destroy")

    one = nn.Trainer(str(model))
    one.set_train_opts(lr, mom, 0.0, 0.0, 0.0, avg, True)
    for s in range(steps):
        x = hip.DevMatrix.from_numpy(np.concatenate([X[s]] * R))
        t = hip.DevMatrix.from_numpy(np.concatenate([T[s]] * R), pitched=False)
        one.train_bunch(x.ptr, x.stride, N * R, dims[0], t.ptr)
    for i, (W, b) in zip((0, 2, 4), dp_params):
        W1, b1 = one.layer_params(i, dims[i // 2 + 1], dims[i // 2])
        np.testing.assert_allclose(W, W1, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(b, b1, rtol=2e-5, atol=2e-6)
    one.close()


def test_forward_sharded_over_two_ranks_equals_single(hip, nn, tmp_path):
    rs = np.random.RandomState(5)
    dims = [30, 64, 21]
    data, _ = _model_bytes(rs, dims)
    (tmp_path / "m.nnet").write_bytes(data)
    lens = [7, 130, 33, 1, 64, 250, 18, 90, 5]
    utts = [("u%02d" % i, rs.standard_normal((n, dims[0])).astype(f32)) for i, n in enumerate(lens)]
    (tmp_path / "f.ark").write_bytes(O.write_matrix_ark(utts))
    exe = os.path.join(nn.BIN_DIR, "nnet-forward")
    base = [exe, "--silent=true", str(tmp_path / "m.nnet"), "ark:" + str(tmp_path / "f.ark")]
    r = subprocess.run(base + ["ark:" + str(tmp_path / "single.ark")], stderr=subprocess.PIPE, timeout=110)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    single = dict(O.read_matrix_ark((tmp_path / "single.ark").read_bytes()))
    want = nn.shard_assign(2, lens)
    # the same table as a script file: "key archive:offset" (offset of the entry's value)
    ark = (tmp_path / "f.ark").read_bytes()
    pos, lines = 0, []
    for k, m in utts:
        pos = ark.index(k.encode() + b" \0B", pos) + len(k) + 1
        lines.append("%s %s:%d\n" % (k, tmp_path / "f.ark", pos))
    (tmp_path / "f.scp").write_text("".join(lines))
    payload = [n * dims[0] * 4 for n in lens]
    for rspec in ("ark:" + str(tmp_path / "f.ark"), "scp:" + str(tmp_path / "f.scp")):
        procs = [subprocess.Popen([exe, "--num-ranks=2", "--rank=%d" % k, "--gpu-id=0", str(tmp_path / "m.nnet"), rspec,
                                   "ark:" + str(tmp_path / "shard.JOB.ark")], stderr=subprocess.PIPE) for k in range(2)]
        read = []
        for p in procs:
            _, err = p.communicate(timeout=110)
            assert p.returncode == 0, err.decode()[-2000:]
            read.append(int(re.search(r"feature payload read (\d+) bytes", err.decode()).group(1)))
        shards = [dict(O.read_matrix_ark((tmp_path / ("shard.%d.ark" % (k + 1))).read_bytes())) for k in range(2)]
        assert not set(shards[0]) & set(shards[1])
        assert set(shards[0]) | set(shards[1]) == set(single)
        for i, (k, _) in enumerate(utts):
            assert k in shards[want[i]]                                    # the documented rule
            np.testing.assert_array_equal(shards[want[i]][k], single[k])   # same kernels, same rows: bit for bit
        # a rank reads the payload of its own utterances only (the others are skipped on their header)
        for k in range(2):
            assert read[k] == sum(b for b, w in zip(payload, want) if w == k), (rspec, read, payload, list(want))


def test_forward_rank_with_an_empty_shard_succeeds(hip, nn, tmp_path):
    """A table with fewer utterances than ranks: the rank that gets nothing ends with status 0 and an empty
    archive (a single process that finds nothing to do fails, like the reference, nnet-forward.cc:177 -- but this
    rank's job has work) -- found by tests/fuzz/shard_fuzz.py.  An empty table still fails on every rank."""
    rs = np.random.RandomState(6)
    dims = [12, 16, 5]
    data, _ = _model_bytes(rs, dims)
    (tmp_path / "m.nnet").write_bytes(data)
    utts = [("u0", rs.standard_normal((9, 12)).astype(f32)), ("u1", rs.standard_normal((4, 12)).astype(f32))]
    (tmp_path / "f.ark").write_bytes(O.write_matrix_ark(utts))
    (tmp_path / "empty.ark").write_bytes(b"")
    exe = os.path.join(nn.BIN_DIR, "nnet-forward")
    got = {}
    for rank in range(3):
        r = subprocess.run([exe, "--num-ranks=3", "--rank=%d" % rank, "--gpu-id=0", str(tmp_path / "m.nnet"),
                            "ark:" + str(tmp_path / "f.ark"), "ark:" + str(tmp_path / "s.JOB.ark")], stderr=subprocess.PIPE, timeout=110)
        assert r.returncode == 0, (rank, r.stderr.decode()[-1500:])
        got[rank] = O.read_matrix_ark((tmp_path / ("s.%d.ark" % (rank + 1))).read_bytes())
    assert sorted(k for v in got.values() for k, _ in v) == ["u0", "u1"] and [len(v) for v in got.values()].count(0) == 1
    r = subprocess.run([exe, "--num-ranks=3", "--rank=1", "--gpu-id=0", str(tmp_path / "m.nnet"), "ark:" + str(tmp_path / "empty.ark"),
                        "ark:/dev/null"], stderr=subprocess.PIPE, timeout=110)
    assert r.returncode == 1


@pytest.mark.parametrize("mode", ["alone", "torch"])
def test_rccl_communicator_single_rank(hip, mode):
    """The real RCCL path of the device library (unique id, ncclCommInitRank, sum / min all-reduce, barrier,
    destroy) with one rank, in a process of its own; "torch": after torch has loaded its own RCCL and opened
    a process group, as bench.py does when launched with several ranks."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "rccl_probe.py")] + (["torch"] if mode == "torch" else [])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=110, env=env)
    assert r.returncode == 0 and b"rccl probe ok" in r.stdout, r.stderr.decode()[-2000:]


def test_dp_tool_loop_uneven_shards_and_report_contract(hip, nn, tmp_path):
    """The multi-rank loop of nnet-train-xent-hardlab-frmshuff, each rank of 2 run on its own with the stand-in
    communicator (KALDI_AMD_COMM_LOOPBACK: sums x2, min / max local): a table on which rank 1's shard runs dry
    before its cache holds one bunch (it must neither abort in Cache::Randomize nor leave the collectives), every
    rank reads only its own payload, and the log keeps the recipes' contract (train_scheduler.sh:85,109,120:
    awk '/FRAME_ACCURACY/{acc=$3}' keeps the LAST match): rank 0 prints the all-rank report as its only
    FRAME_ACCURACY line, in Xent::Report()'s format; the other rank prints none."""
    rs = np.random.RandomState(9)
    dims = [20, 32, 9]
    data, _ = _model_bytes(rs, dims)
    (tmp_path / "m.nnet").write_bytes(data)
    lens = [300, 20]                   # rank 0 gets 300 frames, rank 1 20: less than one bunch of 64
    utts = [("u%02d" % i, rs.standard_normal((n, dims[0])).astype(f32)) for i, n in enumerate(lens)]
    alis = [(k, rs.randint(0, dims[-1], m.shape[0]).astype(np.int32)) for k, m in utts]
    (tmp_path / "f.ark").write_bytes(O.write_matrix_ark(utts))
    (tmp_path / "a.ark").write_bytes(O.write_int32_vector_ark(alis))
    want = nn.shard_assign(2, lens)
    assert list(want) == [0, 1]
    exe = os.path.join(nn.BIN_DIR, "nnet-train-xent-hardlab-frmshuff")
    env = dict(os.environ, KALDI_AMD_COMM_LOOPBACK="1")
    logs = []
    for rank in range(2):
        r = subprocess.run([exe, "--num-ranks=2", "--rank=%d" % rank, "--gpu-id=0", "--bunchsize=64", "--cachesize=512",
                            "--learn-rate=0.01", "ark:" + str(tmp_path / "f.ark"), "ark:" + str(tmp_path / "a.ark"),
                            str(tmp_path / "m.nnet"), str(tmp_path / ("out%d.nnet" % rank))],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=110, env=env)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        logs.append(r.stderr.decode())
        (tmp_path / ("log%d" % rank)).write_text(logs[-1])
        read = int(re.search(r"feature payload read (\d+) bytes", logs[-1]).group(1))
        assert read == sum(n * dims[0] * 4 for n, w in zip(lens, want) if w == rank)
    # rank 0: 300 frames -> 4 whole bunches of 64; the stand-in communicator doubles the sums
    m = re.search(r"Xent:(\S+) frames:(\d+) err/frm:(\S+)\nFRAME_ACCURACY >> (\S+)% <<", logs[0])
    assert m and int(m.group(2)) == 2 * 256, logs[0][-1500:]
    assert logs[0].count("FRAME_ACCURACY") == 1 and "FRAME_ACCURACY" not in logs[1]
    assert "shard frames:0 " in logs[1]                      # rank 1 trained nothing and still finished cleanly
    awk = subprocess.run(["awk", '/FRAME_ACCURACY/{ acc=$3; sub(/%/,"",acc); } END{print acc}', str(tmp_path / "log0")],
                         stdout=subprocess.PIPE).stdout.decode().strip()
    assert abs(float(awk) - float(m.group(4))) < 1e-6
    assert (tmp_path / "out0.nnet").exists() and not (tmp_path / "out1.nnet").exists()   # rank 0 writes the model


@pytest.mark.parametrize("exchange", ["sum", "gather", "shard"])
def test_data_parallel_three_steps_rm_shape_equal_oracle_on_stacked_bunch(hip, nn, tmp_path, monkeypatch, exchange):
    """The three-stream data-parallel step (gradient GEMMs, exchange and shared step beside the back-propagation) over
    three consecutive steps at the RM size, kernels long enough to overlap: two stand-in ranks on a bunch equal the
    numpy oracle on that bunch stacked twice -- weights and biases of every layer, momentum 0.9."""
    rs = np.random.RandomState(2025)
    dims, N, R, lr, mom = [440, 1024, 1024, 1024, 1024, 1483], 256, 2, 0.02, 0.9
    layers = []
    for i in range(len(dims) - 1):
        last = i == len(dims) - 2
        layers.append(O.BiasedLinearity((0.1 * rs.randn(dims[i + 1], dims[i])).astype(f32),
                                        np.zeros(dims[i + 1], f32) if last else (-4.1 + 0.2 * rs.rand(dims[i + 1])).astype(f32)))
        layers.append(O.Softmax(dims[i + 1]) if last else O.Sigmoid(dims[i + 1]))
    model = tmp_path / "m.nnet"
    model.write_bytes(O.write_nnet(layers, True))
    X = rs.randn(3, N, dims[0]).astype(f32)
    T = rs.randint(0, dims[-1], (3, N)).astype(np.int32)
    net = O.Nnet(layers)
    net.set_train_opts(lr, momentum=mom, average_grad=True)
    xent = O.Xent()
    hip.call("hipK_comm_init_loopback", R)
    try:
        dp = nn.Trainer(str(model))
        dp.set_train_opts(lr, mom, 0.0, 0.0, 0.0, True, True)
        monkeypatch.setenv("KALDI_AMD_DP_EXCHANGE", exchange)
        dp.set_data_parallel(R)
        for s in range(3):
            x, t = hip.DevMatrix.from_numpy(X[s]), hip.DevMatrix.from_numpy(T[s], pitched=False)
            dp.train_bunch(x.ptr, x.stride, N, dims[0], t.ptr)
            diff, _, _, _ = xent.eval_vec(net.propagate(np.concatenate([X[s]] * R)), np.concatenate([T[s]] * R))
            net.backpropagate(diff)
        for i in range(0, 2 * (len(dims) - 1), 2):
            W, b = dp.layer_params(i, dims[i // 2 + 1], dims[i // 2])
            np.testing.assert_allclose(W, net.layers[i].W, rtol=2e-4, atol=2e-6 * np.abs(net.layers[i].W).max())
            np.testing.assert_allclose(b, net.layers[i].b, rtol=2e-4, atol=2e-6 * max(np.abs(net.layers[i].b).max(), 1e-3))
        dp.close()
    finally:
        hip.call("hipK_comm_destroy")


@pytest.mark.parametrize("R", [2, 4, 5])
def test_sharded_exchange_gives_the_bits_of_the_summed_one(hip, nn, tmp_path, monkeypatch, R):
    """KALDI_AMD_DP_EXCHANGE=shard (reduce-scatter, step on the own shard, all-gather of the new weights) against
    =sum (all-reduce, whole step everywhere): same weights, biases and momentum state bit for bit after three steps
    with momentum, weight decay and the row-norm bound.  With 5 stand-in ranks no layer cuts evenly: the chunks of
    the parameter arena are padded to a whole number of 256-byte lines per rank."""
    rs = np.random.RandomState(77)
    dims = [40, 96, 64, 33]
    data, _ = _model_bytes(rs, dims)
    model = tmp_path / "m.nnet"
    model.write_bytes(data)
    N, steps = 64, 3
    X = rs.standard_normal((steps, N, dims[0])).astype(f32)
    T = rs.randint(0, dims[-1], (steps, N)).astype(np.int32)
    got = {}
    hip.call("hipK_comm_init_loopback", R)
    try:
        # (the sharded exchange with its layers in one chunk, in the default three, and one layer per chunk)
        for exchange, chunks in (("sum", None), ("shard", "1"), ("shard", "3"), ("shard", "8")):
            monkeypatch.setenv("KALDI_AMD_DP_EXCHANGE", exchange)
            if chunks:
                monkeypatch.setenv("KALDI_AMD_DP_CHUNKS", chunks)
            dp = nn.Trainer(str(model))
            dp.set_train_opts(0.05, 0.7, 1e-4, 0.0, 1.5, True, True)
            dp.set_data_parallel(R)
            for s_ in range(steps):
                x, t = hip.DevMatrix.from_numpy(X[s_]), hip.DevMatrix.from_numpy(T[s_], pitched=False)
                dp.train_bunch(x.ptr, x.stride, N, dims[0], t.ptr)
            # the written model goes through Nnet::Write, which waits for the exchange of the last step
            dp.write(str(tmp_path / ("out_%s_%s.nnet" % (exchange, chunks))))
            got[(exchange, chunks)] = [dp.layer_params(i, dims[i // 2 + 1], dims[i // 2]) for i in (0, 2, 4)]
            dp.close()
    finally:
        hip.call("hipK_comm_destroy")
        monkeypatch.delenv("KALDI_AMD_DP_CHUNKS", raising=False)
    for key in (("shard", "1"), ("shard", "3"), ("shard", "8")):
        for (Wa, ba), (Wb, bb) in zip(got[("sum", None)], got[key]):
            assert np.array_equal(Wa, Wb) and np.array_equal(ba, bb), key
        written = [l for l in O.read_nnet((tmp_path / ("out_%s_%s.nnet" % key)).read_bytes()) if isinstance(l, O.BiasedLinearity)]
        for (Wa, ba), l in zip(got[key], written):
            assert np.array_equal(Wa, l.W) and np.array_equal(ba, l.b), key
    assert not np.array_equal(got[("sum", None)][0][0], np.zeros_like(got[("sum", None)][0][0]))


def test_rccl_entry_points_with_one_rank(nn):
    """Every collective of the library through RCCL itself, on a one-rank communicator (the box has one GPU):
    all-reduce, all-gather, the grouped reduce-scatter + all-reduce and the in-place all-gather of the sharded
    exchange, min / max, barrier."""
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rccl_probe.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=110)
    assert r.returncode == 0 and b"rccl probe ok" in r.stdout, r.stdout.decode()[-2000:]


def _swbd_layers(seed):
    rs = np.random.RandomState(seed)
    dims = [1320] + [2048] * 7 + [9000]
    layers = []
    for i in range(len(dims) - 1):
        last = i == len(dims) - 2
        layers.append(O.BiasedLinearity((0.1 * rs.standard_normal((dims[i + 1], dims[i]))).astype(f32),
                                        np.zeros(dims[i + 1], f32) if last else (-4.1 + 0.2 * rs.random_sample(dims[i + 1])).astype(f32)))
        layers.append(O.Softmax(dims[i + 1]) if last else O.Sigmoid(dims[i + 1]))
    return dims, layers, rs


def test_data_parallel_two_steps_swbd_shape_equal_oracle_on_stacked_bunch(hip, nn, tmp_path, monkeypatch):
    """The data-parallel step at the size it is built for: the Switchboard-sized network (1320 -> 7 x 2048 -> 9000),
    two stand-in ranks with a 256-frame bunch each, two steps with momentum, every exchange -- against the numpy
    oracle on the stacked 512-row bunch (weights and biases of all eight layers), and the three exchanges against
    each other: shard and sum bit for bit (one gradient kernel, one element-wise formula), gather to rounding
    (it sums the 512 rows in one product)."""
    dims, layers, rs = _swbd_layers(2031)
    N, R, lr, mom, steps = 256, 2, 0.008, 0.5, 2
    model = tmp_path / "m.nnet"
    model.write_bytes(O.write_nnet(layers, True))
    X = rs.standard_normal((steps, N, dims[0])).astype(f32)
    T = rs.randint(0, dims[-1], (steps, N)).astype(np.int32)
    net = O.Nnet(layers)
    net.set_train_opts(lr, momentum=mom, average_grad=False)
    xent = O.Xent()
    for s in range(steps):
        diff, _, _, _ = xent.eval_vec(net.propagate(np.concatenate([X[s]] * R)), np.concatenate([T[s]] * R))
        net.backpropagate(diff)
    got = {}
    hip.call("hipK_comm_init_loopback", R)
    try:
        for exchange in ("shard", "sum", "gather"):
            monkeypatch.setenv("KALDI_AMD_DP_EXCHANGE", exchange)
            dp = nn.Trainer(str(model))
            dp.set_train_opts(lr, mom, 0.0, 0.0, 0.0, False, True)
            dp.set_data_parallel(R)
            for s in range(steps):
                x, t = hip.DevMatrix.from_numpy(X[s]), hip.DevMatrix.from_numpy(T[s], pitched=False)
                dp.train_bunch(x.ptr, x.stride, N, dims[0], t.ptr)
            frames, _, loss = dp.stats()
            assert frames == steps * N
            got[exchange] = [dp.layer_params(i, dims[i // 2 + 1], dims[i // 2]) for i in range(0, 2 * (len(dims) - 1), 2)]
            dp.close()
            # (the loss of the rank's own rows: half of the stacked bunch's)
            assert abs(loss - xent.loss / R) < 1e-3 * abs(xent.loss / R), exchange
    finally:
        hip.call("hipK_comm_destroy")
    for exchange, params in got.items():
        for k, (W, b) in enumerate(params):
            ref = net.layers[2 * k]
            np.testing.assert_allclose(W, ref.W, rtol=1e-3, atol=2e-6 * np.abs(ref.W).max(), err_msg="%s layer %d" % (exchange, k))
            np.testing.assert_allclose(b, ref.b, rtol=1e-3, atol=2e-6 * max(np.abs(ref.b).max(), 1e-3), err_msg="%s layer %d" % (exchange, k))
    for (Wa, ba), (Wb, bb) in zip(got["shard"], got["sum"]):
        assert np.array_equal(Wa, Wb) and np.array_equal(ba, bb)


def test_forward_sharded_over_eight_ranks_swbd_shape_equals_single(hip, nn, tmp_path):
    """nnet-forward --num-ranks=8 on the Switchboard-sized network: eight rank processes share the GPU (the path
    has no collective), 44 utterances of 20..160 frames.  The union of their outputs equals the single-process
    archive bit for bit -- the ranks cut the table into other batches than one process does, and the row-invariant
    products make that invisible -- every utterance sits with the rank the rule names, and the ranks read the
    feature payload exactly once between them."""
    dims, layers, rs = _swbd_layers(2032)
    (tmp_path / "m.nnet").write_bytes(O.write_nnet(layers, True))
    lens = [int(n) for n in rs.randint(20, 161, 44)]
    utts = [("utt%02d" % i, rs.standard_normal((n, dims[0])).astype(f32)) for i, n in enumerate(lens)]
    (tmp_path / "f.ark").write_bytes(O.write_matrix_ark(utts))
    exe = os.path.join(nn.BIN_DIR, "nnet-forward")
    R = 8
    r = subprocess.run([exe, "--silent=true", str(tmp_path / "m.nnet"), "ark:" + str(tmp_path / "f.ark"),
                        "ark:" + str(tmp_path / "single.ark")], stderr=subprocess.PIPE, timeout=110)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    single = dict(O.read_matrix_ark((tmp_path / "single.ark").read_bytes()))
    assert len(single) == len(utts)
    procs = [subprocess.Popen([exe, "--num-ranks=%d" % R, "--rank=%d" % k, "--gpu-id=0", "--batch-frames=512", str(tmp_path / "m.nnet"),
                               "ark:" + str(tmp_path / "f.ark"), "ark:" + str(tmp_path / "shard.JOB.ark")], stderr=subprocess.PIPE)
             for k in range(R)]
    read = []
    for p in procs:
        _, err = p.communicate(timeout=115)
        assert p.returncode == 0, err.decode()[-2000:]
        read.append(int(re.search(r"feature payload read (\d+) bytes", err.decode()).group(1)))
    want = nn.shard_assign(R, lens)
    shards = [dict(O.read_matrix_ark((tmp_path / ("shard.%d.ark" % (k + 1))).read_bytes())) for k in range(R)]
    assert sum(len(s) for s in shards) == len(utts)
    for i, (k, _) in enumerate(utts):
        assert k in shards[want[i]]
        np.testing.assert_array_equal(shards[want[i]][k], single[k])
    payload = [n * dims[0] * 4 for n in lens]
    assert sum(read) == sum(payload)
    for k in range(R):
        assert read[k] == sum(b for b, w in zip(payload, want) if w == k)
    # one oracle check on top of the bit-equality: a short utterance's posteriors
    i = int(np.argmin(lens))
    np.testing.assert_allclose(single[utts[i][0]], O.Nnet(layers).feedforward(utts[i][1]), rtol=1e-4, atol=1e-9)


def _device_count(hip):
    import ctypes
    n = ctypes.c_int(0)
    hip.call("hipK_device_count", ctypes.byref(n))
    return n.value


def _shared_gpu_env():
    """extra environment of the rank processes (KALDI_TEST_RANK_ENV="A=1,B=2"; diagnostics)"""
    e = {}
    for kv in os.environ.get("KALDI_TEST_RANK_ENV", "").split(","):
        if kv:
            e[kv.split("=")[0]] = kv.split("=")[1]
    return e


# Rank processes that TRAIN together on one shared GPU are opt-in (KALDI_TEST_MULTIPROC=1): on the boxes these tests
# were developed on, two processes that alternate short bursts of GPU work with waiting for each other saw the
# whole GPU stop for 3-55 s at a time (the runtime's log shows a 2 KB copy "executing" for 55 s of GPU time, and
# both processes released at the same microsecond), so a run takes anything from 1 s to minutes.  The results are
# right whenever the run ends; the stalls are not the library's.  A log of a passing run is kept under profiles/.
multiproc = pytest.mark.skipif(not os.environ.get("KALDI_TEST_MULTIPROC"),
                               reason="rank processes sharing one GPU stall for up to a minute on this pool; set KALDI_TEST_MULTIPROC=1")


def _wait_ranks(procs, seconds, err_files):
    """Waits for the rank processes together: the first one that fails (or the time limit) ends the others -- which
    would otherwise sit in a collective until the communicator gives up -- and the ranks' stderr (written to
    err_files) is the assertion message.  -> [stderr text of rank r]"""
    import time
    t0, bad = time.time(), None
    while any(p.poll() is None for p in procs) and bad is None:
        bad = next((p for p in procs if p.poll() not in (None, 0)), None)
        if time.time() - t0 > seconds:
            bad = next(p for p in procs if p.poll() is None)
        time.sleep(0.05)
    bad = bad or next((p for p in procs if p.returncode != 0), None)
    for p in procs:
        if p.poll() is None:
            p.kill()
        p.wait()
    errs = [open(f, "rb").read().decode(errors="replace") for f in err_files]
    if os.environ.get("KALDI_TEST_KEEP_RANK_LOGS"):   # (diagnostics: every rank's stderr, also of runs that pass)
        for r, e in enumerate(errs):
            with open("%s_%d_r%d.log" % (os.environ["KALDI_TEST_KEEP_RANK_LOGS"], int(t0), r), "w") as f:
                f.write(e)
    assert bad is None, "rank %d: rc %s\n%s" % (procs.index(bad), bad.returncode,
                                                  "\n".join("--- rank %d ---\n%s" % (r, e[-1500:]) for r, e in enumerate(errs)))
    return errs


def _run_host_ranks(tmp_path, tag, model, data, R, env_extra, transport="host"):
    """R processes of tests/dp_host_rank.py over one fresh exchange file -> [params of rank r].  transport "host": all on
    GPU 0 with the host-staged communicator; "rccl": rank r on GPU r, RCCL."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    xfile = "/dev/shm/kaldi_amd_test_%d_%s" % (os.getpid(), tag)
    if os.path.exists(xfile):
        os.unlink(xfile)
    env = dict(os.environ, KALDI_HIP_COMM_TIMEOUT="150", **_shared_gpu_env(), **env_extra)
    try:
        procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "dp_host_rank.py"), str(model), str(data), str(r),
                                   str(R), ("rccl:" if transport == "rccl" else "") + xfile, str(tmp_path / ("%s_r%d.npz" % (tag, r)))],
                                  stdout=subprocess.DEVNULL, stderr=open(tmp_path / ("%s_r%d.err" % (tag, r)), "wb"), env=env)
                 for r in range(R)]
        _wait_ranks(procs, 400, [tmp_path / ("%s_r%d.err" % (tag, r)) for r in range(R)])
    finally:
        if os.path.exists(xfile):
            os.unlink(xfile)
    return [dict(np.load(tmp_path / ("%s_r%d.npz" % (tag, r)))) for r in range(R)]


@multiproc
@pytest.mark.parametrize("R,transport", [(2, "host"), (3, "host"), (2, "rccl")])
def test_real_ranks_with_their_own_rows_train_like_one_process_on_all_rows(hip, nn, tmp_path, R, transport):
    """R rank PROCESSES, each with its own third / half of every bunch.  "host": all on this GPU, joined by the
    host-staged communicator (real reduce-scatter / all-gather / all-reduce semantics: rank-dependent shard offsets,
    in-place buffers, pieces smaller than a layer); "rccl": one GPU per rank over RCCL, where the box has two.
    After three steps with momentum, weight decay and the row-norm bound: every rank holds the same bits; the
    sharded exchange (1 and 3 chunks) equals the summed one -- bit for bit with the host transport, which adds in
    rank order in both, to rounding with RCCL; and the result equals one process trained on the whole bunches, and
    the numpy oracle, to rounding."""
    if transport == "rccl" and _device_count(hip) < R:
        pytest.skip("needs %d GPUs" % R)
    rs = np.random.RandomState(123 + R)
    dims = [40, 96, 64, 50, 33]
    data, layers = _model_bytes(rs, dims)
    model = tmp_path / "m.nnet"
    model.write_bytes(data)
    per, steps = 32, 3
    N = per * R
    X = rs.standard_normal((steps, N, dims[0])).astype(f32)
    T = rs.randint(0, dims[-1], (steps, N)).astype(np.int32)
    opts = np.array([0.05, 0.7, 1e-4, 0.0, 1.5], f32)
    np.savez(tmp_path / "d.npz", X=X, T=T, split=np.arange(R + 1) * per, opts=opts, dims=np.array(dims))
    got = {}
    for exchange, chunks in (("sum", "3"), ("shard", "1"), ("shard", "3"), ("gather", "3")):
        tag = "%s%s" % (exchange, chunks)
        got[tag] = _run_host_ranks(tmp_path, tag, model, tmp_path / "d.npz", R,
                                   {"KALDI_AMD_DP_EXCHANGE": exchange, "KALDI_AMD_DP_CHUNKS": chunks}, transport)
        for r in range(1, R):
            for k in got[tag][0]:
                assert np.array_equal(got[tag][0][k], got[tag][r][k]), (tag, r, k)      # replicas stay identical
        written = [l for l in O.read_nnet((tmp_path / (tag + "_r0.npz.nnet")).read_bytes()) if isinstance(l, O.BiasedLinearity)]
        for k, l in enumerate(written):
            assert np.array_equal(l.W, got[tag][0]["W%d" % k]) and np.array_equal(l.b, got[tag][0]["b%d" % k])
    for tag in ("shard1", "shard3"):
        for k in got["sum3"][0]:
            if transport == "host":
                assert np.array_equal(got["sum3"][0][k], got[tag][0][k]), (tag, k)
            else:
                np.testing.assert_allclose(got["sum3"][0][k], got[tag][0][k], rtol=1e-5, atol=1e-6)
    # one process on the whole bunches
    one = nn.Trainer(str(model))
    one.set_train_opts(*[float(v) for v in opts], True, True)
    net = O.Nnet(layers)
    net.set_train_opts(float(opts[0]), momentum=float(opts[1]), l2_penalty=float(opts[2]), l2_upper_bound=float(opts[4]), average_grad=True)
    xent = O.Xent()
    for s in range(steps):
        x, t = hip.DevMatrix.from_numpy(X[s]), hip.DevMatrix.from_numpy(T[s], pitched=False)
        one.train_bunch(x.ptr, x.stride, N, dims[0], t.ptr)
        diff, _, _, _ = xent.eval_vec(net.propagate(X[s]), T[s])
        net.backpropagate(diff)
    for k in range(len(dims) - 1):
        W1, b1 = one.layer_params(2 * k, dims[k + 1], dims[k])
        for ref_W, ref_b in ((W1, b1), (net.layers[2 * k].W, net.layers[2 * k].b)):
            np.testing.assert_allclose(got["shard3"][0]["W%d" % k], ref_W, rtol=1e-4, atol=2e-6)
            np.testing.assert_allclose(got["shard3"][0]["b%d" % k], ref_b, rtol=1e-4, atol=2e-6)
        assert not np.array_equal(W1, layers[2 * k].W)
        # the gather exchange runs the single-process update on all rows: the bits of the single process
        # (max-norm bound off: with it the fused update is a different kernel sequence, equal to rounding)
        np.testing.assert_allclose(got["gather3"][0]["W%d" % k], W1, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(got["gather3"][0]["b%d" % k], b1, rtol=1e-4, atol=2e-6)
    one.close()


@multiproc
def test_two_rank_training_job_on_one_gpu_shard_equals_sum(hip, nn, tmp_path):
    """nnet-train-xent-hardlab-frmshuff as a real 2-rank job (two processes, each reading its own utterances, joined
    by the host-staged communicator, KALDI_AMD_COMM_HOST_FILE): the job finishes although the shards are uneven,
    rank 0's report counts the frames of both ranks, and the model written with the sharded exchange has the bits
    of the one written with the summed exchange -- and differs from the input model."""
    rs = np.random.RandomState(19)
    dims = [20, 48, 32, 9]
    data, layers = _model_bytes(rs, dims)
    (tmp_path / "m.nnet").write_bytes(data)
    lens = [200, 90, 75, 130, 40, 61]
    utts = [("u%02d" % i, rs.standard_normal((n, dims[0])).astype(f32)) for i, n in enumerate(lens)]
    alis = [(k, rs.randint(0, dims[-1], m.shape[0]).astype(np.int32)) for k, m in utts]
    (tmp_path / "f.ark").write_bytes(O.write_matrix_ark(utts))
    (tmp_path / "a.ark").write_bytes(O.write_int32_vector_ark(alis))
    want = nn.shard_assign(2, lens)
    frames = [sum(n for n, w in zip(lens, want) if w == r) for r in range(2)]
    exe = os.path.join(nn.BIN_DIR, "nnet-train-xent-hardlab-frmshuff")
    models = {}
    for exchange in ("sum", "shard"):
        xfile = "/dev/shm/kaldi_amd_test_%d_job_%s" % (os.getpid(), exchange)
        if os.path.exists(xfile):
            os.unlink(xfile)
        env = dict(os.environ, KALDI_AMD_COMM_HOST_FILE=xfile, KALDI_AMD_DP_EXCHANGE=exchange, KALDI_HIP_COMM_TIMEOUT="150",
                   **_shared_gpu_env())
        try:
            procs = [subprocess.Popen([exe, "--num-ranks=2", "--rank=%d" % r, "--gpu-id=0", "--bunchsize=32", "--cachesize=128",
                                       "--learn-rate=0.02", "--momentum=0.5", "ark:" + str(tmp_path / "f.ark"),
                                       "ark:" + str(tmp_path / "a.ark"), str(tmp_path / "m.nnet"),
                                       str(tmp_path / ("out_%s_%d.nnet" % (exchange, r)))],
                                      stdout=subprocess.DEVNULL, stderr=open(tmp_path / ("%s_r%d.err" % (exchange, r)), "wb"), env=env)
                     for r in range(2)]
            logs = _wait_ranks(procs, 400, [tmp_path / ("%s_r%d.err" % (exchange, r)) for r in range(2)])
        finally:
            if os.path.exists(xfile):
                os.unlink(xfile)
        m = re.search(r"Xent:(\S+) frames:(\d+) err/frm:(\S+)\nFRAME_ACCURACY >> (\S+)% <<", logs[0])
        assert m, logs[0][-1500:]
        total = int(m.group(2))
        # every rank trains the same number of whole bunches: at most what the smaller shard holds
        assert 0 < total <= 2 * min(frames) and total % 64 == 0, (total, frames)
        assert "FRAME_ACCURACY" not in logs[1]
        models[exchange] = (tmp_path / ("out_%s_0.nnet" % exchange)).read_bytes()
    assert models["sum"] == models["shard"]
    assert models["sum"] != (tmp_path / "m.nnet").read_bytes()
