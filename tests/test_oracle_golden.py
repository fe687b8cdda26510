"""Pins the CPU oracle (oracle/nnet1_oracle.py) against fixtures produced by the REFERENCE build
(tests/golden/make_golden.py -> oracle/_ref).  Runs without a GPU.

Tolerances: the oracle forms GEMM sums in float64 and rounds once while the reference sums in
float32 inside BLAS, so tensors agree to a few float32 ulp of the accumulated magnitude; integer
results (arg-max ids, frame / correct counts, cache order) must agree exactly.
"""
import os

import numpy as np
import pytest

from oracle import nnet1_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


def load(name):
    return np.load(os.path.join(G, name))


def close(a, b, rtol=2e-5, atol=2e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def close_scaled(a, b, rel=1e-5):
    """agreement relative to the largest entry: gradients span many orders of magnitude and each
    entry carries the rounding error of a sum whose terms are as large as the largest ones"""
    b = np.asarray(b)
    np.testing.assert_allclose(np.asarray(a), b, rtol=rel * 10, atol=rel * max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("cfg", ["plain", "mom_avg", "l1", "l2ub"])
@pytest.mark.parametrize("enc", ["model_bin", "model_txt"])
def test_dnn_steps_match_reference(cfg, enc):
    g = load("dnn_small_%s.npz" % cfg)
    layers = O.read_nnet(g[enc].tobytes())
    assert [l.marker for l in layers] == ["<biasedlinearity>", "<sigmoid>"] * 2 + ["<biasedlinearity>", "<softmax>"]
    net = O.Nnet(layers)
    net.set_train_opts(float(g["opt_lr"]), float(g["opt_mom"]), float(g["opt_l2"]), float(g["opt_l1"]),
                       float(g["opt_l2ub"]), bool(g["opt_avg"]))
    xent = O.Xent()
    tol = dict(rtol=2e-5, atol=2e-6) if enc == "model_bin" else dict(rtol=5e-4, atol=5e-6)
    for s in range(int(g["opt_nsteps"])):
        post = net.propagate(g["x"])
        for i, buf in enumerate(net.propagate_buf):
            close(buf, g["s%d_prop%d" % (s, i)], **tol)
        close(post, g["s%d_post" % s], **tol)
        diff, log_post, max_id, correct = xent.eval_vec(post, g["tgt"])
        close(diff, g["s%d_globerr" % s], **tol)
        assert xent.frames == int(g["s%d_cumframes" % s])
        # the reference prints 6 significant digits
        assert abs(xent.loss - float(g["s%d_cumloss" % s])) <= 2e-5 * abs(xent.loss) + 1e-4
        assert abs(100.0 * xent.correct / xent.frames - float(g["s%d_cumacc" % s])) < 1e-3
        net.backpropagate(diff)
        for i, bb in enumerate(net.backpropagate_buf):
            close_scaled(bb, g["s%d_bprop%d" % (s, i)], tol["rtol"])
        for i, l in enumerate(net.layers):
            if isinstance(l, O.BiasedLinearity):
                close(l.W, g["s%d_W%d" % (s, i)], **tol)
                close(l.b, g["s%d_b%d" % (s, i)][0], **tol)
                close_scaled(l.W_corr, g["s%d_Wcorr%d" % (s, i)], tol["rtol"])
                close_scaled(l.b_corr, g["s%d_bcorr%d" % (s, i)][0], tol["rtol"])


def test_model_writers_match_reference_bytes():
    """Text and binary model files written by the oracle are byte-identical to the reference's."""
    g = load("train_e2e_small.npz")
    for key, binary in [("norand_model_out", True), ("norand_model_out_txt", False)]:
        data = g[key].tobytes()
        layers = O.read_nnet(data)
        again = O.write_nnet(layers, binary)
        assert again == data, "round trip of %s differs" % key


def test_forward_tool_matches_reference():
    g = load("forward_small.npz")
    net = O.Nnet(O.read_nnet(g["model"].tobytes()))
    net_nosm = O.Nnet(O.read_nnet(g["model_nosm"].tobytes()))
    counts = g["counts"].astype(f32)
    pri = (counts * f32(1.0 / counts.astype(np.float64).sum())).astype(f32)   # nnet-forward.cc:94-95
    for k in ["utt0", "utt1", "utt2", "utt3"]:
        x = g["in_" + k]
        post = net.feedforward(x)
        close(post, g["out_plain_" + k], rtol=1e-4, atol=1e-9)
        close(np.log(post), g["out_log_" + k], rtol=1e-4, atol=1e-5)
        close(post * np.power(pri, f32(-1.0)), g["out_counts_" + k], rtol=1e-4, atol=1e-9)
        close(np.log(post) + f32(-0.7) * np.log(pri), g["out_counts_log_scale_" + k], rtol=1e-4, atol=1e-5)
        close(net_nosm.feedforward(x) + f32(-1.0) * np.log(pri), g["out_nosoftmax_counts_" + k], rtol=1e-4, atol=1e-5)
    # the archive layout: text-mode output parses back to the binary-mode values
    txt = dict(O.read_matrix_ark(g["out_plain_text_ark"].tobytes()))
    for k in txt:
        close(txt[k], g["out_plain_" + k], rtol=1e-5, atol=1e-9)


def run_trainer(g, opts, randomize):
    """nnetbin/nnet-train-xent-hardlab-frmshuff.cc:111-181 with bunchsize 16, cachesize 64."""
    net = O.Nnet(O.read_nnet(g["model"].tobytes()))
    net.set_train_opts(**opts)
    feats = O.read_matrix_ark(g["feats_ark"].tobytes())
    alis = {}
    for line in g["ali_ark_txt"].tobytes().decode().splitlines():
        parts = line.split()
        alis[parts[0]] = np.array(parts[1:], np.int32)
    cache = O.Cache(64, 16)
    xent = O.Xent()
    cache_lines, pos, num_done, tot = [], 0, 0, 0
    O.srand(1)  # an unseeded process starts from srand(1)
    while True:
        while not cache.full() and pos < len(feats):
            key, mat = feats[pos]
            cache.add_data(mat, alis[key])
            num_done += 1
            pos += 1
        if randomize:
            mask = list(range(cache.filling_pos))
            for i in range(1, len(mask)):       # std::random_shuffle with RandInt(0, i)
                j = O.rand_int(0, i)
                mask[i], mask[j] = mask[j], mask[i]
            cache.randomize(mask)
        cache_lines.append("Cache #%d %s segments: %d frames: %d" % (len(cache_lines) + 1,
                                                                      "[RND]" if cache.randomized else "[NO-RND]",
                                                                      num_done, tot))
        while not cache.empty():
            x, t = cache.get_bunch()
            post = net.propagate(x)
            diff, _, _, _ = xent.eval_vec(post, t)
            if opts["learn_rate"] > 0:
                net.backpropagate(diff)
            tot += x.shape[0]
        if pos >= len(feats):
            break
    return net, xent, cache_lines


def _transform(g):
    return O.Nnet(O.read_nnet(g["transf_bin"].tobytes()))


def test_feature_transform_matches_reference():
    """<expand> / <copy> components and --feature-transform (nnet/nnet-various.h:35-135, nnetbin/nnet-forward.cc:66-75,
    127-131): the oracle against the reference's nnet-copy (both file formats, byte for byte: offsets stored as a
    double vector, <copy> indices 1-based on disk), its nnet-forward on the transform alone and in front of a
    network, and its trainer with the transform in front."""
    g = load("transform_small.npz")
    for enc in ("transf_bin", "transf_txt"):
        layers = O.read_nnet(g[enc].tobytes())
        assert [type(l).__name__ for l in layers] == ["Expand", "Copy", "BiasedLinearity", "Sigmoid"]
        assert layers[0].offsets == list(g["offsets"]) and layers[1].indices == list(g["pick0"])
        assert O.write_nnet(layers, True) == g["transf_bin"].tobytes()
        assert O.write_nnet(layers, False) == g["transf_txt"].tobytes()
    transf, net = _transform(g), O.Nnet(O.read_nnet(g["model"].tobytes()))
    for key, mat in O.read_matrix_ark(g["feats_ark"].tobytes()):
        t = transf.propagate(mat)
        close(t, g["transformed_" + key])
        for enc in ("bin", "txt"):
            np.testing.assert_allclose(net.propagate(t), g["post_%s_%s" % (enc, key)], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("name,opts,randomize", [("train", dict(learn_rate=0.008), False),
                                                 ("train_rand", dict(learn_rate=0.008), True),
                                                 ("cv", dict(learn_rate=0.0), False)])
def test_trainer_with_feature_transform_matches_reference(name, opts, randomize):
    g = dict(load("transform_small.npz"))
    transf = _transform(g)
    # the trainer sees the transformed utterances (nnet-train-xent-hardlab-frmshuff.cc:148)
    g["feats_ark"] = np.frombuffer(O.write_matrix_ark([(k, transf.propagate(m)) for k, m in
                                                       O.read_matrix_ark(g["feats_ark"].tobytes())]), np.uint8)
    net, xent, _ = run_trainer(g, opts, randomize)
    assert xent.frames == int(g[name + "_frames"])
    assert abs(xent.loss - float(g[name + "_xent"])) <= 2e-5 * abs(xent.loss)
    assert abs(100.0 * xent.correct / xent.frames - float(g[name + "_acc"])) < 1e-3
    if name != "cv":
        for a, b in zip(net.layers, O.read_nnet(g[name + "_model_out"].tobytes())):
            if isinstance(a, O.BiasedLinearity):
                close(a.W, b.W, rtol=5e-5, atol=5e-6)
                close(a.b, b.b, rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("name,opts,randomize", [
    ("norand", dict(learn_rate=0.008), False),
    ("rand", dict(learn_rate=0.008), True),
    ("norand_mom", dict(learn_rate=0.1, momentum=0.9, average_grad=True), False),
    ("cv", dict(learn_rate=0.0), False),
])
def test_trainer_end_to_end_matches_reference(name, opts, randomize):
    g = load("train_e2e_small.npz")
    net, xent, cache_lines = run_trainer(g, opts, randomize)
    assert "\n".join(cache_lines) == g[name + "_cache_lines"].tobytes().decode()   # fill / carry-over sequence
    assert xent.frames == int(g[name + "_frames"])                                   # dropped partial bunches
    assert abs(xent.loss - float(g[name + "_xent"])) <= 2e-5 * abs(xent.loss)
    assert abs(100.0 * xent.correct / xent.frames - float(g[name + "_acc"])) < 1e-3
    assert xent.report().split("\n")[1] == g[name + "_report"].tobytes().decode().split("\n")[1]
    if name != "cv":
        ref_layers = O.read_nnet(g[name + "_model_out"].tobytes())
        for a, b in zip(net.layers, ref_layers):
            if isinstance(a, O.BiasedLinearity):
                close(a.W, b.W, rtol=5e-5, atol=5e-6)
                close(a.b, b.b, rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("name", ["bern_bern", "gauss_bern", "bern_gauss"])
def test_rbm_cd1_matches_reference(name):
    g = load("rbm_small_%s.npz" % name)
    (rbm,) = O.read_nnet(g["model"].tobytes())
    rbm.learn_rate, rbm.momentum, rbm.l2_penalty = f32(g["lr"]), f32(g["mom"]), f32(g["l2"])
    for s in range(3):
        out = O.rbm_cd1_step(rbm, g["x"], uniform=g["u%d" % s], gauss=g["g%d" % s])
        close(out["pos_hid"], g["s%d_pos_hid" % s])
        # a probability within rounding of its uniform draw may flip; none does in the fixture
        if rbm.hid_type == "bern":
            assert (out["neg_hid_sample"] == g["s%d_neg_hid_sample" % s]).all()
        else:
            close(out["neg_hid_sample"], g["s%d_neg_hid_sample" % s])
        close(out["neg_vis"], g["s%d_neg_vis" % s])
        close(out["neg_hid"], g["s%d_neg_hid" % s])
        close(rbm.W, g["s%d_W" % s])
        close(rbm.vis_bias, g["s%d_vis_bias" % s][0])
        close(rbm.hid_bias, g["s%d_hid_bias" % s][0])
        close(out["neg_vis"] - g["x"], g["s%d_mse_diff" % s], atol=1e-6)
    # WriteAsNnet (rbm-convert-to-nnet path), text form
    as_nnet = O.BiasedLinearity(rbm.W, rbm.hid_bias)
    layers = [as_nnet] + ([O.Sigmoid(rbm.dim_out)] if rbm.hid_type == "bern" else [])
    ref = O.read_nnet(g["as_nnet_txt"].tobytes())
    assert [l.marker for l in ref] == [l.marker for l in layers]
    close(ref[0].W, rbm.W, rtol=1e-5)


def test_rm_shape_step_matches_reference():
    g = load("rm_shape_step.npz")
    rs = np.random.RandomState(int(g["seed"]))
    dims, N = list(g["dims"]), int(g["N"])
    layers = []
    for i in range(len(dims) - 1):
        W = (0.1 * rs.randn(dims[i + 1], dims[i])).astype(f32)
        last = i == len(dims) - 2
        b = np.zeros(dims[i + 1], f32) if last else (-4.1 + 0.2 * rs.rand(dims[i + 1])).astype(f32)
        layers.append(O.BiasedLinearity(W, b))
        layers.append(O.Softmax(dims[i + 1]) if last else O.Sigmoid(dims[i + 1]))
    x = rs.randn(N, dims[0]).astype(f32)
    tgt = rs.randint(0, dims[-1], N).astype(np.int32)
    net = O.Nnet(layers)
    net.set_train_opts(0.008)
    post = net.propagate(x)
    np.testing.assert_allclose(post[:, :16], g["post_cols"], rtol=1e-4)
    np.testing.assert_allclose(post[np.arange(N), tgt], g["post_tgt"], rtol=1e-4)
    assert (post.argmax(1) == g["post_argmax"]).all()
    xent = O.Xent()
    diff, _, _, _ = xent.eval_vec(post, tgt)
    assert abs(xent.loss - float(g["cumloss"])) <= 1e-5 * abs(xent.loss)
    assert abs(100.0 * xent.correct / N - float(g["cumacc"])) < 1e-3
    net.backpropagate(diff)
    for i in range(0, 2 * (len(dims) - 1), 2):
        l = layers[i]
        # north_star bar: gradients within 1e-3 relative (of the gradient's scale)
        close_scaled(l.W_corr[:4], g["grad%d_rows" % i], 1e-4)
        G = l.W_corr
        close_scaled(np.array([G[j % G.shape[0], (7 * j) % G.shape[1]] for j in range(512)]), g["grad%d_diag" % i], 1e-4)
        assert abs(np.abs(l.W_corr.astype(np.float64)).sum() - float(g["grad%d_abs_sum" % i])) < 1e-4 * float(g["grad%d_abs_sum" % i])
        close_scaled(l.b_corr, g["bgrad%d" % i], 1e-4)
        close(l.W[:4], g["W%d_rows" % i], rtol=1e-5, atol=1e-7)
        close(l.b, g["b%d" % i], rtol=1e-5, atol=1e-7)


def _alis(g):
    alis = {}
    for line in g["ali_ark_txt"].tobytes().decode().splitlines():
        parts = line.split()
        alis[parts[0]] = np.array(parts[1:], np.int32)
    return alis


def _check_models(net, ref_bytes, rtol=5e-5, atol=5e-6):
    for a, b in zip(net.layers, O.read_nnet(ref_bytes)):
        if isinstance(a, O.BiasedLinearity):
            close(a.W, b.W, rtol=rtol, atol=atol)
            close(a.b, b.b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name,opts", [("perutt", dict(learn_rate=0.008)),
                                       ("perutt_mom", dict(learn_rate=0.02, momentum=0.5))])
def test_perutt_trainer_matches_reference(name, opts):
    """nnet-train-xent-hardlab-perutt: one update per utterance, in archive order."""
    g = load("siblings_small.npz")
    net = O.Nnet(O.read_nnet(g["model"].tobytes()))
    net.set_train_opts(**opts)
    alis, xent = _alis(g), O.Xent()
    for key, mat in O.read_matrix_ark(g["feats_ark"].tobytes()):
        diff, _, _, _ = xent.eval_vec(net.propagate(mat), alis[key])
        net.backpropagate(diff)
    assert xent.frames == int(g[name + "_frames"])
    assert abs(xent.loss - float(g[name + "_xent"])) <= 2e-5 * abs(xent.loss)
    assert xent.report().split("\n")[1] == g[name + "_report"].tobytes().decode().split("\n")[1]
    _check_models(net, g[name + "_model_out"].tobytes())


@pytest.mark.parametrize("name,opts", [("mse", dict(learn_rate=0.008)), ("mse_cv", dict(learn_rate=0.0)),
                                       ("mse_mom", dict(learn_rate=0.05, momentum=0.9, average_grad=True))])
def test_mse_trainer_matches_reference(name, opts):
    """nnet-train-mse-tgtmat-frmshuff: the frame cache with a target matrix per utterance; the
    targets ride through the oracle's cache as row numbers into their concatenation."""
    g = load("siblings_small.npz")
    net = O.Nnet(O.read_nnet(g["ae_model"].tobytes()))
    net.set_train_opts(**opts)
    feats = O.read_matrix_ark(g["feats_ark"].tobytes())
    tgts = O.read_matrix_ark(g["tgt_ark"].tobytes())
    all_t = np.concatenate([m for _, m in tgts])
    starts = np.cumsum([0] + [m.shape[0] for _, m in tgts])
    cache = O.Cache(64, 16)
    cache_lines, pos, num_done, tot, loss = [], 0, 0, 0, 0.0
    O.srand(1)
    while True:
        while not cache.full() and pos < len(feats):
            key, mat = feats[pos]
            assert tgts[pos][0] == key
            cache.add_data(mat, np.arange(starts[pos], starts[pos + 1], dtype=np.int32))
            num_done += 1
            pos += 1
        if name != "mse_cv":
            mask = list(range(cache.filling_pos))
            for i in range(1, len(mask)):
                j = O.rand_int(0, i)
                mask[i], mask[j] = mask[j], mask[i]
            cache.randomize(mask)
        cache_lines.append("Cache #%d %s segments: %d frames: %d" % (len(cache_lines) + 1,
                                                                      "[RND]" if cache.randomized else "[NO-RND]",
                                                                      num_done, tot))
        while not cache.empty():
            x, ids = cache.get_bunch()
            diff, _, l = O.mse_eval(net.propagate(x), all_t[np.asarray(ids)])
            loss += l
            if opts["learn_rate"] > 0:
                net.backpropagate(diff)
            tot += x.shape[0]
        if pos >= len(feats):
            break
    assert "\n".join(cache_lines) == g[name + "_cache_lines"].tobytes().decode()
    assert tot == int(g[name + "_frames"])
    assert abs(loss - float(g[name + "_loss"])) <= 5e-5 * abs(loss)
    if name != "mse_cv":
        _check_models(net, g[name + "_model_out"].tobytes())


def test_nnet_copy_and_rbm_convert_match_reference_bytes():
    g = load("siblings_small.npz")
    layers = O.read_nnet(g["model"].tobytes())
    assert O.write_nnet(layers, False) == g["copy_txt"].tobytes()
    assert O.write_nnet(O.read_nnet(g["copy_txt"].tobytes()), True) == g["copy_bin"].tobytes()
    for name in ("bb", "gb", "bg"):
        rbm = O.read_nnet(g["rbm_" + name].tobytes())[0]
        assert O.write_rbm_as_nnet(rbm, True) == g["rbm_%s_nnet_bin" % name].tobytes()
        assert O.write_rbm_as_nnet(rbm, False) == g["rbm_%s_nnet_txt" % name].tobytes()


def test_libc_rand_helpers():
    O.srand(1)
    a = [O.rand_int(0, 99) for _ in range(5)]
    O.srand(1)
    assert a == [O.rand_int(0, 99) for _ in range(5)]
    # glibc's first value for the default seed is 1804289383
    O.srand(1)
    assert O._rand() == 1804289383


def _rm_net(seed):
    rs = np.random.RandomState(seed)
    dims = [440, 1024, 1024, 1024, 1024, 1483]
    N = 256
    layers = []
    for i in range(len(dims) - 1):
        W = (0.1 * rs.randn(dims[i + 1], dims[i])).astype(f32)
        last = i == len(dims) - 2
        b = np.zeros(dims[i + 1], f32) if last else (-4.1 + 0.2 * rs.rand(dims[i + 1])).astype(f32)
        layers.append(O.BiasedLinearity(W, b))
        layers.append(O.Softmax(dims[i + 1]) if last else O.Sigmoid(dims[i + 1]))
    return dims, N, layers, rs.randn(N, dims[0]).astype(f32), rs.randint(0, dims[-1], N).astype(np.int32)


@pytest.mark.parametrize("name,mom", [("mom0", 0.0), ("mom9", 0.9)])
def test_rm_shape_multistep_matches_reference(name, mom):
    """three consecutive steps at the RM size against the reference's (ref_dump): the oracle stays pinned at this
    size when one step's updated weights feed the next step's forward pass"""
    g = load("rm_shape_multistep.npz")
    dims, N, layers, x, tgt = _rm_net(int(g["seed"]))
    net = O.Nnet(layers)
    net.set_train_opts(float(g["lr"]), momentum=mom)
    xent = O.Xent()
    for s in range(int(g["nsteps"])):
        post = net.propagate(x)
        np.testing.assert_allclose(post[:, :16], g["%s_s%d_post_cols" % (name, s)], rtol=1e-4, atol=1e-12)
        np.testing.assert_allclose(post[np.arange(N), tgt], g["%s_s%d_post_tgt" % (name, s)], rtol=1e-4, atol=1e-12)
        assert (post.argmax(1) == g["%s_s%d_post_argmax" % (name, s)]).all()
        diff, _, _, _ = xent.eval_vec(post, tgt)
        assert abs(xent.loss - float(g["%s_s%d_cumloss" % (name, s)])) <= 1e-4 * abs(xent.loss)
        assert abs(100.0 * xent.correct / xent.frames - float(g["%s_s%d_cumacc" % (name, s)])) < 1e-3
        net.backpropagate(diff)
    for i in range(0, 2 * (len(dims) - 1), 2):
        W = net.layers[i].W
        close(W[:4], g["%s_W%d_rows" % (name, i)], rtol=1e-4, atol=1e-6)
        close(np.array([W[j % W.shape[0], (7 * j) % W.shape[1]] for j in range(512)]), g["%s_W%d_diag" % (name, i)],
              rtol=1e-4, atol=1e-6)
        assert abs(np.abs(W.astype(np.float64)).sum() - float(g["%s_W%d_abs_sum" % (name, i)])) < 1e-4 * float(g["%s_W%d_abs_sum" % (name, i)])
        close(net.layers[i].b, g["%s_b%d" % (name, i)], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name,opts", [("plain", dict(learn_rate=0.008)),
                                       ("mom", dict(learn_rate=0.05, momentum=0.9, average_grad=True))])
def test_relu_softrelu_scale_match_reference(name, opts):
    """The oracle's <relu>, <softrelu> and ,<scale> (nnet/nnet-activation.h:27-174) against the reference's tools on a
    24 -> 32 relu -> 32 softrelu -> scale 0.5 -> 17 softmax network: model bytes in both formats, nnet-forward
    posteriors, and one epoch of the trainer (report, trained weights)."""
    g = load("activations_small.npz")
    layers = O.read_nnet(g["model"].tobytes())
    assert [l.marker for l in layers] == ["<biasedlinearity>", "<relu>", "<biasedlinearity>", "<softrelu>", ",<scale>",
                                          "<biasedlinearity>", "<softmax>"]
    assert O.write_nnet(layers, False) == g["model_txt"].tobytes()
    assert O.write_nnet(O.read_nnet(g["model_txt"].tobytes()), True) == g["model_bin_rewritten"].tobytes()
    net = O.Nnet(layers)
    feats = dict(O.read_matrix_ark(g["feats_ark"].tobytes()))
    some_above_4 = False
    for k, ref in O.read_matrix_ark(g["forward_ark"].tobytes()):
        close(net.feedforward(feats[k]), ref, rtol=2e-5, atol=1e-7)
        net.propagate(feats[k])
        some_above_4 |= bool((net.propagate_buf[3] > 4.0).any()) and bool((net.propagate_buf[3] <= 4.0).any())
    assert some_above_4                                        # both branches of <softrelu> ran
    net, xent, _ = run_trainer(g, opts, False)
    assert xent.frames == int(g[name + "_frames"])
    assert abs(xent.loss - float(g[name + "_xent"])) <= 2e-5 * abs(xent.loss)
    assert xent.report().split("\n")[1] == g[name + "_report"].tobytes().decode().split("\n")[1]
    _check_models(net, g[name + "_model_out"].tobytes())
