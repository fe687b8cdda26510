"""Generates tests/golden/*.npz by running the REFERENCE implementation built under
oracle/_ref (make -C oracle ref) on small seeded inputs.  Run in the build container only
(needs /root/reference via oracle/_ref); the GPU box and the CPU tests read the committed .npz.

    python tests/golden/make_golden.py

Each fixture holds the inputs, the model file bytes as the reference wrote / read them, and the
tensors dumped by oracle/ref_dump.cc or produced by the reference's own command-line tools.
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import nnet1_oracle as O  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "bin")
f32 = np.float32


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    if r.returncode != 0:
        raise RuntimeError("%s failed (%d):\n%s" % (cmd, r.returncode, r.stderr.decode()[-2000:]))
    return r.stdout, r.stderr.decode()


def make_dnn(rs, dims, softmax=True):
    """weights 0.1*N(0,1), hidden biases U(-4.1,-3.9) scaled down for the small net, output bias 0"""
    layers = []
    for i in range(len(dims) - 1):
        W = (0.1 * rs.randn(dims[i + 1], dims[i])).astype(f32)
        last = i == len(dims) - 2
        b = np.zeros(dims[i + 1], f32) if last else (-0.5 + 0.2 * rs.rand(dims[i + 1])).astype(f32)
        layers.append(O.BiasedLinearity(W, b))
        if not last:
            layers.append(O.Sigmoid(dims[i + 1]))
        elif softmax:
            layers.append(O.Softmax(dims[i + 1]))
    return layers


def archive_to_dict(data):
    d = {}
    for k, m in O.read_matrix_ark(data):
        d[k] = m
    return d


def scalar(m):
    return float(np.float64(m[0, 0]) + np.float64(m[0, 1]))


def gen_dnn_steps(tmp):
    rs = np.random.RandomState(1234)
    dims = [24, 32, 32, 17]
    N = 16
    layers = make_dnn(rs, dims)
    x = rs.randn(N, dims[0]).astype(f32)
    tgt = rs.randint(0, dims[-1], N).astype(np.int32)
    model_bin, model_txt = O.write_nnet(layers, True), O.write_nnet(layers, False)
    open(os.path.join(tmp, "m.bin"), "wb").write(model_bin)
    open(os.path.join(tmp, "m.txt"), "wb").write(model_txt)
    open(os.path.join(tmp, "f.ark"), "wb").write(O.write_matrix_ark([("utt1", x)]))
    open(os.path.join(tmp, "a.ark"), "wb").write(O.write_int32_vector_ark([("utt1", tgt)]))
    cfgs = {
        "plain": dict(nsteps=2, lr=0.008, mom=0.0, l2=0.0, l1=0.0, l2ub=0.0, avg=0),
        # no --l2-penalty anywhere: the reference's CPU branch aborts on it (self-AddMat is taken
        # as a transposed add and trips its assert on a non-square matrix, matrix/kaldi-matrix.cc:275-281)
        "mom_avg": dict(nsteps=3, lr=0.5, mom=0.9, l2=0.0, l1=0.0, l2ub=0.0, avg=1),
        "l1": dict(nsteps=2, lr=0.05, mom=0.5, l2=0.0, l1=2e-3, l2ub=0.0, avg=0),
        "l2ub": dict(nsteps=2, lr=0.5, mom=0.0, l2=0.0, l1=0.0, l2ub=0.4, avg=0),
    }
    for name, c in cfgs.items():
        out = {}
        for mfile in ("m.bin", "m.txt"):  # both model encodings must give the same numbers
            oark = os.path.join(tmp, "o.ark")
            run([os.path.join(BIN, "ref_dump"), "dnn", os.path.join(tmp, mfile), os.path.join(tmp, "f.ark"),
                 os.path.join(tmp, "a.ark"), oark, str(c["nsteps"]), repr(c["lr"]), repr(c["mom"]), repr(c["l2"]),
                 repr(c["l1"]), repr(c["l2ub"]), str(c["avg"])])
            d = archive_to_dict(open(oark, "rb").read())
            if mfile == "m.bin":
                out = d
            else:  # text model: printed at 7 significant digits, so only close
                for k in d:
                    np.testing.assert_allclose(d[k], out[k], rtol=2e-4, atol=2e-6)
        fix = {"x": x, "tgt": tgt, "model_bin": np.frombuffer(model_bin, np.uint8),
               "model_txt": np.frombuffer(model_txt, np.uint8)}
        fix.update({"opt_" + k: np.float64(v) for k, v in c.items()})
        for k, v in out.items():
            fix[k] = np.float64(scalar(v)) if k.endswith(("cumloss", "cumframes", "cumacc")) else v
        np.savez_compressed(os.path.join(HERE, "dnn_small_%s.npz" % name), **fix)
        print("dnn_small_%s: %d arrays" % (name, len(fix)))


def gen_forward(tmp):
    rs = np.random.RandomState(4321)
    dims = [24, 32, 17]
    layers = make_dnn(rs, dims)
    lengths = [37, 1, 5, 64]
    utts = [("utt%d" % i, rs.randn(n, dims[0]).astype(f32)) for i, n in enumerate(lengths)]
    counts = (rs.randint(1, 1000, dims[-1])).astype(np.float64)
    model = O.write_nnet(layers, True)
    model_nosm = O.write_nnet(layers[:-1], True)
    open(os.path.join(tmp, "fm.bin"), "wb").write(model)
    open(os.path.join(tmp, "fm_nosm.bin"), "wb").write(model_nosm)
    open(os.path.join(tmp, "ff.ark"), "wb").write(O.write_matrix_ark(utts))
    open(os.path.join(tmp, "counts.txt"), "wb").write(O.write_vector(counts, False))
    fix = {"model": np.frombuffer(model, np.uint8), "model_nosm": np.frombuffer(model_nosm, np.uint8),
           "counts": counts, "counts_txt": np.frombuffer(O.write_vector(counts, False), np.uint8)}
    for k, m in utts:
        fix["in_" + k] = m
    variants = {
        "plain": ("fm.bin", []),
        "log": ("fm.bin", ["--apply-log=true"]),
        "counts": ("fm.bin", ["--class-frame-counts=" + os.path.join(tmp, "counts.txt")]),
        "counts_log_scale": ("fm.bin", ["--class-frame-counts=" + os.path.join(tmp, "counts.txt"), "--apply-log=true",
                                        "--prior-scale=0.7"]),
        "nosoftmax_counts": ("fm_nosm.bin", ["--no-softmax=true",
                                             "--class-frame-counts=" + os.path.join(tmp, "counts.txt")]),
    }
    for name, (mfile, opts) in variants.items():
        oark = os.path.join(tmp, "fo.ark")
        _, err = run([os.path.join(BIN, "nnet-forward")] + opts + [os.path.join(tmp, mfile),
                                                                   "ark:" + os.path.join(tmp, "ff.ark"), "ark:" + oark])
        for k, m in O.read_matrix_ark(open(oark, "rb").read()):
            fix["out_%s_%s" % (name, k)] = m
    # text-mode archive output, byte for byte
    oark = os.path.join(tmp, "fo_t.ark")
    run([os.path.join(BIN, "nnet-forward"), os.path.join(tmp, "fm.bin"), "ark:" + os.path.join(tmp, "ff.ark"),
         "ark,t:" + oark])
    fix["out_plain_text_ark"] = np.frombuffer(open(oark, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "forward_small.npz"), **fix)
    print("forward_small: %d arrays" % len(fix))


def gen_train_e2e(tmp):
    """The reference trainer itself on 9 ragged utterances: cache refills, carried-over rows, the
    dropped last partial bunch, with and without shuffling (the shuffle draws from an unseeded
    rand(), so it is reproducible)."""
    rs = np.random.RandomState(777)
    dims = [24, 32, 32, 17]
    layers = make_dnn(rs, dims)
    lengths = [30, 5, 40, 17, 1, 38, 33, 21, 29]  # each shorter than the 64-row cache
    utts = [("spk%02d" % i, rs.randn(n, dims[0]).astype(f32)) for i, n in enumerate(lengths)]
    alis = [(k, rs.randint(0, dims[-1], m.shape[0]).astype(np.int32)) for k, m in utts]
    model = O.write_nnet(layers, True)
    open(os.path.join(tmp, "tm.bin"), "wb").write(model)
    open(os.path.join(tmp, "tf.ark"), "wb").write(O.write_matrix_ark(utts))
    open(os.path.join(tmp, "ta.ark"), "wb").write(O.write_int32_vector_ark(alis, binary=False))
    fix = {"model": np.frombuffer(model, np.uint8), "lengths": np.array(lengths),
           "feats_ark": np.frombuffer(O.write_matrix_ark(utts), np.uint8),
           "ali_ark_txt": np.frombuffer(O.write_int32_vector_ark(alis, binary=False), np.uint8)}
    for name, opts in {"norand": ["--randomize=false"], "rand": [], "cv": ["--cross-validate=true"],
                       "norand_mom": ["--randomize=false", "--momentum=0.9", "--learn-rate=0.1",
                                      "--average-grad=true"]}.items():
        mout = os.path.join(tmp, "tout.nnet")
        args = [os.path.join(BIN, "nnet-train-xent-hardlab-frmshuff"), "--bunchsize=16", "--cachesize=64",
                "--binary=true"] + opts + ["ark:" + os.path.join(tmp, "tf.ark"), "ark:" + os.path.join(tmp, "ta.ark"),
                                           os.path.join(tmp, "tm.bin")]
        if name != "cv":
            args.append(mout)
        _, err = run(args)
        m = re.search(r"Xent:(\S+) frames:(\d+) err/frm:(\S+)\s*\nFRAME_ACCURACY >> (\S+)% <<", err)
        fix["%s_xent" % name] = np.float64(m.group(1))
        fix["%s_frames" % name] = np.int64(m.group(2))
        fix["%s_acc" % name] = np.float64(m.group(4))
        fix["%s_report" % name] = np.frombuffer(m.group(0).encode(), np.uint8)
        fix["%s_cache_lines" % name] = np.frombuffer("\n".join(re.findall(r"Cache #.*", err)).encode(), np.uint8)
        if name != "cv":
            fix["%s_model_out" % name] = np.frombuffer(open(mout, "rb").read(), np.uint8)
    # text-mode model output of the reference, to pin the text writer byte for byte
    mout = os.path.join(tmp, "tout_txt.nnet")
    run([os.path.join(BIN, "nnet-train-xent-hardlab-frmshuff"), "--bunchsize=16", "--cachesize=64", "--randomize=false",
         "ark:" + os.path.join(tmp, "tf.ark"), "ark:" + os.path.join(tmp, "ta.ark"), os.path.join(tmp, "tm.bin"), mout])
    fix["norand_model_out_txt"] = np.frombuffer(open(mout, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "train_e2e_small.npz"), **fix)
    print("train_e2e_small: %d arrays" % len(fix))


def gen_rbm(tmp):
    for name, (vt, ht) in {"bern_bern": ("bern", "bern"), "gauss_bern": ("gauss", "bern"),
                           "bern_gauss": ("bern", "gauss")}.items():
        rs = np.random.RandomState(99)
        V, H, N, nsteps = 24, 32, 16, 3
        W = (0.1 * rs.randn(H, V)).astype(f32)
        vb = (0.1 * rs.randn(V)).astype(f32)
        hb = (0.1 * rs.randn(H)).astype(f32)
        rbm = O.Rbm(W, vb, hb, vt, ht)
        x = rs.randn(N, V).astype(f32) if vt == "gauss" else rs.rand(N, V).astype(f32)
        us = [rs.rand(N, H).astype(f32) for _ in range(nsteps)]
        gs = [rs.randn(N, H).astype(f32) for _ in range(nsteps)]
        model = O.write_nnet([rbm], True)
        open(os.path.join(tmp, "r.bin"), "wb").write(model)
        open(os.path.join(tmp, "rf.ark"), "wb").write(O.write_matrix_ark([("u", x)]))
        open(os.path.join(tmp, "rr.ark"), "wb").write(
            O.write_matrix_ark([("u%d" % i, u) for i, u in enumerate(us)] + [("g%d" % i, g) for i, g in enumerate(gs)]))
        oark = os.path.join(tmp, "ro.ark")
        out, _ = run([os.path.join(BIN, "ref_dump"), "rbm", os.path.join(tmp, "r.bin"), os.path.join(tmp, "rf.ark"),
                      os.path.join(tmp, "rr.ark"), oark, str(nsteps), "0.05", "0.5", "0.0002"])
        fix = {"x": x, "model": np.frombuffer(model, np.uint8), "as_nnet_txt": np.frombuffer(out, np.uint8),
               "lr": np.float64(0.05), "mom": np.float64(0.5), "l2": np.float64(0.0002)}
        for i in range(nsteps):
            fix["u%d" % i], fix["g%d" % i] = us[i], gs[i]
        fix.update(archive_to_dict(open(oark, "rb").read()))
        np.savez_compressed(os.path.join(HERE, "rbm_small_%s.npz" % name), **fix)
        print("rbm_small_%s: %d arrays" % (name, len(fix)))
    # the reference tool end to end (gaussian hidden units work on its CPU path; bernoulli hidden
    # units abort there: BinarizeProbs never sizes its state matrix, cu-rand-inl.h:162-168): shuffle + sampling
    # both draw from rand()
    rs = np.random.RandomState(100)
    V, H = 24, 32
    rbm = O.Rbm((0.1 * rs.randn(H, V)).astype(f32), np.zeros(V, f32), np.zeros(H, f32), "bern", "gauss")
    utts = [("u%d" % i, rs.rand(n, V).astype(f32)) for i, n in enumerate([40, 9, 70])]
    open(os.path.join(tmp, "rg.bin"), "wb").write(O.write_nnet([rbm], True))
    open(os.path.join(tmp, "rgf.ark"), "wb").write(O.write_matrix_ark(utts))
    _, err = run([os.path.join(BIN, "rbm-train-cd1-frmshuff"), "--bunchsize=16", "--cachesize=64", "--binary=true",
                  os.path.join(tmp, "rg.bin"), "ark:" + os.path.join(tmp, "rgf.ark"), os.path.join(tmp, "rg.out")])
    m = re.search(r"Mse:(\S+) frames:(\d+)", err)
    np.savez_compressed(os.path.join(HERE, "rbm_e2e_gauss_hidden.npz"), model=np.frombuffer(O.write_nnet([rbm], True), np.uint8),
                        feats_ark=np.frombuffer(O.write_matrix_ark(utts), np.uint8),
                        model_out=np.frombuffer(open(os.path.join(tmp, "rg.out"), "rb").read(), np.uint8),
                        stderr=np.frombuffer(err.encode(), np.uint8))
    print("rbm_e2e_gauss_hidden: Mse report:", m.group(0) if m else None)


def gen_rm_shape(tmp):
    """One training step at the RM size (440 -> 4x1024 -> 1483, bunch 256): only summaries are
    stored; the test regenerates the same model and data from the seed."""
    rs = np.random.RandomState(2024)
    dims = [440, 1024, 1024, 1024, 1024, 1483]
    N = 256
    layers = []
    for i in range(len(dims) - 1):
        W = (0.1 * rs.randn(dims[i + 1], dims[i])).astype(f32)
        last = i == len(dims) - 2
        b = np.zeros(dims[i + 1], f32) if last else (-4.1 + 0.2 * rs.rand(dims[i + 1])).astype(f32)
        layers.append(O.BiasedLinearity(W, b))
        layers.append(O.Softmax(dims[i + 1]) if last else O.Sigmoid(dims[i + 1]))
    x = rs.randn(N, dims[0]).astype(f32)
    tgt = rs.randint(0, dims[-1], N).astype(np.int32)
    open(os.path.join(tmp, "rm.bin"), "wb").write(O.write_nnet(layers, True))
    open(os.path.join(tmp, "rmf.ark"), "wb").write(O.write_matrix_ark([("u", x)]))
    open(os.path.join(tmp, "rma.ark"), "wb").write(O.write_int32_vector_ark([("u", tgt)]))
    oark = os.path.join(tmp, "rmo.ark")
    run([os.path.join(BIN, "ref_dump"), "dnn", os.path.join(tmp, "rm.bin"), os.path.join(tmp, "rmf.ark"),
         os.path.join(tmp, "rma.ark"), oark, "1", "0.008", "0.0", "0.0", "0.0", "0.0", "0"])
    d = archive_to_dict(open(oark, "rb").read())
    fix = {"seed": np.int64(2024), "dims": np.array(dims), "N": np.int64(N),
           "post_cols": d["s0_post"][:, :16].copy(), "post_argmax": d["s0_post"].argmax(1).astype(np.int32),
           "post_tgt": d["s0_post"][np.arange(N), tgt].copy(),
           "cumloss": np.float64(scalar(d["s0_cumloss"])), "cumacc": np.float64(scalar(d["s0_cumacc"]))}
    for i in range(0, 2 * (len(dims) - 1), 2):
        G = d["s0_Wcorr%d" % i]                                # momentum 0: the minibatch gradient
        fix["grad%d_rows" % i] = G[:4].copy()
        fix["grad%d_diag" % i] = np.array([G[j % G.shape[0], (7 * j) % G.shape[1]] for j in range(512)])
        fix["grad%d_abs_sum" % i] = np.float64(np.abs(G.astype(np.float64)).sum())
        fix["bgrad%d" % i] = d["s0_bcorr%d" % i][0].copy()
        fix["W%d_rows" % i] = d["s0_W%d" % i][:4].copy()       # updated weights
        fix["b%d" % i] = d["s0_b%d" % i][0].copy()
        fix["hid%d_cols" % i] = d["s0_prop%d" % (i + 2)][:, :8].copy() if i + 2 <= 2 * (len(dims) - 1) else 0
    np.savez_compressed(os.path.join(HERE, "rm_shape_step.npz"), **fix)
    print("rm_shape_step: %d arrays" % len(fix))


def gen_siblings(tmp):
    """The sibling tools that share the trainer's kernels: per-utterance cross-entropy training,
    frame-shuffled MSE training on target matrices (shuffled and cross-validation runs), nnet-copy in both
    directions, rbm-convert-to-nnet."""
    rs = np.random.RandomState(4242)
    dims = [24, 32, 32, 17]
    layers = make_dnn(rs, dims)
    lengths = [30, 5, 40, 17, 1, 38, 33, 21, 29]
    utts = [("spk%02d" % i, rs.randn(n, dims[0]).astype(f32)) for i, n in enumerate(lengths)]
    alis = [(k, rs.randint(0, dims[-1], m.shape[0]).astype(np.int32)) for k, m in utts]
    model = O.write_nnet(layers, True)
    P = lambda n: os.path.join(tmp, n)
    open(P("sm.bin"), "wb").write(model)
    open(P("sf.ark"), "wb").write(O.write_matrix_ark(utts))
    open(P("sa.ark"), "wb").write(O.write_int32_vector_ark(alis, binary=False))
    fix = {"model": np.frombuffer(model, np.uint8), "feats_ark": np.frombuffer(O.write_matrix_ark(utts), np.uint8),
           "ali_ark_txt": np.frombuffer(O.write_int32_vector_ark(alis, binary=False), np.uint8)}
    # per-utterance trainer
    for name, opts in {"perutt": [], "perutt_mom": ["--momentum=0.5", "--learn-rate=0.02"]}.items():
        _, err = run([os.path.join(BIN, "nnet-train-xent-hardlab-perutt"), "--binary=true"] + opts +
                     ["ark:" + P("sf.ark"), "ark:" + P("sa.ark"), P("sm.bin"), P("so.nnet")])
        m = re.search(r"Xent:(\S+) frames:(\d+) err/frm:(\S+)\s*\nFRAME_ACCURACY >> (\S+)% <<", err)
        fix[name + "_xent"], fix[name + "_frames"] = np.float64(m.group(1)), np.int64(m.group(2))
        fix[name + "_report"] = np.frombuffer(m.group(0).encode(), np.uint8)
        fix[name + "_model_out"] = np.frombuffer(open(P("so.nnet"), "rb").read(), np.uint8)
    # MSE trainer: 24 -> 32 sigmoid -> 24 linear, targets = noisy copies of the input
    ae = make_dnn(rs, [24, 32, 24], softmax=False)
    ae_model = O.write_nnet(ae, True)
    tgts = [(k, (m + 0.1 * rs.randn(*m.shape)).astype(f32)) for k, m in utts]
    open(P("am.bin"), "wb").write(ae_model)
    open(P("at.ark"), "wb").write(O.write_matrix_ark(tgts))
    fix["ae_model"] = np.frombuffer(ae_model, np.uint8)
    fix["tgt_ark"] = np.frombuffer(O.write_matrix_ark(tgts), np.uint8)
    for name, opts in {"mse": [], "mse_cv": ["--cross-validate=true"],
                       "mse_mom": ["--momentum=0.9", "--learn-rate=0.05", "--average-grad=true"]}.items():
        args = [os.path.join(BIN, "nnet-train-mse-tgtmat-frmshuff"), "--bunchsize=16", "--cachesize=64", "--binary=true"] + opts + [
            "ark:" + P("sf.ark"), "ark:" + P("at.ark"), P("am.bin")]
        if name != "mse_cv":
            args.append(P("ao.nnet"))
        _, err = run(args)
        m = re.search(r"Mse:(\S+) frames:(\d+) err/frm:(\S+)", err)
        fix[name + "_loss"], fix[name + "_frames"] = np.float64(m.group(1)), np.int64(m.group(2))
        fix[name + "_cache_lines"] = np.frombuffer("\n".join(re.findall(r"Cache #.*", err)).encode(), np.uint8)
        if name != "mse_cv":
            fix[name + "_model_out"] = np.frombuffer(open(P("ao.nnet"), "rb").read(), np.uint8)
    # nnet-copy both ways
    run([os.path.join(BIN, "nnet-copy"), "--binary=false", P("sm.bin"), P("sm.txt")])
    run([os.path.join(BIN, "nnet-copy"), "--binary=true", P("sm.txt"), P("sm2.bin")])
    fix["copy_txt"] = np.frombuffer(open(P("sm.txt"), "rb").read(), np.uint8)
    fix["copy_bin"] = np.frombuffer(open(P("sm2.bin"), "rb").read(), np.uint8)
    # rbm-convert-to-nnet for both hidden unit types
    for name, (vt, ht) in {"bb": ("bern", "bern"), "gb": ("gauss", "bern"), "bg": ("bern", "gauss")}.items():
        rbm = O.Rbm((0.1 * rs.randn(32, 24)).astype(f32), (0.1 * rs.randn(24)).astype(f32), (0.1 * rs.randn(32)).astype(f32), vt, ht)
        open(P("r.rbm"), "wb").write(O.write_nnet([rbm], True))
        run([os.path.join(BIN, "rbm-convert-to-nnet"), "--binary=true", P("r.rbm"), P("r.nnet")])
        run([os.path.join(BIN, "rbm-convert-to-nnet"), P("r.rbm"), P("r.txt")])
        fix["rbm_" + name] = np.frombuffer(O.write_nnet([rbm], True), np.uint8)
        fix["rbm_%s_nnet_bin" % name] = np.frombuffer(open(P("r.nnet"), "rb").read(), np.uint8)
        fix["rbm_%s_nnet_txt" % name] = np.frombuffer(open(P("r.txt"), "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "siblings_small.npz"), **fix)
    print("siblings_small: %d arrays" % len(fix))


def gen_transform(tmp):
    """--feature-transform in the three tools (nnetbin/nnet-forward.cc:66-75,127-131,
    nnet-train-xent-hardlab-frmshuff.cc:148, rbm-train-cd1-frmshuff.cc:117): a transform network
    <expand> (+-5 frames) -> <copy> (a permuted subset of the spliced columns, 1-based on disk) ->
    <biasedlinearity> <sigmoid>, as the stacked-RBM recipe builds it (egs/aurora4/a00/steps/pretrain_dnn.sh:120-133),
    written by the reference's nnet-copy in text and in binary."""
    rs = np.random.RandomState(9091)
    D, ctx = 6, 5
    offsets = list(range(-ctx, ctx + 1))
    spliced = D * len(offsets)                       # 66
    pick = rs.permutation(spliced)[:40]              # 0-based column indices kept by <copy>
    bl = O.BiasedLinearity((0.3 * rs.randn(24, 40)).astype(f32), (-0.5 + 0.2 * rs.rand(24)).astype(f32))
    transf = [O.Expand(D, spliced, offsets), O.Copy(spliced, 40, pick), bl, O.Sigmoid(24)]
    net = make_dnn(rs, [24, 32, 17])
    lengths = [30, 3, 41, 1, 12, 25, 37]             # utterances shorter than the context included
    utts = [("fx%02d" % i, rs.randn(n, D).astype(f32)) for i, n in enumerate(lengths)]
    alis = [(k, rs.randint(0, 17, m.shape[0]).astype(np.int32)) for k, m in utts]
    P = lambda n: os.path.join(tmp, n)
    open(P("t.txt"), "wb").write(O.write_nnet(transf, False))
    open(P("n.bin"), "wb").write(O.write_nnet(net, True))
    open(P("xf.ark"), "wb").write(O.write_matrix_ark(utts))
    open(P("xa.ark"), "wb").write(O.write_int32_vector_ark(alis, binary=False))
    fix = {"transf_txt_in": np.frombuffer(O.write_nnet(transf, False), np.uint8),
           "model": np.frombuffer(O.write_nnet(net, True), np.uint8),
           "feats_ark": np.frombuffer(O.write_matrix_ark(utts), np.uint8),
           "ali_ark_txt": np.frombuffer(O.write_int32_vector_ark(alis, binary=False), np.uint8),
           "offsets": np.array(offsets), "pick0": np.array(pick)}
    # the transform as the reference re-writes it, both formats, both directions
    run([os.path.join(BIN, "nnet-copy"), "--binary=true", P("t.txt"), P("t.bin")])
    run([os.path.join(BIN, "nnet-copy"), "--binary=false", P("t.bin"), P("t2.txt")])
    fix["transf_bin"] = np.frombuffer(open(P("t.bin"), "rb").read(), np.uint8)
    fix["transf_txt"] = np.frombuffer(open(P("t2.txt"), "rb").read(), np.uint8)
    # the transform alone and transform + network through nnet-forward
    _, err = run([os.path.join(BIN, "nnet-forward"), P("t.bin"), "ark:" + P("xf.ark"), "ark:" + P("xt.ark")])
    for k, m in O.read_matrix_ark(open(P("xt.ark"), "rb").read()):
        fix["transformed_" + k] = m
    for name, tf in (("bin", "t.bin"), ("txt", "t2.txt")):
        run([os.path.join(BIN, "nnet-forward"), "--feature-transform=" + P(tf), P("n.bin"), "ark:" + P("xf.ark"), "ark:" + P("xo.ark")])
        for k, m in O.read_matrix_ark(open(P("xo.ark"), "rb").read()):
            fix["post_%s_%s" % (name, k)] = m
    # training with the transform in front
    for name, opts in {"train": ["--randomize=false"], "train_rand": [], "cv": ["--cross-validate=true"]}.items():
        args = [os.path.join(BIN, "nnet-train-xent-hardlab-frmshuff"), "--bunchsize=16", "--cachesize=64", "--binary=true",
                "--feature-transform=" + P("t.bin")] + opts + ["ark:" + P("xf.ark"), "ark:" + P("xa.ark"), P("n.bin")]
        if name != "cv":
            args.append(P("no.nnet"))
        _, err = run(args)
        m = re.search(r"Xent:(\S+) frames:(\d+) err/frm:(\S+)\s*\nFRAME_ACCURACY >> (\S+)% <<", err)
        fix[name + "_xent"], fix[name + "_frames"], fix[name + "_acc"] = np.float64(m.group(1)), np.int64(m.group(2)), np.float64(m.group(4))
        fix[name + "_report"] = np.frombuffer(m.group(0).encode(), np.uint8)
        if name != "cv":
            fix[name + "_model_out"] = np.frombuffer(open(P("no.nnet"), "rb").read(), np.uint8)
    # RBM pre-training on the transformed features (Bernoulli visible = the transform's sigmoid outputs, Gaussian
    # hidden: the only hidden type the reference's CPU path can run).  Its noise comes from rand(), this
    # repository's from the device generator, so only the frame count of its report can be pinned.
    rbm = O.Rbm((0.1 * rs.randn(20, 24)).astype(f32), np.zeros(24, f32), np.zeros(20, f32), "bern", "gauss")
    open(P("r.rbm"), "wb").write(O.write_nnet([rbm], True))
    fix["rbm"] = np.frombuffer(O.write_nnet([rbm], True), np.uint8)
    _, err = run([os.path.join(BIN, "rbm-train-cd1-frmshuff"), "--bunchsize=16", "--cachesize=64", "--binary=true",
                  "--learn-rate=0.001", "--feature-transform=" + P("t.bin"), P("r.rbm"), "ark:" + P("xf.ark"), P("ro.rbm")])
    m = re.search(r"Mse:(\S+) frames:(\d+) err/frm:(\S+)", err)
    fix["rbm_frames"], fix["rbm_mse_per_frame_reference"] = np.int64(m.group(2)), np.float64(m.group(3))
    np.savez_compressed(os.path.join(HERE, "transform_small.npz"), **fix)
    print("transform_small: %d arrays" % len(fix))


def gen_activations(tmp):
    """The components the factory builds beside <sigmoid> / <softmax> (nnet/nnet-activation.h:27-174) through the
    reference's own tools on its CPU path: a 24 -> 32 <relu> -> 32 <softrelu> -> ,<scale> 0.5 -> 17 <softmax> network
    in nnet-forward (posteriors, and the text / binary model as nnet-copy re-writes it) and one epoch of
    nnet-train-xent-hardlab-frmshuff (report and trained model).  Four units of the second layer get a bias of 4.5,
    so both branches of <softrelu> (x > 4) run.  <dropout> cannot run on the reference's CPU path (its random
    generator's CPU branch never sizes its state) and has no fixture."""
    rs = np.random.RandomState(4242)
    dims = [24, 32, 32, 17]
    W = [(0.1 * rs.randn(dims[i + 1], dims[i])).astype(f32) for i in range(3)]
    b = [(-0.5 + 0.2 * rs.rand(dims[1])).astype(f32), (-0.5 + 0.2 * rs.rand(dims[2])).astype(f32), np.zeros(dims[3], f32)]
    b[1][:4] = 4.5
    layers = [O.BiasedLinearity(W[0], b[0]), O.Relu(dims[1]), O.BiasedLinearity(W[1], b[1]), O.SoftRelu(dims[2]),
              O.Scale(dims[2], 0.5), O.BiasedLinearity(W[2], b[2]), O.Softmax(dims[3])]
    lengths = [30, 5, 40, 17, 1, 38, 33, 21, 29]
    utts = [("spk%02d" % i, rs.randn(n, dims[0]).astype(f32)) for i, n in enumerate(lengths)]
    alis = [(k, rs.randint(0, dims[-1], m.shape[0]).astype(np.int32)) for k, m in utts]
    model = O.write_nnet(layers, True)
    P = lambda n: os.path.join(tmp, n)
    open(P("am.bin"), "wb").write(model)
    open(P("af.ark"), "wb").write(O.write_matrix_ark(utts))
    open(P("aa.ark"), "wb").write(O.write_int32_vector_ark(alis, binary=False))
    fix = {"model": np.frombuffer(model, np.uint8), "feats_ark": np.frombuffer(O.write_matrix_ark(utts), np.uint8),
           "ali_ark_txt": np.frombuffer(O.write_int32_vector_ark(alis, binary=False), np.uint8)}
    run([os.path.join(BIN, "nnet-copy"), "--binary=false", P("am.bin"), P("am.txt")])
    run([os.path.join(BIN, "nnet-copy"), "--binary=true", P("am.txt"), P("am2.bin")])
    fix["model_txt"] = np.frombuffer(open(P("am.txt"), "rb").read(), np.uint8)
    fix["model_bin_rewritten"] = np.frombuffer(open(P("am2.bin"), "rb").read(), np.uint8)
    run([os.path.join(BIN, "nnet-forward"), P("am.bin"), "ark:" + P("af.ark"), "ark:" + P("ao.ark")])
    fix["forward_ark"] = np.frombuffer(open(P("ao.ark"), "rb").read(), np.uint8)
    for name, opts in {"plain": ["--randomize=false"], "mom": ["--randomize=false", "--momentum=0.9", "--learn-rate=0.05", "--average-grad=true"]}.items():
        _, err = run([os.path.join(BIN, "nnet-train-xent-hardlab-frmshuff"), "--bunchsize=16", "--cachesize=64", "--binary=true"] + opts +
                     ["ark:" + P("af.ark"), "ark:" + P("aa.ark"), P("am.bin"), P("aout.nnet")])
        m = re.search(r"Xent:(\S+) frames:(\d+) err/frm:(\S+)\s*\nFRAME_ACCURACY >> (\S+)% <<", err)
        fix["%s_xent" % name] = np.float64(m.group(1))
        fix["%s_frames" % name] = np.int64(m.group(2))
        fix["%s_report" % name] = np.frombuffer(m.group(0).encode(), np.uint8)
        fix["%s_model_out" % name] = np.frombuffer(open(P("aout.nnet"), "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "activations_small.npz"), **fix)
    print("activations_small: %d arrays" % len(fix))


def rm_shape_net(seed):
    """the RM-sized model, bunch and labels of gen_rm_shape / gen_rm_multistep, regenerated from the seed by the tests"""
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
    x = rs.randn(N, dims[0]).astype(f32)
    tgt = rs.randint(0, dims[-1], N).astype(np.int32)
    return dims, N, layers, x, tgt


def gen_rm_multistep(tmp):
    """Three consecutive training steps at the RM size (bunch 256), momentum 0 and 0.9 (a learning rate large enough
    for the steps to differ visibly): per-step summaries of the posteriors and the running loss / accuracy, and
    samples of every layer's weights and biases after the last step.  Pins the oracle at this size over more than one
    step: the next forward pass must see the weights the previous step wrote."""
    dims, N, layers, x, tgt = rm_shape_net(2025)
    open(os.path.join(tmp, "rmm.bin"), "wb").write(O.write_nnet(layers, True))
    open(os.path.join(tmp, "rmmf.ark"), "wb").write(O.write_matrix_ark([("u", x)]))
    open(os.path.join(tmp, "rmma.ark"), "wb").write(O.write_int32_vector_ark([("u", tgt)]))
    fix = {"seed": np.int64(2025), "dims": np.array(dims), "N": np.int64(N), "nsteps": np.int64(3), "lr": np.float64(0.05)}
    for name, mom in (("mom0", "0.0"), ("mom9", "0.9")):
        oark = os.path.join(tmp, "rmmo.ark")
        run([os.path.join(BIN, "ref_dump"), "dnn", os.path.join(tmp, "rmm.bin"), os.path.join(tmp, "rmmf.ark"),
             os.path.join(tmp, "rmma.ark"), oark, "3", "0.05", mom, "0.0", "0.0", "0.0", "0"])
        d = archive_to_dict(open(oark, "rb").read())
        for s in range(3):
            post = d["s%d_post" % s]
            fix["%s_s%d_post_cols" % (name, s)] = post[:, :16].copy()
            fix["%s_s%d_post_tgt" % (name, s)] = post[np.arange(N), tgt].copy()
            fix["%s_s%d_post_argmax" % (name, s)] = post.argmax(1).astype(np.int32)
            fix["%s_s%d_cumloss" % (name, s)] = np.float64(scalar(d["s%d_cumloss" % s]))
            fix["%s_s%d_cumacc" % (name, s)] = np.float64(scalar(d["s%d_cumacc" % s]))
        for i in range(0, 2 * (len(dims) - 1), 2):
            W = d["s2_W%d" % i]
            fix["%s_W%d_rows" % (name, i)] = W[:4].copy()
            fix["%s_W%d_diag" % (name, i)] = np.array([W[j % W.shape[0], (7 * j) % W.shape[1]] for j in range(512)])
            fix["%s_W%d_abs_sum" % (name, i)] = np.float64(np.abs(W.astype(np.float64)).sum())
            fix["%s_b%d" % (name, i)] = d["s2_b%d" % i][0].copy()
    np.savez_compressed(os.path.join(HERE, "rm_shape_multistep.npz"), **fix)
    print("rm_shape_multistep: %d arrays" % len(fix))


GENERATORS = {"rm_multistep": gen_rm_multistep, "dnn_steps": gen_dnn_steps, "forward": gen_forward, "train_e2e": gen_train_e2e, "rbm": gen_rbm,
              "rm_shape": gen_rm_shape, "siblings": gen_siblings, "transform": gen_transform, "activations": gen_activations}

if __name__ == "__main__":
    if not os.path.exists(os.path.join(BIN, "ref_dump")):
        sys.exit("build the reference first: make -C oracle ref")
    which = sys.argv[1:] or list(GENERATORS)      # python tests/golden/make_golden.py [name ...]
    with tempfile.TemporaryDirectory() as tmp:
        for name in which:
            GENERATORS[name](tmp)
