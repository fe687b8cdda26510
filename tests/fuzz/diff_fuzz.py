#!/usr/bin/env python3
"""Differential test of the tools against the reference's own tools (oracle/_ref/bin, built from the reference's
sources for the CPU): random small networks, feature archives and option sets; nnet-forward archives and
nnet-train-xent-hardlab-frmshuff models / reports of the two must agree.  Runs where a GPU and the reference
binaries are (the GPU box); a diagnostic beside the fixture tests, which pin chosen cases.

    python tests/fuzz/diff_fuzz.py [--cases 40] [--seed 1] [--siblings] [--exotic] [--keep DIR]
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32
REF = os.path.join(ROOT, "oracle", "_ref", "bin")
OURS = os.path.join(ROOT, "kaldi_amd", "bin")


def run(exe, args, env=None):
    r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=100, env=env)
    return r.returncode, r.stderr.decode(errors="replace")


def make_case(rs, tmp, exotic=False):
    in_dim = int(rs.choice([13, 39, 40, 120, 440]))
    splice = rs.rand() < 0.35 and in_dim <= 40
    hidden = [int(rs.choice([17, 32, 100, 256, 333, 1024])) for _ in range(rs.randint(0, 4))]
    out_dim = int(rs.choice([9, 50, 201, 1483]))
    # feature transform: a splice, then perhaps a column selection, then perhaps an affine map (as the recipes' global
    # normalisation / LDA transforms are)
    tlayers, net_in = [], in_dim
    if splice:
        ctx = int(rs.choice([1, 2, 5]))
        tlayers.append(O.Expand(in_dim, in_dim * (2 * ctx + 1), list(range(-ctx, ctx + 1))))
        net_in = in_dim * (2 * ctx + 1)
        if exotic and rs.rand() < 0.5:
            keep = rs.permutation(net_in)[:max(3, net_in // 2)]
            tlayers.append(O.Copy(net_in, len(keep), [int(k) for k in keep]))
            net_in = len(keep)
        if exotic and rs.rand() < 0.4:
            out = int(rs.choice([net_in, 24, 60]))
            tlayers.append(O.BiasedLinearity((rs.standard_normal((out, net_in)) / np.sqrt(net_in)).astype(f32),
                                             (0.1 * rs.standard_normal(out)).astype(f32)))
            net_in = out
    dims = [net_in] + hidden + [out_dim]
    layers = []
    for i in range(len(dims) - 1):
        layers.append(O.BiasedLinearity((rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(f32),
                                        (0.1 * rs.standard_normal(dims[i + 1])).astype(f32)))
        if i + 2 < len(dims):
            # mostly the sigmoid; now and then the other activations the component factory builds
            kind = rs.choice(["sigmoid", "sigmoid", "sigmoid", "relu", "softrelu", "scaled"]) if exotic else "sigmoid"
            if kind == "relu":
                layers.append(O.Relu(dims[i + 1]))
            elif kind == "softrelu":
                layers.append(O.SoftRelu(dims[i + 1]))
            else:
                layers.append(O.Sigmoid(dims[i + 1]))
                if kind == "scaled":
                    layers.append(O.Scale(dims[i + 1], float(rs.choice([0.5, 2.0]))))
        else:
            layers.append(O.Softmax(dims[i + 1]))
    binary = bool(rs.rand() < 0.5)
    open(os.path.join(tmp, "m.nnet"), "wb").write(O.write_nnet(layers, binary))
    open(os.path.join(tmp, "m_nosm.nnet"), "wb").write(O.write_nnet(layers[:-1], binary))
    exotic_model = any(not isinstance(l, (O.BiasedLinearity, O.Sigmoid, O.Softmax)) for l in layers)
    transform = None
    if splice:
        transform = os.path.join(tmp, "t.nnet")
        open(transform, "wb").write(O.write_nnet(tlayers, binary))
    n_utts = rs.randint(1, 10)
    lens = [int(rs.choice([1, 2, 7, 64, 130, 300, 700])) for _ in range(n_utts)]
    utts = [("utt%03d" % i, rs.standard_normal((n, in_dim)).astype(f32)) for i, n in enumerate(lens)]
    alis = [(k, rs.randint(0, out_dim, m.shape[0]).astype(np.int32)) for k, m in utts]
    if exotic and len(alis) > 2:
        if rs.rand() < 0.3:
            del alis[rs.randint(0, len(alis))]                       # an utterance without an alignment
        # (An alignment of the wrong length is not drawn: the reference's loop meets it with `continue` in front of
        # its feature_reader.Next(), nnet-train-xent-hardlab-frmshuff.cc:139-144, and never ends.  This build counts
        # the utterance under "other errors" and goes on.)
    open(os.path.join(tmp, "f.ark"), "wb").write(O.write_matrix_ark(utts))
    open(os.path.join(tmp, "a.ark"), "wb").write(O.write_int32_vector_ark(alis, binary=not (exotic and rs.rand() < 0.5)))
    counts = rs.randint(1, 1000, out_dim).astype(np.float64)
    open(os.path.join(tmp, "counts.txt"), "wb").write(O.write_vector(counts, False))
    return dict(dims=dims, transform=transform, lens=lens, binary=binary, exotic_model=exotic_model)


def forward_case(rs, tmp, c):
    opts, model = [], "m.nnet"
    if rs.rand() < 0.3:
        opts.append("--apply-log=true")
    if rs.rand() < 0.4:
        opts.append("--class-frame-counts=" + os.path.join(tmp, "counts.txt"))
        if rs.rand() < 0.5:
            opts.append("--prior-scale=%.2f" % rs.uniform(0.2, 1.5))
        if rs.rand() < 0.4 and "--apply-log=true" not in opts:
            opts.append("--no-softmax=true")
            model = "m_nosm.nnet"
    if c["transform"]:
        opts.append("--feature-transform=" + c["transform"])
    ours_opts = list(opts)
    if rs.rand() < 0.5:
        ours_opts.append("--batch-frames=%d" % int(rs.choice([0, 100, 1024, 4096])))
    tail = [os.path.join(tmp, model), "ark:" + os.path.join(tmp, "f.ark")]
    rc_r, err_r = run(os.path.join(REF, "nnet-forward"), opts + tail + ["ark:" + os.path.join(tmp, "o_ref.ark")])
    rc_o, err_o = run(os.path.join(OURS, "nnet-forward"), ours_opts + tail + ["ark:" + os.path.join(tmp, "o_ours.ark")])
    if rc_r != rc_o:
        return "exit codes differ: reference %d, ours %d\n%s\n%s" % (rc_r, rc_o, err_r[-800:], err_o[-800:])
    if rc_r != 0:
        return None
    a = O.read_matrix_ark(open(os.path.join(tmp, "o_ref.ark"), "rb").read())
    b = O.read_matrix_ark(open(os.path.join(tmp, "o_ours.ark"), "rb").read())
    if [k for k, _ in a] != [k for k, _ in b]:
        return "keys differ"
    logs = any(o.startswith(("--apply-log", "--class-frame", "--no-softmax")) for o in opts)
    for (k, x), (_, y) in zip(a, b):
        if x.shape != y.shape:
            return "shape of %s differs: %s %s" % (k, x.shape, y.shape)
        ok = np.allclose(x, y, rtol=2e-4, atol=2e-4 if logs else 1e-7)
        if not ok:
            return "values of %s differ: max abs %.3g (options %s)" % (k, np.abs(x - y).max(), ours_opts)
    return None


def train_case(rs, tmp, c):
    bunch = int(rs.choice([1, 7, 64, 256]))
    cache = bunch * int(rs.choice([1, 2, 5, 16]))
    opts = ["--bunchsize=%d" % bunch, "--cachesize=%d" % cache, "--learn-rate=%g" % rs.choice([0.002, 0.01, 0.05]),
            "--binary=%s" % ("true" if rs.rand() < 0.5 else "false")]
    if rs.rand() < 0.4:
        opts.append("--momentum=%g" % rs.choice([0.5, 0.9]))
    if rs.rand() < 0.3:
        opts.append("--average-grad=true")
    if rs.rand() < 0.25:
        opts.append("--l1-penalty=%g" % rs.choice([1e-5, 1e-4]))
    if rs.rand() < 0.25:
        opts.append("--l2-upper-bound=%g" % rs.choice([1.0, 2.5]))
    if rs.rand() < 0.3:
        opts.append("--randomize=false")
    if rs.rand() < 0.2:
        opts.append("--cross-validate=true")
    if rs.rand() < 0.25:
        opts.append("--learn-factors=" + ",".join("%g" % rs.choice([0.0, 0.5, 1.0, 2.0]) for _ in range(len(c["dims"]) - 1)))
    if c["transform"]:
        opts.append("--feature-transform=" + c["transform"])
    cv = "--cross-validate=true" in opts
    tail = ["ark:" + os.path.join(tmp, "f.ark"), "ark:" + os.path.join(tmp, "a.ark"), os.path.join(tmp, "m.nnet")]
    outs = {}
    for who, d in (("ref", REF), ("ours", OURS)):
        out = os.path.join(tmp, "out_%s.nnet" % who)
        rc, err = run(os.path.join(d, "nnet-train-xent-hardlab-frmshuff"), opts + tail + ([] if cv else [out]))
        outs[who] = (rc, err, out)
    if outs["ref"][0] != outs["ours"][0]:
        return "exit codes differ: reference %d, ours %d (options %s)\n%s\n%s" % (
            outs["ref"][0], outs["ours"][0], opts, outs["ref"][1][-800:], outs["ours"][1][-800:])
    if outs["ref"][0] != 0:
        return None
    rep = {}
    for who in outs:
        m = re.search(r"Xent:(\S+) frames:(\d+) err/frm:(\S+)\nFRAME_ACCURACY >> (\S+)% <<", outs[who][1])
        if not m:
            return "no report from %s" % who
        rep[who] = (float(m.group(1)), int(m.group(2)), float(m.group(4)))
    (xr, fr, ar), (xo, fo, ao) = rep["ref"], rep["ours"]
    if fr != fo:
        return "frame counts differ: %d %d (options %s)" % (fr, fo, opts)
    if fr and (_loss_differs(xr, xo) or abs(ar - ao) > 100.0 * 1.5 / fr + 1e-4):
        return "reports differ: reference %s, ours %s (options %s)" % (rep["ref"], rep["ours"], opts)
    dr = re.findall(r"Done \d+ files, \d+ with no alignments, \d+ with other errors", outs["ref"][1])
    do = re.findall(r"Done \d+ files, \d+ with no alignments, \d+ with other errors", outs["ours"][1])
    if dr != do:
        return "file counts differ (options %s): %s %s" % (opts, dr, do)
    cr = [l for l in outs["ref"][1].splitlines() if l.startswith("Cache #")]
    co = [l for l in outs["ours"][1].splitlines() if l.startswith("Cache #")]
    if cr != co:
        return "cache lines differ (options %s):\n%s\n%s" % (opts, cr, co)
    if cv:
        return None
    return _compare_models(outs["ref"][2], outs["ours"][2], opts)


def _loss_differs(ref, ours, one_sided=True):
    """Summed cross-entropy of the two tools.  The reference's CUDA kernel floors the target's posterior at 1e-20
    before the logarithm (cu-kernels.cu:733-745) and so does this build; the reference's CPU path, which these
    binaries run, takes the logarithm as it is (cu-math-inl.h:373-377).  Where a net has blown up and posteriors
    fall under the floor, the CPU figure is therefore the larger one: that direction is not a difference."""
    if (np.isnan(ref) and np.isnan(ours)) or abs(ref - ours) <= 2e-3 * max(abs(ref), 1.0):
        return False
    return not (one_sided and ours < ref)


def _compare_models(ref_path, ours_path, opts):
    la = [l for l in O.read_nnet(open(ref_path, "rb").read()) if isinstance(l, O.BiasedLinearity)]
    lb = [l for l in O.read_nnet(open(ours_path, "rb").read()) if isinstance(l, O.BiasedLinearity)]
    for i, (x, y) in enumerate(zip(la, lb)):
        for name, u, v in (("W", x.W, y.W), ("b", x.b, y.b)):
            if not (np.isfinite(u).all() and np.abs(u).max() < 20.0):
                # the random step sizes made this training blow up in the reference (initial weights are under 1):
                # rounding differences are amplified without bound, nothing to compare
                return None
            l1 = any(o.startswith("--l1-penalty") for o in opts)
            # The L1 step sets a weight to zero when its sign would change (cu-math-inl.h RegularizeL1): a weight
            # within rounding of that can end on either side, and everything trained after it differs a little.
            # With an L1 penalty a few isolated elements may therefore be off, and the rest is held to 1 %.
            close = np.isclose(u, v, rtol=1e-2 if l1 else 2e-3, atol=(1e-3 if l1 else 1e-4) * max(1.0, float(np.abs(u).max())))
            if (~close).sum() > (max(2, u.size // 1000) if l1 else 0):
                return "layer %d %s differs: max abs %.3g of %.3g in %d of %d elements (options %s)" % (
                    i, name, np.abs(u - v).max(), np.abs(u).max(), (~close).sum(), u.size, opts)
    return None


def sibling_case(rs, tmp, c, tool):
    """nnet-train-xent-hardlab-perutt (one update per utterance) / nnet-train-mse-tgtmat-frmshuff (matrix targets)"""
    opts = ["--learn-rate=%g" % rs.choice([0.002, 0.01]), "--binary=%s" % ("true" if rs.rand() < 0.5 else "false")]
    if rs.rand() < 0.4:
        opts.append("--momentum=%g" % rs.choice([0.5, 0.9]))
    if rs.rand() < 0.25:
        opts.append("--l1-penalty=%g" % rs.choice([1e-5, 1e-4]))
    if rs.rand() < 0.2:
        opts.append("--cross-validate=true")
    if c["transform"]:
        opts.append("--feature-transform=" + c["transform"])
    second = "ark:" + os.path.join(tmp, "a.ark")
    pattern = r"Xent:(\S+) frames:(\d+) err/frm:(\S+)"
    if tool == "nnet-train-mse-tgtmat-frmshuff":
        bunch = int(rs.choice([7, 64, 256]))
        opts += ["--bunchsize=%d" % bunch, "--cachesize=%d" % (bunch * int(rs.choice([1, 4, 16])))]
        if rs.rand() < 0.3:
            opts.append("--average-grad=true")
        # posterior-like targets per frame
        feats = O.read_matrix_ark(open(os.path.join(tmp, "f.ark"), "rb").read())
        tg = [(k, rs.random_sample((m.shape[0], c["dims"][-1])).astype(f32)) for k, m in feats]
        open(os.path.join(tmp, "tgt.ark"), "wb").write(O.write_matrix_ark(tg))
        second = "ark:" + os.path.join(tmp, "tgt.ark")
        pattern = r"Mse:(\S+) frames:(\d+) err/frm:(\S+)"
    cv = "--cross-validate=true" in opts
    outs = {}
    # (the command, for a closer look at a kept case: tool, then the arguments without the output file)
    open(os.path.join(tmp, "cmd.txt"), "w").write("\n".join([tool] + opts + ["ark:" + os.path.join(tmp, "f.ark"), second, os.path.join(tmp, "m.nnet")]))
    for who, d in (("ref", REF), ("ours", OURS)):
        out = os.path.join(tmp, "out_%s.nnet" % who)
        rc, err = run(os.path.join(d, tool), opts + ["ark:" + os.path.join(tmp, "f.ark"), second, os.path.join(tmp, "m.nnet")] +
                      ([] if cv else [out]))
        outs[who] = (rc, err, out)
    if outs["ref"][0] != outs["ours"][0]:
        return "exit codes differ: reference %d, ours %d (options %s)\n%s\n%s" % (
            outs["ref"][0], outs["ours"][0], opts, outs["ref"][1][-800:], outs["ours"][1][-800:])
    if outs["ref"][0] != 0:
        return None
    rep = {}
    for who in outs:
        m = re.search(pattern, outs[who][1])
        if not m:
            return "no report from %s:\n%s" % (who, outs[who][1][-600:])
        rep[who] = (float(m.group(1)), int(m.group(2)))
    if rep["ref"][1] != rep["ours"][1]:
        return "frame counts differ: %s %s (options %s)" % (rep["ref"], rep["ours"], opts)
    if _loss_differs(rep["ref"][0], rep["ours"][0], one_sided=(tool != "nnet-train-mse-tgtmat-frmshuff")):
        return "reports differ: reference %s, ours %s (options %s)" % (rep["ref"], rep["ours"], opts)
    return None if cv else _compare_models(outs["ref"][2], outs["ours"][2], opts)


def copy_case(rs, tmp, c):
    """nnet-copy (text <-> binary) of the case's model and rbm-convert-to-nnet of a random RBM: the written files
    must be the reference's byte for byte."""
    vis, hid = int(rs.choice([13, 40, 333])), int(rs.choice([17, 256, 100]))
    rbm = O.Rbm((0.1 * rs.standard_normal((hid, vis))).astype(f32), (0.1 * rs.standard_normal(vis)).astype(f32),
                (0.1 * rs.standard_normal(hid)).astype(f32), str(rs.choice(["bern", "gauss"])), "bern")
    open(os.path.join(tmp, "r.rbm"), "wb").write(O.write_nnet([rbm], bool(rs.rand() < 0.5)))
    jobs = []
    for binary in ("true", "false"):
        jobs.append(("nnet-copy", ["--binary=" + binary, os.path.join(tmp, "m.nnet")], "copy_%s" % binary))
        jobs.append(("rbm-convert-to-nnet", ["--binary=" + binary, os.path.join(tmp, "r.rbm")], "conv_%s" % binary))
    for tool, args, tag in jobs:
        out = {}
        for who, d in (("ref", REF), ("ours", OURS)):
            path = os.path.join(tmp, "%s_%s.out" % (tag, who))
            rc, err = run(os.path.join(d, tool), args + [path])
            out[who] = (rc, open(path, "rb").read() if rc == 0 and os.path.exists(path) else None, err)
        if out["ref"][0] != out["ours"][0]:
            return "%s %s: exit codes differ: %d %d\n%s" % (tool, args[0], out["ref"][0], out["ours"][0], out["ours"][2][-600:])
        if out["ref"][1] != out["ours"][1]:
            return "%s %s: files differ (%d / %d bytes)" % (tool, args[0], len(out["ref"][1] or b""), len(out["ours"][1] or b""))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--keep", default="")
    ap.add_argument("--first", type=int, default=0, help="index of the first case (every case has a seed of its own)")
    ap.add_argument("--exotic", action="store_true", help="also <relu> / <softrelu> / <scale> layers, text alignment "
                    "archives, utterances without or with mis-sized alignments")
    ap.add_argument("--siblings", action="store_true", help="also nnet-train-xent-hardlab-perutt, nnet-train-mse-tgtmat-frmshuff, nnet-copy and rbm-convert-to-nnet")
    args = ap.parse_args()
    for t in ("nnet-forward", "nnet-train-xent-hardlab-frmshuff"):
        for d in (REF, OURS):
            if not os.path.exists(os.path.join(d, t)):
                sys.exit("missing " + os.path.join(d, t))
    bad = 0
    for i in range(args.first, args.first + args.cases):
        rs = np.random.RandomState(args.seed * 1000 + i)
        tmp = tempfile.mkdtemp(prefix="diff_fuzz_")
        c = make_case(rs, tmp, args.exotic)
        kind = ("forward", "train", "perutt", "mse", "copy")[i % 5] if args.siblings else ("forward", "train")[i % 2]
        if kind == "forward":
            msg = forward_case(rs, tmp, c)
        elif kind == "train":
            msg = train_case(rs, tmp, c)
        elif kind == "copy":
            msg = copy_case(rs, tmp, c)
        else:
            msg = sibling_case(rs, tmp, c, "nnet-train-xent-hardlab-perutt" if kind == "perutt" else "nnet-train-mse-tgtmat-frmshuff")
        print("case %d (%s, dims %s, %d utterances, %s model): %s" % (
            i, kind, c["dims"], len(c["lens"]), "binary" if c["binary"] else "text", "ok" if msg is None else "DIFFERS: " + msg), flush=True)
        if msg is not None:
            bad += 1
            if args.keep:
                shutil.copytree(tmp, os.path.join(args.keep, "case%d" % i))
        shutil.rmtree(tmp)
    print("%d of %d cases differ" % (bad, args.cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
