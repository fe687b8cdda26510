#!/usr/bin/env python3
"""Class-level differential check of the RBM: the same random walk over the Rbm API (tests/fuzz/rbm_walk.cc) run
against the reference's classes (CPU branch, oracle/_ref/bin/rbm_walk) and against this build's on the GPU
(kaldi_amd/bin/rbm_walk), on random stacks of one to three <rbm> components (Gaussian or Bernoulli visible units at
the bottom, Bernoulli or Gaussian hidden units at the top, text or binary files).  The top component is trained by
contrastive-divergence steps on probabilities; outputs, reconstructions, parameters and the written forms
(<rbm>, WriteAsNnet, the auto-encoder halves) are compared record by record.

usage: python tests/fuzz/rbm_walk.py [--cases N] [--steps K] [--seed S] [--out FILE]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nnet_walk as W  # noqa: E402

O, f32 = W.O, np.float32
REF = os.path.join(W.ROOT, "oracle", "_ref", "bin", "rbm_walk")
OURS = os.path.join(W.ROOT, "kaldi_amd", "bin", "rbm_walk")
UPDATES = ("cd1_err", "update_weight")


def random_stack(rs):
    depth = int(rs.choice([1, 1, 2, 3]))
    dims = [int(rs.choice([3, 7, 16, 33, 64, 100, 257])) for _ in range(depth + 1)]
    layers = []
    for k in range(depth):
        vis_type = "gauss" if (k == 0 and rs.rand() < 0.5) else "bern"
        hid_type = "gauss" if (k == depth - 1 and rs.rand() < 0.25) else "bern"
        layers.append(O.Rbm((0.1 * rs.standard_normal((dims[k + 1], dims[k]))).astype(f32),
                            (0.1 * rs.standard_normal(dims[k])).astype(f32),
                            (0.1 * rs.standard_normal(dims[k + 1]) - 1.0).astype(f32), vis_type, hid_type))
    return layers, " ".join("%s-%s %dx%d" % (l.vis_type, l.hid_type, l.dim_in, l.dim_out) for l in layers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    for b in (REF, OURS):
        if not os.path.exists(b):
            sys.exit("missing " + b + " (make -C oracle ref; make -C kaldi_amd/csrc host)")
    rs = np.random.RandomState(args.seed)
    bad, counts, records = [], {}, 0
    with tempfile.TemporaryDirectory() as tmp:
        for case in range(args.cases):
            layers, desc = random_stack(rs)
            binary = bool(rs.rand() < 0.5)
            model = os.path.join(tmp, "m.rbm")
            open(model, "wb").write(O.write_nnet(layers, binary))
            name = "case %d (%s, %s)" % (case, desc, "binary" if binary else "text")
            fr, fo = os.path.join(tmp, "ref.bin"), os.path.join(tmp, "ours.bin")
            rc_r, err_r = W.run(REF, model, 300 + case, args.steps, fr)
            rc_o, err_o = W.run(OURS, model, 300 + case, args.steps, fo)
            if rc_r != 0 or rc_o != 0:
                bad.append("%s: exit %d / %d\n    ref : %s\n    ours: %s" % (name, rc_r, rc_o, err_r[-300:], err_o[-300:]))
                continue
            ref, ours = W.read_dump(fr), W.read_dump(fo)
            records += len(ref)
            W.compare(name, ref, ours, bad, counts, update_tags=UPDATES)
            print("%s: %d records, %d differences so far" % (name, len(ref), len(bad)), flush=True)
    lines = ["rbm_walk: %d stacks x %d steps, %d records compared, %d differ" % (args.cases, args.steps, records, len(bad))]
    lines += ["  %-28s %d" % (k, counts[k]) for k in sorted(counts)]
    lines += ["DIFF " + b for b in bad[:40]]
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
