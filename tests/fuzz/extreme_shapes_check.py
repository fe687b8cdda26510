#!/usr/bin/env python3
"""Shapes at the ends of what the kernels plan for, through the forward tool and the trainer of this build and of the
reference (oracle/_ref/bin): very wide and one-column output layers, one-unit hidden layers, widths that are not
multiples of 4, bunches of 1, 3 and 4096 frames, one-frame utterances only.  Outputs and trained weights must agree.

    python tests/fuzz/extreme_shapes_check.py [--out FILE]
"""
import argparse
import os
import re
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rs = np.random.RandomState(33)
    tmp = tempfile.mkdtemp(prefix="extreme_")
    cases = [  # (name, dims, utterance lengths, trainer options)
        ("32000 outputs", [40, 64, 32000], [50, 300], ["--bunchsize=64", "--cachesize=512"]),
        ("70001 outputs, 3 hidden units", [13, 3, 70001], [40], ["--bunchsize=8", "--cachesize=64"]),
        ("one output", [24, 16, 1], [30, 70], ["--bunchsize=16", "--cachesize=64"]),
        ("two outputs, one hidden unit", [24, 1, 2], [30, 70], ["--bunchsize=16", "--cachesize=64"]),
        ("one input column", [1, 33, 7], [100, 41], ["--bunchsize=32", "--cachesize=128"]),
        ("widths 1023 / 2047 / 1025", [39, 1023, 2047, 1025], [300, 200, 150], ["--bunchsize=256", "--cachesize=512"]),
        ("bunch of 1", [20, 30, 9], [10, 13], ["--bunchsize=1", "--cachesize=8"]),
        ("bunch of 3, cache of 3", [20, 30, 9], [10, 13, 5], ["--bunchsize=3", "--cachesize=3"]),
        ("bunch of 4096", [40, 128, 50], [3000, 3000, 3000], ["--bunchsize=4096", "--cachesize=8192"]),
        ("bunch of 5000 (not a power of two), cache 5000", [40, 100, 50], [2600, 2700, 2800], ["--bunchsize=5000", "--cachesize=5000"]),
        ("one-frame utterances only", [24, 32, 11], [1] * 40, ["--bunchsize=8", "--cachesize=16"]),
        ("utterance of 20000 frames", [13, 64, 40], [20000], ["--bunchsize=256", "--cachesize=16384"]),
        ("9 layers of 5 units", [7] + [5] * 8 + [4], [90, 60], ["--bunchsize=32", "--cachesize=64", "--momentum=0.9", "--l1-penalty=0.0001"]),
        ("momentum 0.99, learn rate 1e-7", [24, 40, 11], [100, 80], ["--bunchsize=16", "--cachesize=64", "--momentum=0.99", "--learn-rate=1e-7"]),
        ("learn rate 0 (nothing moves)", [24, 40, 11], [100, 80], ["--bunchsize=16", "--cachesize=64", "--learn-rate=0"]),
    ]
    lines, bad = [], 0
    for name, dims, lens, topts in cases:
        layers = []
        for i in range(len(dims) - 1):
            layers.append(O.BiasedLinearity((rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(f32),
                                            (0.1 * rs.standard_normal(dims[i + 1])).astype(f32)))
            layers.append(O.Softmax(dims[i + 1]) if i + 2 == len(dims) else O.Sigmoid(dims[i + 1]))
        open(tmp + "/m.nnet", "wb").write(O.write_nnet(layers, True))
        utts = [("utt%03d" % i, rs.standard_normal((n, dims[0])).astype(f32)) for i, n in enumerate(lens)]
        open(tmp + "/f.ark", "wb").write(O.write_matrix_ark(utts))
        open(tmp + "/a.ark", "wb").write(O.write_int32_vector_ark([(k, rs.randint(0, dims[-1], m.shape[0]).astype(np.int32)) for k, m in utts]))
        why = []
        got = {}
        for who, d in (("ref", REF), ("ours", OURS)):
            r1 = subprocess.run([os.path.join(d, "nnet-forward"), tmp + "/m.nnet", "ark:" + tmp + "/f.ark", "ark:%s/post_%s.ark" % (tmp, who)],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            opts = topts if any(o.startswith("--learn-rate") for o in topts) else topts + ["--learn-rate=0.004"]
            r2 = subprocess.run([os.path.join(d, "nnet-train-xent-hardlab-frmshuff"), "--binary=true"] + opts +
                                ["ark:" + tmp + "/f.ark", "ark:" + tmp + "/a.ark", tmp + "/m.nnet", "%s/out_%s.nnet" % (tmp, who)],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            got[who] = (r1.returncode, r2.returncode, r1.stderr.decode(errors="replace"), r2.stderr.decode(errors="replace"))
        if (got["ref"][0] == 0) != (got["ours"][0] == 0):
            why.append("forward status %s / %s: %s" % (got["ref"][0], got["ours"][0], got["ours"][2].strip().splitlines()[-1:][0][:200] if got["ours"][2].strip() else ""))
        elif got["ref"][0] == 0:
            a = O.read_matrix_ark(open(tmp + "/post_ref.ark", "rb").read())
            b = O.read_matrix_ark(open(tmp + "/post_ours.ark", "rb").read())
            if [k for k, _ in a] != [k for k, _ in b] or any(x.shape != y.shape for (_, x), (_, y) in zip(a, b)):
                why.append("forward keys / shapes differ")
            else:
                worst = max(float(np.max(np.abs(x - y))) for (_, x), (_, y) in zip(a, b))
                if not all(np.allclose(x, y, rtol=5e-5, atol=1e-7) for (_, x), (_, y) in zip(a, b)):
                    why.append("posteriors differ (max |diff| %.3g)" % worst)
        if (got["ref"][1] == 0) != (got["ours"][1] == 0):
            e = got["ours"][3] if got["ours"][1] != 0 else got["ref"][3]
            why.append("train status %s / %s: %s" % (got["ref"][1], got["ours"][1], " | ".join([l for l in e.splitlines() if "ERROR" in l or "ASSERT" in l][:1])[:240]))
        elif got["ref"][1] == 0:
            la = [l for l in O.read_nnet(open(tmp + "/out_ref.nnet", "rb").read()) if isinstance(l, O.BiasedLinearity)]
            lb = [l for l in O.read_nnet(open(tmp + "/out_ours.nnet", "rb").read()) if isinstance(l, O.BiasedLinearity)]
            for k, (x, y) in enumerate(zip(la, lb)):
                scale = max(1e-6, float(np.abs(x.W).max()))
                if not np.allclose(x.W, y.W, rtol=3e-4, atol=3e-6 * scale) or not np.allclose(x.b, y.b, rtol=3e-4, atol=3e-6 * max(1e-6, float(np.abs(x.b).max()))):
                    why.append("layer %d weights differ (max |dW| %.3g of %.3g)" % (k, float(np.abs(x.W - y.W).max()), scale))
                    break
            fr = [re.findall(r"frames:(\d+)", got[w][3])[-1:] for w in ("ref", "ours")]
            acc = [re.findall(r"FRAME_ACCURACY >> ([-0-9.naif]+)%", got[w][3])[-1:] for w in ("ref", "ours")]
            if fr[0] != fr[1]:
                why.append("frames %s / %s" % (fr[0], fr[1]))
            if acc[0] and acc[1] and abs(float(acc[0][0]) - float(acc[1][0])) > 0.5:
                why.append("accuracy %s / %s" % (acc[0][0], acc[1][0]))
        bad += 1 if why else 0
        lines.append("%-48s dims %-28s forward rc %s/%s train rc %s/%s  %s" % (name, "-".join(map(str, dims)), got["ref"][0], got["ours"][0],
                                                                                got["ref"][1], got["ours"][1], ("<<< " + "; ".join(why)) if why else "ok"))
        print(lines[-1], flush=True)
    lines.insert(0, "extreme_shapes_check: %d cases, %d differ" % (len(cases), bad))
    print(lines[0])
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
