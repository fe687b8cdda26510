#!/usr/bin/env python3
"""Class-level differential check of the network layer: the same random walk over the Nnet / Component / Xent / Mse /
MseProgress / Cache API (tests/fuzz/nnet_walk.cc) run against the reference's classes (CPU branch,
oracle/_ref/bin/nnet_walk) and against this build's classes on the GPU (kaldi_amd/bin/nnet_walk), on random model
files written here (1-5 components, text or binary, with and without a softmax; an empty file too).  The dumps are
compared record by record: matrices to float rounding, text token by token (numbers to float rounding, everything
else exactly).

usage: python tests/fuzz/nnet_walk.py [--cases N] [--steps K] [--seed S] [--out FILE]
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32
REF = os.path.join(ROOT, "oracle", "_ref", "bin", "nnet_walk")
OURS = os.path.join(ROOT, "kaldi_amd", "bin", "nnet_walk")


def random_model(rs):
    """(layers, description): mostly affine + activation stacks; now and then a lone component or nothing at all."""
    kind = rs.randint(0, 12)
    if kind == 0:
        return [], "empty"
    dim = lambda: int(rs.choice([3, 7, 16, 33, 64, 100, 257]))  # noqa: E731
    affine = lambda o, i: O.BiasedLinearity((rs.standard_normal((o, i)) / np.sqrt(i)).astype(f32),  # noqa: E731
                                            (0.1 * rs.standard_normal(o)).astype(f32))
    if kind == 1:
        d = dim()
        one = [O.Softmax(d), O.Sigmoid(d), O.Relu(d), O.SoftRelu(d), O.Scale(d, 0.5)][rs.randint(0, 5)]
        return [one], "lone " + type(one).__name__
    if kind == 2:
        i, o = dim(), dim()
        return [affine(o, i)], "lone affine"
    layers, dims = [], [dim() for _ in range(rs.randint(2, 5))]
    if rs.rand() < 0.3:  # a splice (and perhaps a column selection) in front, as a feature transform would be
        ctx = int(rs.choice([1, 2]))
        layers.append(O.Expand(dims[0], dims[0] * (2 * ctx + 1), list(range(-ctx, ctx + 1))))
        width = dims[0] * (2 * ctx + 1)
        if rs.rand() < 0.5:
            keep = [int(k) for k in rs.permutation(width)[:max(2, width // 2)]]
            layers.append(O.Copy(width, len(keep), keep))
            width = len(keep)
        dims = [dims[0], width] + dims[1:]
        start = 1
    else:
        start = 0
    for k in range(start, len(dims) - 1):
        layers.append(affine(dims[k + 1], dims[k]))
        last = k + 2 == len(dims)
        if last:
            end = rs.randint(0, 4)
            if end <= 1:
                layers.append(O.Softmax(dims[k + 1]))
            elif end == 2:
                layers.append(O.Sigmoid(dims[k + 1]))
        else:
            act = rs.choice(["sigmoid", "sigmoid", "relu", "softrelu", "scaled", "none"])
            if act == "relu":
                layers.append(O.Relu(dims[k + 1]))
            elif act == "softrelu":
                layers.append(O.SoftRelu(dims[k + 1]))
            elif act != "none":
                layers.append(O.Sigmoid(dims[k + 1]))
                if act == "scaled":
                    layers.append(O.Scale(dims[k + 1], 2.0))
    return layers, " ".join(type(l).__name__ for l in layers)


def read_dump(path):
    recs = []
    with open(path, "rb") as f:
        while True:
            head = f.readline()
            if not head:
                break
            kind, tag, r, c = head.decode().split()
            r, c = int(r), int(c)
            if kind == "mat":
                val = np.frombuffer(f.read(4 * r * c), dtype=np.float32).reshape(r, c)
            else:
                val = f.read(r).decode(errors="replace")
            assert f.read(1) == b"\n", "dump out of step at " + tag
            recs.append((kind, tag, val))
    return recs


def run(binary, model, seed, steps, path):
    p = subprocess.run([binary, model, str(seed), str(steps), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    return p.returncode, p.stderr.decode(errors="replace")


def _number(tok):
    try:
        return float(tok)
    except ValueError:
        return None


def text_differs(a, b, rtol, atol, xent=False, loose_scale=1.0):
    """Token by token ("Xent:12.5" counts as two tokens).  xent: the summed cross-entropy of a report may be the
    smaller one here -- the reference's CUDA kernel floors the target's posterior at 1e-20 before the logarithm
    (cu-kernels.cu:733-745) and so does this build, the reference's CPU branch, which the walk runs, does not
    (cu-math-inl.h:373-377): where posteriors fall under the floor its figure is larger, up to inf."""
    ta, tb = a.replace(":", ": ").split(), b.replace(":", ": ").split()
    if len(ta) != len(tb):
        return "token counts %d / %d" % (len(ta), len(tb))
    # (a written matrix: the rounding of an element goes with the size of the sums behind it, not with its own
    # size, so small elements are held to the scale of the largest one, as in the comparison of dumped matrices)
    nums = [abs(v) for v in (_number(t.strip("[]%(),")) for t in ta) if v is not None and np.isfinite(v)]
    if len(nums) > 16:
        atol = max(atol, 1e-5 * loose_scale) * max(1.0, max(nums))
    for k, (x, y) in enumerate(zip(ta, tb)):
        if x == y:
            continue
        if xent and k > 0 and ta[k - 1] in ("Xent:", "err/frm:") and _number(x) is not None and _number(y) is not None \
                and _number(y) < _number(x):
            continue
        fx, fy = _number(x.strip("[]%(),")), _number(y.strip("[]%(),"))
        if fx is None or fy is None:
            return "%r / %r" % (x, y)
        if not (abs(fx - fy) <= atol + rtol * abs(fx) or (np.isnan(fx) and np.isnan(fy))):
            return "%r / %r" % (x, y)
    return None


def compare(name, ref, ours, bad, counts, update_tags=("train_err",)):
    if len(ref) != len(ours):
        bad.append("%s: %d / %d records" % (name, len(ref), len(ours)))
    # after training steps the two drift apart by rounding; the tolerance follows the number of updates so far
    updates = 0
    # <relu> (and the L1 step, see below) has a kink: where a pre-activation is within rounding of zero the two code
    # bases can land on different sides of it, and the back-propagated error through that unit is all or nothing.  In
    # such a walk a record that disagrees by no more than 10 % of its largest element is counted as a kink event (not
    # as a difference), and the rest of the walk -- now on slightly different weights -- is held to a 30 times wider
    # tolerance (twice at most).  Walks without <relu> and before their first L1 step stay at rounding.
    relu, loosen = "Relu" in name, 1.0
    for (ka, ta, va), (kb, tb, vb) in zip(ref, ours):
        if ka == "text" and ta.endswith("_knobs"):
            # "... | global | rate momentum l2 l1 average | ..." per updatable layer: an L1 penalty is a kink of the same
            # kind (the step sets a weight to zero when its sign would change, cu-math-inl.h RegularizeL1)
            groups = [g.split() for g in va.split("|")[2:]]
            if any(len(g) >= 4 and float(g[3]) != 0.0 for g in groups):
                relu = True
        if (ka, ta) != (kb, tb):
            bad.append("%s: record order differs: %s / %s" % (name, ta, tb))
            return
        what = ta.split("_", 1)[1] if ta.startswith("step") else ta
        counts[what] = counts.get(what, 0) + 1
        if what in update_tags:
            updates += 1
        rtol = 2e-5 * (1 + updates) ** 0.5 * loosen
        atol = 2e-6 * (1 + updates) ** 0.5 * loosen
        if ka == "mat" and va.size and (not np.isfinite(va).all() or float(np.abs(va).max()) > 1e3):
            # a network without a bounded output trained on random targets can blow up: from there on the two are
            # amplified rounding (and end in inf / nan); the rest of this walk is not compared
            counts["walks that blew up in the reference"] = counts.get("walks that blew up in the reference", 0) + 1
            return
        if ka == "mat":
            if va.size and np.isfinite(va).all():
                atol *= max(1.0, float(np.abs(va).max()))  # (elements near zero beside large ones)
            if va.shape != vb.shape:
                bad.append("%s %s: shape %s / %s" % (name, ta, va.shape, vb.shape))
            elif not np.allclose(va, vb, rtol=rtol, atol=atol, equal_nan=True):
                scale = float(np.abs(va).max()) if va.size and np.isfinite(va).all() else 0.0
                worst = float(np.nanmax(np.abs(va.astype(np.float64) - vb))) if va.size else 0.0
                if relu and loosen < 1000.0 and scale > 0.0 and worst <= 0.1 * scale:
                    # one unit (or one weight, and with momentum its whole history) on the other side of a kink: small
                    # against the matrix, but far outside rounding, and it touches whatever that unit feeds
                    counts["kink events (relu / L1)"] = counts.get("kink events (relu / L1)", 0) + 1
                    loosen *= 30.0
                    continue
                d = np.abs(va.astype(np.float64) - vb)
                d = np.where(np.isfinite(d), d, np.inf if not np.array_equal(np.isnan(va), np.isnan(vb)) else 0.0)
                k = np.unravel_index(np.argmax(d), d.shape)
                bad.append("%s %s: max |diff| %.3g at %s: ref %.9g ours %.9g (after %d updates)" %
                           (name, ta, d[k], k, va[k], vb[k], updates))
        else:
            if loosen > 1.0 and what in ("write_text", "write_front", "component_write"):
                # after a kink event single weights (with momentum, their whole history) differ outright
                counts["written models not compared after a kink"] = counts.get("written models not compared after a kink", 0) + 1
                continue
            why = text_differs(va, vb, max(rtol, 1e-4), atol, xent=what in ("xent_report", "soft_report"), loose_scale=(1 + updates) ** 0.5)
            if why:
                bad.append("%s %s: %s\n    ref : %s\n    ours: %s" % (name, ta, why, va[:300], vb[:300]))


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
            layers, desc = random_model(rs)
            binary = bool(rs.rand() < 0.5)
            model = os.path.join(tmp, "m.nnet")
            open(model, "wb").write(O.write_nnet(layers, binary))
            name = "case %d (%s, %s)" % (case, desc, "binary" if binary else "text")
            fr, fo = os.path.join(tmp, "ref.bin"), os.path.join(tmp, "ours.bin")
            rc_r, err_r = run(REF, model, 100 + case, args.steps, fr)
            rc_o, err_o = run(OURS, model, 100 + case, args.steps, fo)
            if rc_r != 0 or rc_o != 0:
                # both may refuse the same thing; one alone is a difference
                if (rc_r != 0) != (rc_o != 0):
                    bad.append("%s: exit %d / %d\n    ref : %s\n    ours: %s" % (name, rc_r, rc_o, err_r[-300:], err_o[-300:]))
                else:
                    counts["both refuse"] = counts.get("both refuse", 0) + 1
                    print(name + ": both end with an error: " + err_r.strip().splitlines()[0][:120], flush=True)
                continue
            ref, ours = read_dump(fr), read_dump(fo)
            records += len(ref)
            compare(name, ref, ours, bad, counts)
            print("%s: %d records, %d differences so far" % (name, len(ref), len(bad)), flush=True)
    lines = ["nnet_walk: %d models x %d steps, %d records compared, %d differ" % (args.cases, args.steps, records, len(bad))]
    lines += ["  %-24s %d" % (k, counts[k]) for k in sorted(counts)]
    lines += ["DIFF " + b for b in bad[:40]]
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
