"""Damaged feature archives and alignments through nnet-forward and nnet-train-xent-hardlab-frmshuff of both builds (on
the GPU box): the same accept / refuse, the same utterances written before a refusal, the same values; never a signal
or a hang in this build.

usage: python tests/fuzz/archive_mutation.py [--cases N] [--seed S] [--out file]"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))
from oracle import nnet1_oracle as O  # noqa: E402
from table_mutation import mutate, text_matrix_ark, text_ali_ark, binary_ali_ark  # noqa: E402

OURS = os.path.join(ROOT, "kaldi_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref", "bin")
f32 = np.float32


def run(d, tool, argv):
    try:
        # (the reference never returns from an alignment of the wrong length: a short leash for it)
        r = subprocess.run([os.path.join(d, tool)] + argv, capture_output=True, timeout=12 if d == REF else 100)
        return r.returncode, r.stderr.decode(errors="replace")
    except subprocess.TimeoutExpired:
        return "timeout", ""


def read_ark(path):
    try:
        return O.read_matrix_ark(open(path, "rb").read())
    except Exception:  # a partial last entry
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", type=int, default=-1, help="one kind of case only (0..7, see the list in the loop)")
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    D, H, C = 13, 24, 9
    layers = [O.BiasedLinearity((rs.randn(H, D) / 4).astype(f32), (0.1 * rs.randn(H)).astype(f32)), O.Sigmoid(H),
              O.BiasedLinearity((rs.randn(C, H) / 5).astype(f32), (0.1 * rs.randn(C)).astype(f32)), O.Softmax(C)]
    counts, differ = {}, []
    with tempfile.TemporaryDirectory() as tmp:
        P = lambda n: os.path.join(tmp, n)  # noqa: E731
        open(P("m.nnet"), "wb").write(O.write_nnet(layers, True))
        open(P("r.rbm"), "wb").write(O.write_nnet([O.Rbm((0.1 * rs.randn(H, D)).astype(f32), np.zeros(D, f32), np.zeros(H, f32), "gauss", "bern")], True))
        for case in range(args.cases):
            utts = [("utt%d" % i, rs.randn(int(rs.randint(1, 30)), D).astype(f32)) for i in range(5)]
            alis = [(k, rs.randint(0, C, m.shape[0])) for k, m in utts]
            feats = O.write_matrix_ark(utts) if rs.rand() < 0.6 else text_matrix_ark(utts)
            ali = text_ali_ark(alis) if rs.rand() < 0.6 else binary_ali_ark(alis)
            # 0: forward on damaged features; 1: train on damaged features; 2: train on damaged alignments; then the sibling
            # trainers: 3 per-utterance, damaged features or alignments; 4 / 5 towards target matrices, damaged features /
            # targets; 6 contrastive divergence on damaged features (this build alone)
            # 7: forward with damaged class frame counts (a vector file, text or binary)
            what = int(rs.randint(0, 8))
            if args.only >= 0:
                what = args.only
            tgts = O.write_matrix_ark([(k, rs.rand(m.shape[0], C).astype(f32)) for k, m in utts])
            if what in (0, 1, 4, 6) or (what == 3 and rs.rand() < 0.5):
                feats = mutate(rs, feats)
            elif what == 5:
                tgts = mutate(rs, tgts)
            else:
                ali = mutate(rs, ali)
            open(P("f.ark"), "wb").write(feats)
            open(P("a.ark"), "wb").write(ali)
            open(P("t.ark"), "wb").write(tgts)
            if what == 7:
                cnt = rs.randint(1, 1000, C).astype(f32)
                if rs.rand() < 0.5:
                    counts_file = (" [ " + " ".join("%g" % v for v in cnt) + " ]\n").encode()
                else:
                    counts_file = b"\0BFV \x04" + np.int32(C).tobytes() + cnt.tobytes()
                open(P("c.counts"), "wb").write(mutate(rs, counts_file))
            res = {}
            for which, d in (("ref", REF), ("ours", OURS)):
                out = P("out_" + which)
                if os.path.exists(out):
                    os.remove(out)
                if what == 0:
                    rc, err = run(d, "nnet-forward", [P("m.nnet"), "ark:" + P("f.ark"), "ark:" + out])
                elif what == 7:
                    rc, err = run(d, "nnet-forward", ["--class-frame-counts=" + P("c.counts"), "--apply-log=true", P("m.nnet"),
                                                      "ark:" + P("f.ark"), "ark:" + out])
                elif what == 3:
                    rc, err = run(d, "nnet-train-xent-hardlab-perutt", ["--learn-rate=0.01", "--binary=true", "ark:" + P("f.ark"), "ark:" + P("a.ark"),
                                                                        P("m.nnet"), out])
                elif what in (4, 5):
                    rc, err = run(d, "nnet-train-mse-tgtmat-frmshuff",
                                  ["--bunchsize=8", "--cachesize=64", "--learn-rate=0.05", "--binary=true",
                                   "ark:" + P("f.ark"), "ark:" + P("t.ark"), P("m.nnet"), out])
                elif what == 6:
                    rc, err = run(d, "rbm-train-cd1-frmshuff", ["--bunchsize=8", "--cachesize=64", "--learn-rate=0.01", "--binary=true",
                                                                P("r.rbm"), "ark:" + P("f.ark"), out])
                else:
                    rc, err = run(d, "nnet-train-xent-hardlab-frmshuff",
                                  ["--bunchsize=8", "--cachesize=64", "--learn-rate=0.05", "--randomize=false", "--binary=true",
                                   "ark:" + P("f.ark"), "ark:" + P("a.ark"), P("m.nnet"), out])
                res[which] = (rc, err, out)
            (rc_r, err_r, out_r), (rc_o, err_o, out_o) = res["ref"], res["ours"]
            name = "case %d (%s)" % (case, ("forward, damaged features", "train, damaged features", "train, damaged alignments",
                                            "per-utterance trainer", "trainer towards targets, damaged features",
                                            "trainer towards targets, damaged targets", "rbm trainer, damaged features",
                                            "forward, damaged class frame counts")[what])
            kind = "both accept" if rc_r == 0 and rc_o == 0 else "both refuse" if rc_r != 0 and rc_o != 0 else "one refuses"
            why = []
            if rc_o not in (0, 1, 255, -6):
                why.append("this build ends with %s" % rc_o)
            if what == 6:
                # the reference's CPU build cannot run this tool on any input (its sampler's CPU branch never sizes its
                # output, cu-rand-inl.h:162-168): only this build's own end is looked at
                kind = "rbm trainer: " + ("trained" if rc_o == 0 else "refused")
            elif rc_r == "timeout":
                kind = "the reference does not end"   # (an alignment of the wrong length: it never returns)
            elif kind == "one refuses" and rc_r == 0 and "outside the network's" in err_o:
                # the reference's CPU branch indexes the posterior row with such a label (cu-math-inl.h:373-377)
                kind = "label outside the outputs: refused here only"
            elif kind == "one refuses":
                why.append("status %s / %s" % (rc_r, rc_o))
            elif what in (0, 7):
                a, b = read_ark(out_r) if os.path.exists(out_r) else [], read_ark(out_o) if os.path.exists(out_o) else []
                if a is None or b is None:
                    # the reference dies inside an entry now and then; whole entries are compared
                    a = a or []
                    b = b or []
                if kind == "both accept" or (a and b):
                    n = min(len(a), len(b)) if kind == "both refuse" else max(len(a), len(b))
                    if kind == "both accept" and [k for k, _ in a] != [k for k, _ in b]:
                        why.append("keys %s / %s" % ([k for k, _ in a], [k for k, _ in b]))
                    else:
                        for (ka, x), (kb, y) in list(zip(a, b))[:n]:
                            if ka != kb or x.shape != y.shape or not np.allclose(x, y, rtol=1e-4, atol=1e-6, equal_nan=True):
                                why.append("output %s / %s differs" % (ka, kb))
                                break
            elif kind == "both accept" and what not in (6, 7):
                la = [l for l in O.read_nnet(open(out_r, "rb").read()) if isinstance(l, O.BiasedLinearity)]
                lb = [l for l in O.read_nnet(open(out_o, "rb").read()) if isinstance(l, O.BiasedLinearity)]
                if any(not np.allclose(x.W, y.W, rtol=1e-3, atol=1e-5, equal_nan=True) for x, y in zip(la, lb)):
                    why.append("trained weights differ")
            counts[kind] = counts.get(kind, 0) + 1
            if case % 10 == 9:
                print("... %d cases, %d differ" % (case + 1, len(differ)), flush=True)
            if why:
                differ.append("%s: %s\n    ref : %s\n    ours: %s" % (name, "; ".join(why), err_r.strip().split("\n")[-1][:200] if err_r else "",
                                                                     err_o.strip().split("\n")[-1][:200] if err_o else ""))
                os.makedirs(os.path.join(ROOT, "gpurun_out", "archive_mutation"), exist_ok=True)
                for n in ("f.ark", "a.ark", "t.ark", "m.nnet") + (("c.counts",) if what == 7 else ()):
                    open(os.path.join(ROOT, "gpurun_out", "archive_mutation", "case%d_%s" % (case, n)), "wb").write(open(P(n), "rb").read())
    lines = ["archive_mutation: %d cases, %d differ" % (args.cases, len(differ))]
    lines += ["  %-28s %d" % kv for kv in sorted(counts.items())] + differ
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
    return 1 if differ else 0


if __name__ == "__main__":
    sys.exit(main())
