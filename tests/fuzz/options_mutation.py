"""Random command lines -- registered options with good and bad values, both spellings (--opt=value, --opt value),
unknown options, options in odd places, wrong argument counts -- through nnet-forward and the trainer of both builds
(on the GPU box): the same accept / refuse and, where both accept, the same results.

usage: python tests/fuzz/options_mutation.py [--cases N] [--seed S] [--out file]"""
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
from table_mutation import text_ali_ark  # noqa: E402

OURS = os.path.join(ROOT, "kaldi_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref", "bin")
f32 = np.float32

FLOATS = ["0.1", "1e-3", "0", "1", "-0.5", "abc", "", "1,2", "0x10", " 1", "1 ", "1e", ".5", "+2", "nan", "inf", "1e999", "1.0f"]
BOOLS = ["true", "false", "t", "f", "1", "0", "yes", "TRUE", "True", "maybe", "", "tr", "F", "T", "no", "2"]
INTS = ["8", "1", "0", "-3", "1.5", "abc", "", "99999999999", "16", "0x8", " 8", "8 ", "+8", "64", "1e2"]
FORWARD = {"prior-scale": FLOATS, "apply-log": BOOLS, "no-softmax": BOOLS, "silent": BOOLS, "print-args": BOOLS, "verbose": INTS}
# (no --l2-penalty: the reference's CPU branch of that update asserts, matrix/kaldi-matrix.cc:281)
TRAIN = {"learn-rate": FLOATS, "momentum": FLOATS, "l2-upper-bound": FLOATS, "l1-penalty": FLOATS,
         "average-grad": BOOLS, "binary": BOOLS, "cross-validate": BOOLS, "randomize": BOOLS, "bunchsize": INTS,
         "cachesize": INTS, "learn-factors": ["1,0.5", "1:0.5", "1", "1,0.5,2", "a,b", "", "0,0", "1;2"], "print-args": BOOLS}


def run(d, tool, argv):
    try:
        r = subprocess.run([os.path.join(d, tool)] + argv, capture_output=True, timeout=15 if d == REF else 100)
        return r.returncode, r.stderr.decode(errors="replace")
    except subprocess.TimeoutExpired:
        return "timeout", ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    D, H, C = 13, 24, 9
    layers = [O.BiasedLinearity((rs.randn(H, D) / 4).astype(f32), (0.1 * rs.randn(H)).astype(f32)), O.Sigmoid(H),
              O.BiasedLinearity((rs.randn(C, H) / 5).astype(f32), (0.1 * rs.randn(C)).astype(f32)), O.Softmax(C)]
    utts = [("utt%d" % i, rs.randn(int(rs.randint(5, 30)), D).astype(f32)) for i in range(5)]
    alis = [(k, rs.randint(0, C, m.shape[0])) for k, m in utts]
    counts, differ = {}, []
    pick = lambda seq: seq[int(rs.randint(0, len(seq)))]  # noqa: E731
    with tempfile.TemporaryDirectory() as tmp:
        P = lambda n: os.path.join(tmp, n)  # noqa: E731
        open(P("m.nnet"), "wb").write(O.write_nnet(layers, True))
        open(P("f.ark"), "wb").write(O.write_matrix_ark(utts))
        open(P("a.ark"), "wb").write(text_ali_ark(alis))
        for case in range(args.cases):
            forward = rs.rand() < 0.4
            table = FORWARD if forward else TRAIN
            opts = []
            for _ in range(int(rs.randint(0, 4))):
                r = rs.rand()
                if r < 0.08:
                    opts.append(pick(["--nosuch=1", "-x", "--", "--=3", "--help", "-", "--learn-rate", "--binary"]))
                    continue
                name = pick(sorted(table))
                # mostly good values, so that command lines get past the first option
                value = table[name][0] if rs.rand() < 0.45 else pick(table[name])
                if rs.rand() < 0.15:
                    opts += ["--" + name, value]          # the two-word spelling
                else:
                    opts.append("--%s=%s" % (name, value))
            res = {}
            for which, d in (("ref", REF), ("ours", OURS)):
                out = P("out_" + which)
                if os.path.exists(out):
                    os.remove(out)
                if forward:
                    pos = [P("m.nnet"), "ark:" + P("f.ark"), "ark:" + out]
                else:
                    pos = ["ark:" + P("f.ark"), "ark:" + P("a.ark"), P("m.nnet"), out]
                how = rs.rand() if which == "ref" else how  # noqa: F821  (the same shape of command line for both)
                if how < 0.06:
                    pos = pos[:-1]                          # an argument short
                elif how < 0.10:
                    pos = pos + ["extra"]
                argv = opts + pos if how < 0.85 else pos[:1] + opts + pos[1:]  # options between the arguments
                if not forward and "--cachesize" not in " ".join(opts):
                    argv = ["--cachesize=64"] + argv
                if not forward and "--bunchsize" not in " ".join(opts):
                    argv = ["--bunchsize=8"] + argv
                rc, err = run(d, "nnet-forward" if forward else "nnet-train-xent-hardlab-frmshuff", argv)
                res[which] = (rc, err, out, argv)
            (rc_r, err_r, out_r, argv_r), (rc_o, err_o, out_o, _) = res["ref"], res["ours"]
            kind = "both accept" if rc_r == 0 and rc_o == 0 else "both refuse" if rc_r != 0 and rc_o != 0 else "one refuses"
            why = []
            if rc_o not in (0, 1, 255, -6):
                why.append("this build ends with %s" % rc_o)
            if rc_r == "timeout":
                kind = "the reference does not end"
            elif kind == "one refuses":
                why.append("status %s / %s" % (rc_r, rc_o))
            elif kind == "both accept":
                if os.path.exists(out_r) != os.path.exists(out_o):
                    why.append("output written %s / %s" % (os.path.exists(out_r), os.path.exists(out_o)))
                elif os.path.exists(out_r) and forward:
                    a, b = O.read_matrix_ark(open(out_r, "rb").read()), O.read_matrix_ark(open(out_o, "rb").read())
                    if [k for k, _ in a] != [k for k, _ in b] or any(
                            x.shape != y.shape or not np.allclose(x, y, rtol=1e-4, atol=1e-6, equal_nan=True) for (_, x), (_, y) in zip(a, b)):
                        why.append("outputs differ")
                elif os.path.exists(out_r):
                    da, db = open(out_r, "rb").read(), open(out_o, "rb").read()
                    if (da[:2] == b"\0B") != (db[:2] == b"\0B"):
                        why.append("binary / text form of the written model differs")
                    else:
                        la = [l for l in O.read_nnet(da) if isinstance(l, O.BiasedLinearity)]
                        lb = [l for l in O.read_nnet(db) if isinstance(l, O.BiasedLinearity)]
                        big = max([float(np.nanmax(np.abs(x.W))) if np.isfinite(x.W).any() else np.inf for x in la] + [0.0])
                        if not all(np.isfinite(x.W).all() for x in la) or big > 50:
                            kind = "both accept, the training blew up"  # (a momentum of 16, a learning rate of inf ...)
                        elif any(not np.allclose(x.W, y.W, rtol=1e-3, atol=1e-5, equal_nan=True) for x, y in zip(la, lb)):
                            why.append("trained weights differ")
            counts[kind] = counts.get(kind, 0) + 1
            if case % 10 == 9:
                print("... %d cases, %d differ" % (case + 1, len(differ)), flush=True)
            if why:
                shown = " ".join(a.replace(tmp + "/", "") for a in argv_r)
                differ.append("case %d: %s\n    %s %s\n    ref : %s\n    ours: %s" % (
                    case, "; ".join(why), "nnet-forward" if forward else "nnet-train-xent-hardlab-frmshuff", shown,
                    err_r.strip().split("\n")[-1][:200] if err_r else "", err_o.strip().split("\n")[-1][:200] if err_o else ""))
    lines = ["options_mutation: %d command lines, %d differ" % (args.cases, len(differ))]
    lines += ["  %-28s %d" % kv for kv in sorted(counts.items())] + differ
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
    return 1 if differ else 0


if __name__ == "__main__":
    sys.exit(main())
