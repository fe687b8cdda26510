#!/usr/bin/env python3
"""Option parsing of the tools beside the reference's own (oracle/_ref/bin): the forms a flag can take
(--flag=value, boolean flags with and without a value, --config files, --print-args, --help), unknown flags and
values that do not parse.  Exit status, the output files and the usage / error text that scripts grep for must agree.

    python tests/fuzz/options_check.py [--out FILE]
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
REF = os.path.join(ROOT, "oracle", "_ref", "bin")
OURS = os.path.join(ROOT, "kaldi_amd", "bin")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rs = np.random.RandomState(3)
    tmp = tempfile.mkdtemp(prefix="opts_")
    layers = [O.BiasedLinearity((rs.standard_normal((16, 24)) / 5).astype(f32), np.zeros(16, f32)), O.Sigmoid(16),
              O.BiasedLinearity((rs.standard_normal((11, 16)) / 4).astype(f32), np.zeros(11, f32)), O.Softmax(11)]
    open(tmp + "/m.nnet", "wb").write(O.write_nnet(layers, True))
    utts = [("utt%d" % i, rs.standard_normal((n, 24)).astype(f32)) for i, n in enumerate([5, 70, 33])]
    open(tmp + "/f.ark", "wb").write(O.write_matrix_ark(utts))
    open(tmp + "/a.ark", "wb").write(O.write_int32_vector_ark([(k, rs.randint(0, 11, m.shape[0]).astype(np.int32)) for k, m in utts]))
    open(tmp + "/copy.conf", "w").write("--binary=false\n")
    open(tmp + "/copy_comment.conf", "w").write("# a comment\n--binary=false # trailing\n\n")
    open(tmp + "/bad.conf", "w").write("--nonsense=1\n")
    open(tmp + "/noprefix.conf", "w").write("binary=false\n")
    open(tmp + "/train.conf", "w").write("--learn-rate=0.004\n--momentum=0.5\n--bunchsize=8\n--cachesize=256\n")
    fwd = [tmp + "/m.nnet", "ark:" + tmp + "/f.ark", "ark:{out}"]
    trn = ["ark:" + tmp + "/f.ark", "ark:" + tmp + "/a.ark", tmp + "/m.nnet", "{out}"]
    cases = [
        ("copy plain", "nnet-copy", [tmp + "/m.nnet", "{out}"]),
        ("copy --binary=false", "nnet-copy", ["--binary=false", tmp + "/m.nnet", "{out}"]),
        ("copy --binary=true", "nnet-copy", ["--binary=true", tmp + "/m.nnet", "{out}"]),
        ("copy --binary (no value)", "nnet-copy", ["--binary", tmp + "/m.nnet", "{out}"]),
        ("copy --binary=f", "nnet-copy", ["--binary=f", tmp + "/m.nnet", "{out}"]),
        ("copy --binary=F", "nnet-copy", ["--binary=F", tmp + "/m.nnet", "{out}"]),
        ("copy --binary=0", "nnet-copy", ["--binary=0", tmp + "/m.nnet", "{out}"]),
        ("copy --binary=no", "nnet-copy", ["--binary=no", tmp + "/m.nnet", "{out}"]),
        ("copy --binary=", "nnet-copy", ["--binary=", tmp + "/m.nnet", "{out}"]),
        ("copy --binary false (two words)", "nnet-copy", ["--binary", "false", tmp + "/m.nnet", "{out}"]),
        ("copy -binary=false (one dash)", "nnet-copy", ["-binary=false", tmp + "/m.nnet", "{out}"]),
        ("copy --Binary=false (case)", "nnet-copy", ["--Binary=false", tmp + "/m.nnet", "{out}"]),
        ("copy --config", "nnet-copy", ["--config=" + tmp + "/copy.conf", tmp + "/m.nnet", "{out}"]),
        ("copy --config with comments", "nnet-copy", ["--config=" + tmp + "/copy_comment.conf", tmp + "/m.nnet", "{out}"]),
        ("copy --config then flag overrides", "nnet-copy", ["--config=" + tmp + "/copy.conf", "--binary=true", tmp + "/m.nnet", "{out}"]),
        ("copy flag then --config", "nnet-copy", ["--binary=true", "--config=" + tmp + "/copy.conf", tmp + "/m.nnet", "{out}"]),
        ("copy --config unknown flag inside", "nnet-copy", ["--config=" + tmp + "/bad.conf", tmp + "/m.nnet", "{out}"]),
        ("copy --config line without dashes", "nnet-copy", ["--config=" + tmp + "/noprefix.conf", tmp + "/m.nnet", "{out}"]),
        ("copy --config missing file", "nnet-copy", ["--config=" + tmp + "/nope.conf", tmp + "/m.nnet", "{out}"]),
        ("copy --print-args=false", "nnet-copy", ["--print-args=false", tmp + "/m.nnet", "{out}"]),
        ("copy --verbose=2", "nnet-copy", ["--verbose=2", tmp + "/m.nnet", "{out}"]),
        ("copy --help", "nnet-copy", ["--help"]),
        ("copy no arguments", "nnet-copy", []),
        ("copy three arguments", "nnet-copy", [tmp + "/m.nnet", "{out}", "extra"]),
        ("copy -- ends the flags", "nnet-copy", ["--binary=false", "--", tmp + "/m.nnet", "{out}"]),
        ("copy flag after positional", "nnet-copy", [tmp + "/m.nnet", "--binary=false", "{out}"]),
        ("forward --apply-log (no value)", "nnet-forward", ["--apply-log"] + fwd),
        ("forward --apply-log=true --silent=true", "nnet-forward", ["--apply-log=true", "--silent=true"] + fwd),
        ("forward --prior-scale=abc", "nnet-forward", ["--prior-scale=abc"] + fwd),
        ("forward --prior-scale=1e-1", "nnet-forward", ["--prior-scale=1e-1", "--no-softmax=false"] + fwd),
        ("forward --prior-scale=0.5x", "nnet-forward", ["--prior-scale=0.5x"] + fwd),
        ("forward --help", "nnet-forward", ["--help"]),
        ("train --config", "nnet-train-xent-hardlab-frmshuff", ["--config=" + tmp + "/train.conf"] + trn),
        ("train --bunchsize=4.5", "nnet-train-xent-hardlab-frmshuff", ["--bunchsize=4.5", "--cachesize=256"] + trn),
        ("train --bunchsize=0x8", "nnet-train-xent-hardlab-frmshuff", ["--bunchsize=0x8", "--cachesize=256"] + trn),
        ("train --bunchsize= 8 (space)", "nnet-train-xent-hardlab-frmshuff", ["--bunchsize= 8", "--cachesize=256"] + trn),
        ("train --learn-rate=1e-3 --cross-validate", "nnet-train-xent-hardlab-frmshuff", ["--learn-rate=1e-3", "--cross-validate", "--bunchsize=8", "--cachesize=256"] + trn[:3]),
        ("train --learn-factors=1,0.5", "nnet-train-xent-hardlab-frmshuff", ["--learn-factors=1,0.5", "--bunchsize=8", "--cachesize=256"] + trn),
        ("train --learn-factors=1,0.5,2 (too many)", "nnet-train-xent-hardlab-frmshuff", ["--learn-factors=1,0.5,2", "--bunchsize=8", "--cachesize=256"] + trn),
        ("train --help", "nnet-train-xent-hardlab-frmshuff", ["--help"]),
        ("rbm --help", "rbm-train-cd1-frmshuff", ["--help"]),
    ]
    lines, differ = [], 0
    for name, tool, argv in cases:
        res = []
        for which, d in (("ref", REF), ("ours", OURS)):
            out = os.path.join(tmp, "out_" + which)
            if os.path.exists(out):
                os.remove(out)
            a = [x.replace("{out}", out) for x in argv]
            try:
                r = subprocess.run([os.path.join(d, tool)] + a, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=100)
                rc, err, so = r.returncode, r.stderr.decode(errors="replace"), r.stdout
            except subprocess.TimeoutExpired:
                rc, err, so = "timeout", "", b""
            data = open(out, "rb").read() if os.path.exists(out) else None
            res.append((rc, err, data, so))
        (rc_r, err_r, out_r, so_r), (rc_o, err_o, out_o, so_o) = res
        why = []
        if (rc_r == 0) != (rc_o == 0):
            why.append("status %s / %s" % (rc_r, rc_o))
        # a clean exit with a status of its own (usage: 1) must keep that status
        if rc_r in (0, 1) and rc_r != rc_o:
            why.append("exit code %s / %s" % (rc_r, rc_o))
        if (out_r is None) != (out_o is None):
            why.append("output file %s / %s" % ("written" if out_r is not None else "none", "written" if out_o is not None else "none"))
        elif out_r is not None and tool == "nnet-copy" and out_r != out_o:
            why.append("output bytes differ (%d / %d)" % (len(out_r), len(out_o)))
        elif out_r is not None and tool != "nnet-copy" and (out_r[:2] == b"\0B") != (out_o[:2] == b"\0B"):
            why.append("output mode differs")
        elif out_r is not None and tool == "nnet-train-xent-hardlab-frmshuff":
            # the same option values understood: the trained weights agree
            la = [l for l in O.read_nnet(out_r) if isinstance(l, O.BiasedLinearity)]
            lb = [l for l in O.read_nnet(out_o) if isinstance(l, O.BiasedLinearity)]
            if len(la) != len(lb) or any(not np.allclose(x.W, y.W, rtol=2e-4, atol=2e-6) or not np.allclose(x.b, y.b, rtol=2e-4, atol=2e-6)
                                         for x, y in zip(la, lb)):
                why.append("trained weights differ")
        elif out_r is not None and tool == "nnet-forward":
            a, b = O.read_matrix_ark(out_r), O.read_matrix_ark(out_o)
            if [k for k, _ in a] != [k for k, _ in b] or any(not np.allclose(x, y, rtol=2e-5, atol=2e-6) for (_, x), (_, y) in zip(a, b)):
                why.append("outputs differ")
        if "--help" in argv or not argv:
            # the usage text: the same option names listed, on the same stream
            names_r = sorted(set(w.split()[0] for w in err_r.splitlines() if w.strip().startswith("--")))
            names_o = sorted(set(w.split()[0] for w in err_o.splitlines() if w.strip().startswith("--")))
            missing = [n for n in names_r if n not in names_o]
            if missing:
                why.append("usage lacks " + " ".join(missing))
            if bool(so_r.strip()) != bool(so_o.strip()):
                why.append("usage on stdout %s / %s" % (bool(so_r.strip()), bool(so_o.strip())))
        mark = "" if not why else "<<< DIFFERENT: " + "; ".join(why)
        differ += 1 if why else 0
        lines.append("%-44s reference rc %-8s ours rc %-8s %s" % (name, rc_r, rc_o, mark))
        if why or rc_r != 0:
            pick = lambda e: " | ".join([l for l in e.strip().splitlines() if "ERROR" in l or "ASSERT" in l or "rror" in l][:1] or e.strip().splitlines()[-1:])  # noqa: E731
            lines.append("    ref : " + pick(err_r)[:260])
            lines.append("    ours: " + pick(err_o)[:260])
        print(lines[-1] if not (why or rc_r != 0) else "\n".join(lines[-3:]), flush=True)
    lines.insert(0, "options_check: %d cases, %d differ" % (len(cases), differ))
    print(lines[0])
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")
    return 1 if differ else 0


if __name__ == "__main__":
    sys.exit(main())
