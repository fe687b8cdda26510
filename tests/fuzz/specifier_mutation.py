"""Random table specifiers -- every combination of the option words (ark, scp, t, b, s, ns, cs, ncs, p, np, o, no, f,
nf), in any order and letter case, with words that do not exist, missing colons, the two-file forms, pipes -- as the
feature, alignment and output arguments of nnet-forward and the trainer of both builds (on the GPU box): the same
accept / refuse and, where both accept, the same output files (text forms compared after parsing).

usage: python tests/fuzz/specifier_mutation.py [--cases N] [--seed S] [--out file]"""
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
from table_mutation import text_ali_ark, binary_ali_ark  # noqa: E402

OURS = os.path.join(ROOT, "kaldi_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref", "bin")
f32 = np.float32
# (the words a reader knows, the words a writer knows, and now and then one from the other side or from nowhere)
RWORDS = ["t", "b", "s", "ns", "cs", "ncs", "p", "np", "o", "no"] * 4 + ["f", "nf", "bogus", "", "T", "S", "ark", "scp"]
WWORDS = ["t", "b", "f", "nf"] * 6 + ["s", "p", "o", "bogus", "", "T", "ark", "scp"]


def run(d, tool, argv):
    try:
        r = subprocess.run([os.path.join(d, tool)] + argv, capture_output=True, timeout=15 if d == REF else 100)
        return r.returncode, r.stderr.decode(errors="replace")
    except subprocess.TimeoutExpired:
        return "timeout", ""


def scp_of(ark, keys, path):
    pos, lines = 0, []
    for k in keys:
        pos = ark.index(k.encode() + b" ", pos) + len(k) + 1
        lines.append("%s %s:%d\n" % (k, path, pos))
    return "".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    D, H, C = 13, 24, 9
    layers = [O.BiasedLinearity((rs.randn(H, D) / 4).astype(f32), (0.1 * rs.randn(H)).astype(f32)), O.Sigmoid(H),
              O.BiasedLinearity((rs.randn(C, H) / 5).astype(f32), (0.1 * rs.randn(C)).astype(f32)), O.Softmax(C)]
    utts = [("utt%d" % i, rs.randn(int(rs.randint(5, 30)), D).astype(f32)) for i in range(5)]
    alis = [(k, rs.randint(0, C, m.shape[0])) for k, m in utts]
    keys = [k for k, _ in utts]
    counts, differ = {}, []
    pick = lambda seq: seq[int(rs.randint(0, len(seq)))]  # noqa: E731
    with tempfile.TemporaryDirectory() as tmp:
        P = lambda n: os.path.join(tmp, n)  # noqa: E731
        open(P("m.nnet"), "wb").write(O.write_nnet(layers, True))
        fb, ft = O.write_matrix_ark(utts), O.write_matrix_ark(utts, binary=False)
        ab, at = binary_ali_ark(alis), text_ali_ark(alis)
        for n, d in (("f.ark", fb), ("f.txt", ft), ("a.ark", ab), ("a.txt", at)):
            open(P(n), "wb").write(d)
        open(P("f.scp"), "w").write(scp_of(fb, keys, P("f.ark")))
        open(P("a.scp"), "w").write(scp_of(ab, keys, P("a.ark")))

        def rspec(stem):
            """a read specifier for the features (stem f) or the alignments (stem a)"""
            kind = pick(["ark", "ark", "scp", "ark", "scp", "ark", "ark", pick(["ARK", "Ark", "sc", "", "arc"])])
            words = [kind] + [pick(RWORDS) for _ in range(int(rs.randint(0, 4)))]
            if rs.rand() < 0.3:
                rs.shuffle(words)
            if kind.lower().startswith("sc"):
                target = P(stem + ".scp")
            else:
                target = P(stem + (".txt" if rs.rand() < 0.4 else ".ark"))
            if rs.rand() < 0.1:
                target = "cat %s |" % target
            if rs.rand() < 0.04:
                return ",".join(words) + target                # no colon
            if rs.rand() < 0.04:
                target = P("missing")
            return ",".join(words) + ":" + target

        def wspec(who):
            kind = pick(["ark", "ark", "scp", "ark,scp", "scp,ark", "ark", "ark,scp", "ark", pick(["ARK", "", "arc"])])
            words = kind.split(",") + [pick(WWORDS) for _ in range(int(rs.randint(0, 3)))]
            if rs.rand() < 0.2 and "," not in kind:
                rs.shuffle(words)
            files = {"ark": "o_%s.ark", "scp": "o_%s.scp"}
            if kind in ("ark,scp", "scp,ark"):
                target = ",".join(P(files[w]) for w in kind.split(","))
                if rs.rand() < 0.15:
                    target = P(files["ark"])                 # one file for the two-file form
            elif kind == "scp":
                target = P(files["scp"])
            else:
                target = P(files["ark"])
            if rs.rand() < 0.08 and "," not in kind:
                target = "| cat > " + target
            return ",".join(words) + ":" + target

        for case in range(args.cases):
            forward = rs.rand() < 0.6
            frs, ars, ws = rspec("f"), rspec("a"), wspec("%s")
            res = {}
            for which, d in (("ref", REF), ("ours", OURS)):
                for n in ("o_%s.ark" % which, "o_%s.scp" % which, "o_%s.nnet" % which):
                    if os.path.exists(P(n)):
                        os.remove(P(n))
                if forward:
                    argv = [P("m.nnet"), frs, ws.replace("%s", which)]
                    rc, err = run(d, "nnet-forward", argv)
                else:
                    argv = ["--bunchsize=8", "--cachesize=64", "--randomize=false", "--binary=true", frs, ars, P("m.nnet"), P("o_%s.nnet" % which)]
                    rc, err = run(d, "nnet-train-xent-hardlab-frmshuff", argv)
                res[which] = (rc, err)
            (rc_r, err_r), (rc_o, err_o) = res["ref"], res["ours"]
            kind = "both accept" if rc_r == 0 and rc_o == 0 else "both refuse" if rc_r != 0 and rc_o != 0 else "one refuses"
            why = []
            if rc_o not in (0, 1, 255, -6):
                why.append("this build ends with %s" % rc_o)
            if rc_r == "timeout":
                kind = "the reference does not end"
            elif kind == "one refuses":
                why.append("status %s / %s" % (rc_r, rc_o))
            elif kind == "both accept" and forward:
                for ext in ("ark", "scp"):
                    fr, fo = P("o_ref." + ext), P("o_ours." + ext)
                    if os.path.exists(fr) != os.path.exists(fo):
                        why.append("%s file written %s / %s" % (ext, os.path.exists(fr), os.path.exists(fo)))
                    elif os.path.exists(fr) and ext == "scp":
                        a = open(fr).read().replace("o_ref", "o_X")
                        b = open(fo).read().replace("o_ours", "o_X")
                        text_form = os.path.exists(P("o_ref.ark")) and b"\0B" not in open(P("o_ref.ark"), "rb").read()[:16]
                        if a != b and not text_form:
                            why.append("script files differ")
                        elif a != b:
                            # a text archive: the last digit of a value may differ and with it the length of a line;
                            # the keys must agree and every offset must stand behind "key " in its own archive
                            ka, kb = [l.split()[0] for l in a.splitlines()], [l.split()[0] for l in b.splitlines()]
                            arch = open(P("o_ours.ark"), "rb").read()
                            ok = ka == kb
                            for l in open(fo).read().splitlines():
                                k, loc = l.split()
                                off = int(loc.rsplit(":", 1)[1])
                                ok = ok and arch[off - len(k) - 1:off] == k.encode() + b" "
                            if not ok:
                                why.append("script files differ")
                    elif os.path.exists(fr):
                        da, db = open(fr, "rb").read(), open(fo, "rb").read()
                        if (b"\0B" in da[:16]) != (b"\0B" in db[:16]):
                            why.append("text / binary form of the archive differs")
                        else:
                            a, b = O.read_matrix_ark(da), O.read_matrix_ark(db)
                            if [k for k, _ in a] != [k for k, _ in b] or any(
                                    x.shape != y.shape or not np.allclose(x, y, rtol=1e-4, atol=1e-6) for (_, x), (_, y) in zip(a, b)):
                                why.append("outputs differ")
            elif kind == "both accept":
                la = [l for l in O.read_nnet(open(P("o_ref.nnet"), "rb").read()) if isinstance(l, O.BiasedLinearity)]
                lb = [l for l in O.read_nnet(open(P("o_ours.nnet"), "rb").read()) if isinstance(l, O.BiasedLinearity)]
                if any(not np.allclose(x.W, y.W, rtol=1e-3, atol=1e-5) for x, y in zip(la, lb)):
                    why.append("trained weights differ")
            counts[kind] = counts.get(kind, 0) + 1
            if case % 10 == 9:
                print("... %d cases, %d differ" % (case + 1, len(differ)), flush=True)
            if why:
                shown = ("nnet-forward m.nnet '%s' '%s'" % (frs, ws)) if forward else ("train '%s' '%s'" % (frs, ars))
                differ.append("case %d: %s\n    %s\n    ref : %s\n    ours: %s" % (
                    case, "; ".join(why), shown.replace(tmp + "/", ""),
                    err_r.strip().split("\n")[-1][:220] if err_r else "", err_o.strip().split("\n")[-1][:220] if err_o else ""))
    lines = ["specifier_mutation: %d cases, %d differ" % (args.cases, len(differ))]
    lines += ["  %-28s %d" % kv for kv in sorted(counts.items())] + differ
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
    return 1 if differ else 0


if __name__ == "__main__":
    sys.exit(main())
