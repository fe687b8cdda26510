#!/usr/bin/env python3
"""Inputs at the edge of the formats, through nnet-forward and the trainer of this build and of the reference
(oracle/_ref/bin): utterances without rows, double-precision matrices, text archives with unusual spacing, nan / inf
in features, repeated and very long keys, empty alignments, frame-count vectors with zeros or of the wrong length,
text models in scientific notation.  Exit status and outputs must agree.

    python tests/fuzz/odd_inputs_check.py [--out FILE]
"""
import argparse
import os
import struct
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


def dm(m):
    m = np.asarray(m, np.float64)
    return b"DM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]) + m.astype("<f8").tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rs = np.random.RandomState(9)
    tmp = tempfile.mkdtemp(prefix="odd_")
    p = lambda n: os.path.join(tmp, n)  # noqa: E731
    layers = [O.BiasedLinearity((rs.standard_normal((16, 24)) / 5).astype(f32), (0.1 * rs.standard_normal(16)).astype(f32)), O.Sigmoid(16),
              O.BiasedLinearity((rs.standard_normal((11, 16)) / 4).astype(f32), (0.1 * rs.standard_normal(11)).astype(f32)), O.Softmax(11)]
    open(p("m.nnet"), "wb").write(O.write_nnet(layers, True))
    text = O.write_nnet(layers, False).decode()
    # the same model as text, numbers in scientific notation, tabs and blank lines between tokens
    sci = " ".join(("%.8e" % float(t)) if O_is_number(t) and "." in t else t for t in text.replace("\n", " \n ").split(" "))
    open(p("m_sci.nnet"), "w").write(sci)
    open(p("m_loose.nnet"), "w").write(sci.replace(" [", "\t[\n\n"))  # (a newline inside a vector is not allowed)
    utts = [("utt%d" % i, rs.standard_normal((n, 24)).astype(f32)) for i, n in enumerate([5, 70, 33])]
    alis = [(k, rs.randint(0, 11, m.shape[0]).astype(np.int32)) for k, m in utts]
    open(p("f.ark"), "wb").write(O.write_matrix_ark(utts))
    open(p("a.ark"), "wb").write(O.write_int32_vector_ark(alis))
    # an utterance without rows in the middle (binary and text)
    empty = ("utt0b", np.zeros((0, 24), f32))
    open(p("f_empty_row.ark"), "wb").write(O.write_matrix_ark([utts[0], empty, utts[1]]))
    open(p("f_empty_row.txt"), "wb").write(O.write_matrix_ark([utts[0]], False) + b"utt0b  [ ]\n" + O.write_matrix_ark([utts[1]], False))
    open(p("f_double.ark"), "wb").write(b"".join(k.encode() + b" \0B" + dm(m) for k, m in utts))
    nan = utts[1][1].copy(); nan[3, 4] = np.nan
    inf = utts[1][1].copy(); inf[3, 4] = np.inf
    open(p("f_nan.ark"), "wb").write(O.write_matrix_ark([utts[0], ("utt1", nan), utts[2]]))
    open(p("f_inf.ark"), "wb").write(O.write_matrix_ark([utts[0], ("utt1", inf), utts[2]]))
    open(p("f_nan.txt"), "wb").write(O.write_matrix_ark([utts[0]], False) + b"utt1  [\n  " + b" ".join([b"nan"] * 24) + b" ]\n")
    open(p("f_dupkey.ark"), "wb").write(O.write_matrix_ark([utts[0], utts[0], utts[1]]))
    long_key = "k" * 300
    open(p("f_longkey.ark"), "wb").write(O.write_matrix_ark([(long_key, utts[0][1])]))
    open(p("a_longkey.ark"), "wb").write(O.write_int32_vector_ark([(long_key, alis[0][1])]))
    # text archive: one row per line without the leading spaces, "]" on its own line, Windows line ends
    t = "utt0 [\n" + "\n".join(" ".join("%g" % v for v in r) for r in utts[0][1]) + "\n]\n"
    open(p("f_loose.txt"), "w").write(t)
    open(p("f_crlf.txt"), "w").write(t.replace("\n", "\r\n"))
    open(p("f_ragged.txt"), "w").write("utt0 [\n 1 2 3\n 4 5\n]\n")
    open(p("a_empty_vec.ark"), "wb").write(O.write_int32_vector_ark([alis[0], ("utt1", np.zeros(0, np.int32)), alis[2]]))
    open(p("a_text_loose.txt"), "w").write("".join("%s   %s  \n" % (k, "  ".join(str(int(x)) for x in v)) for k, v in alis))
    open(p("a_float.txt"), "w").write("".join("%s %s\n" % (k, " ".join("%d.0" % int(x) for x in v)) for k, v in alis))
    open(p("a_negative.ark"), "wb").write(O.write_int32_vector_ark([(k, np.where(np.arange(len(v)) == 2, -1, v)) for k, v in alis]))
    open(p("a_too_big.ark"), "wb").write(O.write_int32_vector_ark([(k, np.where(np.arange(len(v)) == 2, 11, v)) for k, v in alis]))
    counts = rs.randint(1, 1000, 11).astype(np.float64)
    open(p("counts.txt"), "wb").write(O.write_vector(counts, False))
    open(p("counts.bin"), "wb").write(b"\0B" + O.write_vector(counts.astype(f32), True))
    z = counts.copy(); z[3] = 0
    open(p("counts_zero.txt"), "wb").write(O.write_vector(z, False))
    open(p("counts_short.txt"), "wb").write(O.write_vector(counts[:10], False))
    open(p("counts_int.txt"), "w").write(" [ " + " ".join(str(int(c)) for c in counts) + " ]\n")

    F = lambda feats, *opts: ("nnet-forward", list(opts) + [p("m.nnet"), feats, "ark:{out}"])  # noqa: E731
    T = lambda feats, ali, *opts: ("nnet-train-xent-hardlab-frmshuff", ["--bunchsize=8", "--cachesize=256", "--randomize=false"] + list(opts) +  # noqa: E731
                                   [feats, ali, p("m.nnet"), "{out}"])
    cases = [
        ("forward: utterance without rows (binary)",) + F("ark:" + p("f_empty_row.ark")),
        ("forward: utterance without rows (text)",) + F("ark,t:" + p("f_empty_row.txt")),
        ("forward: double-precision features",) + F("ark:" + p("f_double.ark")),
        ("forward: nan in features",) + F("ark:" + p("f_nan.ark")),
        ("forward: inf in features",) + F("ark:" + p("f_inf.ark")),
        ("forward: nan written as text",) + F("ark,t:" + p("f_nan.txt")),
        ("forward: repeated key",) + F("ark:" + p("f_dupkey.ark")),
        ("forward: key of 300 characters",) + F("ark:" + p("f_longkey.ark")),
        ("forward: text rows without indent",) + F("ark,t:" + p("f_loose.txt")),
        ("forward: text with CR LF",) + F("ark,t:" + p("f_crlf.txt")),
        ("forward: ragged text matrix",) + F("ark,t:" + p("f_ragged.txt")),
        ("forward: text model, scientific notation", "nnet-forward", [p("m_sci.nnet"), "ark:" + p("f.ark"), "ark:{out}"]),
        ("forward: text model, blank lines after [", "nnet-forward", [p("m_loose.nnet"), "ark:" + p("f.ark"), "ark:{out}"]),
        ("forward: counts text", ) + F("ark:" + p("f.ark"), "--class-frame-counts=" + p("counts.txt")),
        ("forward: counts binary",) + F("ark:" + p("f.ark"), "--class-frame-counts=" + p("counts.bin")),
        ("forward: counts with a zero",) + F("ark:" + p("f.ark"), "--class-frame-counts=" + p("counts_zero.txt")),
        ("forward: counts too short",) + F("ark:" + p("f.ark"), "--class-frame-counts=" + p("counts_short.txt")),
        ("forward: counts as integers",) + F("ark:" + p("f.ark"), "--class-frame-counts=" + p("counts_int.txt")),
        ("forward: counts, no-softmax, apply-log",) + F("ark:" + p("f.ark"), "--class-frame-counts=" + p("counts.txt"), "--no-softmax=true", "--apply-log=true"),
        ("train: utterance without rows",) + T("ark:" + p("f_empty_row.ark"), "ark:" + p("a.ark")),
        ("train: double-precision features",) + T("ark:" + p("f_double.ark"), "ark:" + p("a.ark")),
        ("train: nan in features",) + T("ark:" + p("f_nan.ark"), "ark:" + p("a.ark")),
        ("train: repeated feature key",) + T("ark:" + p("f_dupkey.ark"), "ark:" + p("a.ark")),
        ("train: key of 300 characters",) + T("ark:" + p("f_longkey.ark"), "ark:" + p("a_longkey.ark")),
        ("train: an empty alignment",) + T("ark:" + p("f.ark"), "ark:" + p("a_empty_vec.ark")),
        ("train: text alignments, loose spacing",) + T("ark:" + p("f.ark"), "ark,t:" + p("a_text_loose.txt")),
        ("train: text alignments written as 3.0",) + T("ark:" + p("f.ark"), "ark,t:" + p("a_float.txt")),
        ("train: text model, scientific notation", "nnet-train-xent-hardlab-frmshuff",
         ["--bunchsize=8", "--cachesize=256", "--randomize=false", "ark:" + p("f.ark"), "ark:" + p("a.ark"), p("m_sci.nnet"), "{out}"]),
        ("cv: a label of -1",) + T("ark:" + p("f.ark"), "ark:" + p("a_negative.ark"), "--cross-validate=true")[:1] +
        (["--bunchsize=8", "--cachesize=256", "--cross-validate=true", "ark:" + p("f.ark"), "ark:" + p("a_negative.ark"), p("m.nnet")],),
    ]
    # Where the two are known to part, and why (everything else must agree):
    known = {
        "forward: utterance without rows (binary)": "the reference's Matrix::Read asserts on a 0 x 24 matrix (kaldi-matrix.cc:590); this build writes an output without rows",
        "forward: utterance without rows (text)": "read as 0 x 0 by the reference, whose first component then refuses it (nnet-component.h:254); this build writes an output without rows",
        "train: utterance without rows": "as above (kaldi-matrix.cc:590); this build has no alignment of that length and skips the utterance",
        "train: an empty alignment": "the reference never returns from an alignment of the wrong length (nnet-train-xent-hardlab-frmshuff.cc:139-144); this build skips the utterance",
        "cv: a label of -1": "the reference's CPU branch indexes the posterior row with it (cu-math-inl.h:373-377: memory in front of the row); this build refuses the alignment",
    }
    lines, differ = [], 0
    for name, tool, argv in cases:
        res = []
        for which, d in (("ref", REF), ("ours", OURS)):
            out = p("out_" + which)
            if os.path.exists(out):
                os.remove(out)
            a = [x.replace("{out}", out) for x in argv]
            try:
                r = subprocess.run([os.path.join(d, tool)] + a, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=100)
                rc, err = r.returncode, r.stderr.decode(errors="replace")
            except subprocess.TimeoutExpired:
                rc, err = "timeout", ""
            res.append((rc, err, open(out, "rb").read() if os.path.exists(out) else None))
        (rc_r, err_r, out_r), (rc_o, err_o, out_o) = res
        why = []
        if (rc_r == 0) != (rc_o == 0):
            why.append("status %s / %s" % (rc_r, rc_o))
        elif rc_r == 0 and out_r is not None and out_o is not None:
            if tool == "nnet-forward":
                a, b = O.read_matrix_ark(out_r), O.read_matrix_ark(out_o)
                if [k for k, _ in a] != [k for k, _ in b]:
                    why.append("keys %s / %s" % ([k[:10] for k, _ in a], [k[:10] for k, _ in b]))
                elif any(x.shape != y.shape or not np.allclose(x, y, rtol=3e-5, atol=2e-6, equal_nan=True) for (_, x), (_, y) in zip(a, b)):
                    why.append("outputs differ")
            else:
                la = [l for l in O.read_nnet(out_r) if isinstance(l, O.BiasedLinearity)]
                lb = [l for l in O.read_nnet(out_o) if isinstance(l, O.BiasedLinearity)]
                if any(not np.allclose(x.W, y.W, rtol=2e-4, atol=2e-6, equal_nan=True) for x, y in zip(la, lb)):
                    why.append("trained weights differ")
        elif rc_r == 0 and (out_r is None) != (out_o is None):
            why.append("output file %s / %s" % (out_r is not None, out_o is not None))
        if rc_r == 0 and rc_o == 0:
            def frames(e):
                import re
                m = re.findall(r"frames:(\d+)", e)
                return m[-1] if m else None
            if frames(err_r) != frames(err_o):
                why.append("frames %s / %s" % (frames(err_r), frames(err_o)))
        if why and name in known:
            lines.append("%-46s reference rc %-8s ours rc %-8s known: %s" % (name, rc_r, rc_o, known[name]))
            why = []
        else:
            differ += 1 if why else 0
            lines.append("%-46s reference rc %-8s ours rc %-8s %s" % (name, rc_r, rc_o, ("<<< DIFFERENT: " + "; ".join(why)) if why else ""))
        if why or rc_r != 0:
            pick = lambda e: " | ".join([l for l in e.strip().splitlines() if "ERROR" in l or "ASSERT" in l or "rror" in l or "WARN" in l][:2] or e.strip().splitlines()[-1:])  # noqa: E731
            lines.append("    ref : " + pick(err_r)[:300])
            lines.append("    ours: " + pick(err_o)[:300])
        print("\n".join(lines[-3:] if (why or rc_r != 0) else lines[-1:]), flush=True)
    lines.insert(0, "odd_inputs_check: %d cases, %d differ (%d known differences listed with their reason)" % (len(cases), differ, len(known)))
    print(lines[0])
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")
    return 1 if differ else 0


def O_is_number(t):
    try:
        float(t)
        return True
    except ValueError:
        return False


if __name__ == "__main__":
    sys.exit(main())
