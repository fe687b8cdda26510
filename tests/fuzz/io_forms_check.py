#!/usr/bin/env python3
"""nnet-forward (and, for the alignment table, the trainer) of this build and of the reference (oracle/_ref/bin, CPU) through the table forms the recipes use --
archives and script files, text and binary, pipes on both sides: the outputs must agree (values of text forms are
compared after parsing).  A diagnostic run on the GPU box.    python tests/fuzz/io_forms_check.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32
REF, OURS = os.path.join(ROOT, "oracle", "_ref", "bin"), os.path.join(ROOT, "kaldi_amd", "bin")
tmp = tempfile.mkdtemp(prefix="io_forms_")
rs = np.random.RandomState(5)
dims = [24, 40, 11]
layers = [O.BiasedLinearity((0.3 * rs.standard_normal((40, 24))).astype(f32), (0.1 * rs.standard_normal(40)).astype(f32)), O.Sigmoid(40),
          O.BiasedLinearity((0.3 * rs.standard_normal((11, 40))).astype(f32), (0.1 * rs.standard_normal(11)).astype(f32)), O.Softmax(11)]
open(tmp + "/m.nnet", "wb").write(O.write_nnet(layers, True))
utts = [("utt%d" % i, rs.standard_normal((n, 24)).astype(f32)) for i, n in enumerate([5, 1, 33, 12])]
ark = O.write_matrix_ark(utts)
open(tmp + "/f.ark", "wb").write(ark)
open(tmp + "/f.txt", "wb").write(O.write_matrix_ark(utts, binary=False))
pos, lines = 0, []
for k, m in utts:
    pos = ark.index(k.encode() + b" \0B", pos) + len(k) + 1
    lines.append("%s %s/f.ark:%d\n" % (k, tmp, pos))
open(tmp + "/f.scp", "w").write("".join(lines))
inputs = ["ark:%s/f.ark" % tmp, "ark,t:%s/f.txt" % tmp, "scp:%s/f.scp" % tmp, "ark:cat %s/f.ark |" % tmp, "ark,s,cs:%s/f.ark" % tmp]
outputs = [("ark:%s/o_WHO.ark", ["o_WHO.ark"]), ("ark,t:%s/o_WHO.txt", ["o_WHO.txt"]),
           ("ark,scp:%s/o_WHO.ark,%s/o_WHO.scp", ["o_WHO.ark", "o_WHO.scp"]), ("ark:| cat > %s/o_WHO.ark", ["o_WHO.ark"]),
           ("ark,t:| gzip -c > %s/o_WHO.txt.gz", ["o_WHO.txt.gz"])]
bad = 0
for rspec in inputs:
    for wtempl, files in outputs:
        got = {}
        for who, d in (("ref", REF), ("ours", OURS)):
            w = wtempl.replace("WHO", who)
            w = w % tuple([tmp] * w.count("%s"))
            r = subprocess.run([os.path.join(d, "nnet-forward"), tmp + "/m.nnet", rspec, w], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=100)
            got[who] = (r.returncode, r.stderr.decode(errors="replace"), [tmp + "/" + f.replace("WHO", who) for f in files])
        msg = "ok"
        if got["ref"][0] != got["ours"][0]:
            msg = "exit codes differ: %d %d\n%s" % (got["ref"][0], got["ours"][0], got["ours"][1][-600:])
        elif got["ref"][0] == 0:
            for fr, fo in zip(got["ref"][2], got["ours"][2]):
                a, b = open(fr, "rb").read(), open(fo, "rb").read()
                if fr.endswith(".gz"):
                    import gzip
                    a, b = gzip.decompress(a), gzip.decompress(b)
                if fr.endswith(".scp"):
                    a, b = a.replace(b"o_ref", b"o_X"), b.replace(b"o_ours", b"o_X")
                    if a != b:
                        msg = "script files differ:\n%s\n%s" % (a.decode(), b.decode())
                elif fr.endswith(".ark"):
                    x, y = O.read_matrix_ark(a), O.read_matrix_ark(b)
                    if [k for k, _ in x] != [k for k, _ in y] or not all(np.allclose(u, v, rtol=1e-4, atol=1e-7) for (_, u), (_, v) in zip(x, y)) or len(a) != len(b):
                        msg = "archives differ (%d / %d bytes)" % (len(a), len(b))
                else:
                    ta, tb = a.decode().split(), b.decode().split()
                    if len(ta) != len(tb):
                        msg = "text archives differ in shape: %d / %d tokens" % (len(ta), len(tb))
                    else:
                        for u, v in zip(ta, tb):
                            try:
                                if abs(float(u) - float(v)) > 1e-4 * max(abs(float(u)), 1e-3):
                                    msg = "text values differ: %s %s" % (u, v)
                            except ValueError:
                                if u != v:
                                    msg = "text tokens differ: %s %s" % (u, v)
        if msg != "ok":
            bad += 1
        print("%-40s -> %-45s %s" % (rspec.replace(tmp, "."), wtempl.replace("%s", "."), msg), flush=True)
# ---- the alignment table of the trainer: a random-access read in the order of the feature table -----------------
# (RandomAccessInt32VectorReader, nnet-train-xent-hardlab-frmshuff.cc:96,132-144): archives in another order than the
# features, with and without the "sorted" / "called in sorted order" / "once" claims, script files, text form, a key
# that is missing, a key that comes twice
alis = [(k, rs.randint(0, 11, m.shape[0]).astype(np.int32)) for k, m in utts]
order = {"same": alis, "reversed": alis[::-1], "shuffled": [alis[2], alis[0], alis[3], alis[1]],
         "one missing": [alis[0], alis[1], alis[3]], "one twice": [alis[0], alis[1], alis[1], alis[2], alis[3]],
         "extra keys": [("aaa", alis[0][1])] + alis + [("zzz", alis[1][1])]}
ali_cases = []
for oname, a in order.items():
    stem = tmp + "/a_" + oname.replace(" ", "_")
    data = O.write_int32_vector_ark(a)
    open(stem + ".ark", "wb").write(data)
    open(stem + ".txt", "wb").write(O.write_int32_vector_ark(a, binary=False))
    lines, pos = [], 0
    for k, _ in a:
        pos = data.index(k.encode() + b" \0B", pos) + len(k) + 1
        lines.append("%s %s.ark:%d\n" % (k, stem, pos))
    open(stem + ".scp", "w").write("".join(lines))
    for form in ("ark:%s.ark", "ark,t:%s.txt", "scp:%s.scp", "ark,s:%s.ark", "ark,cs:%s.ark", "ark,s,cs:%s.ark", "ark,o:%s.ark",
                 "ark,p:%s.ark", "scp,p:%s.scp", "ark:cat %s.ark |"):
        ali_cases.append((oname, form % stem, "f.ark"))
# the features in reversed order: the alignment reader is then not called in sorted order
open(tmp + "/f_rev.ark", "wb").write(O.write_matrix_ark(utts[::-1]))
for form in ("ark:%s.ark", "ark,s:%s.ark", "ark,cs:%s.ark", "ark,s,cs:%s.ark", "scp,s,cs:%s.scp"):
    ali_cases.append(("same, feats reversed", form % (tmp + "/a_same"), "f_rev.ark"))
for oname, aspec, feats in ali_cases:
    got = {}
    for who, d in (("ref", REF), ("ours", OURS)):
        out = "%s/trained_%s.nnet" % (tmp, who)
        if os.path.exists(out):
            os.remove(out)
        try:
            r = subprocess.run([os.path.join(d, "nnet-train-xent-hardlab-frmshuff"), "--bunchsize=4", "--cachesize=64", "--learn-rate=0.01",
                                "--randomize=false", "ark:" + tmp + "/" + feats, aspec, tmp + "/m.nnet", out],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
            got[who] = (r.returncode, r.stderr.decode(errors="replace"), open(out, "rb").read() if os.path.exists(out) else None)
        except subprocess.TimeoutExpired:
            got[who] = ("timeout", "", None)
    msg = "ok"
    (rc_r, err_r, m_r), (rc_o, err_o, m_o) = got["ref"], got["ours"]
    if rc_r == "timeout" and rc_o != "timeout":
        msg = "ok (the reference never returns; this build rc %s)" % rc_o
    elif (rc_r == 0) != (rc_o == 0):
        msg = "status differs: %s / %s\n    ref : %s\n    ours: %s" % (rc_r, rc_o, " | ".join(err_r.strip().splitlines()[-2:])[:300],
                                                                      " | ".join(err_o.strip().splitlines()[-2:])[:300])
    elif rc_r == 0:
        la = [l for l in O.read_nnet(m_r) if isinstance(l, O.BiasedLinearity)]
        lb = [l for l in O.read_nnet(m_o) if isinstance(l, O.BiasedLinearity)]
        if any(not np.allclose(x.W, y.W, rtol=2e-4, atol=2e-6) for x, y in zip(la, lb)):
            msg = "trained weights differ"
        else:
            def counts(e):
                return [l for l in e.splitlines() if "no alignment" in l.lower() or "missing" in l.lower() or "done" in l.lower()][-1:]
            if counts(err_r) and counts(err_o) and counts(err_r)[0].split(")")[-1].split() != counts(err_o)[0].split(")")[-1].split():
                msg = "summary lines differ:\n    ref : %s\n    ours: %s" % (counts(err_r)[0], counts(err_o)[0])
    if not msg.startswith("ok"):
        bad += 1
    print("alignments %-20s %-34s rc %s / %s  %s" % (oname, aspec.replace(tmp, "."), rc_r, rc_o, msg), flush=True)
print("%d combinations differ" % bad)
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
sys.exit(1 if bad else 0)
