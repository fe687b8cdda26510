#!/usr/bin/env python3
"""nnet-forward of this build and of the reference (oracle/_ref/bin, CPU) through the table forms the recipes use --
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
print("%d combinations differ" % bad)
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
sys.exit(1 if bad else 0)
