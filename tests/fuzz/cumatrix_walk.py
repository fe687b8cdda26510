#!/usr/bin/env python3
"""Class-level differential check: the same random walk over the CuMatrix / CuVector / cu:: API
(tests/fuzz/cumatrix_walk.cc) run against the reference's classes (CPU branch, oracle/_ref/bin/cumatrix_walk) and
against this build's classes on the GPU (kaldi_amd/bin/cumatrix_walk); the two dump files are compared record by
record.  Shapes must be equal; values agree to float rounding (sums and products are ordered differently on the GPU),
row-maximum ids exactly.

usage: python tests/fuzz/cumatrix_walk.py [--seeds N] [--steps K] [--out FILE]
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "bin", "cumatrix_walk")
OURS = os.path.join(ROOT, "kaldi_amd", "bin", "cumatrix_walk")

OP_NAMES = {0: "SetZero", 1: "Set", 2: "PartSet", 3: "Binarize", 4: "Add", 5: "ApplyLog", 6: "ApplyExp", 7: "Power",
            8: "Scale", 9: "InvertElements", 10: "MulElements", 11: "DivElements", 12: "MulColsVec", 13: "MulRowsVec",
            14: "DivColsVec", 15: "DivRowsVec", 16: "AddMat", 17: "AddVecToCols", 18: "AddVecToRows",
            19: "AddVecToPartialRows", 20: "AddMatMat", 21: "LogAddExpMat", 22: "ApplyRowL2UpperBound",
            23: "CopyRowsFromMat", 24: "Resize+CopyFromMat", 25: "AddRowSumMat", 26: "AddColSumMat",
            27: "CuVector AddVec/Scale/Add/ApplyFloor/ApplyTruncate/MulElements/DivElements/Round",
            28: "Sigmoid+DiffSigmoid", 29: "Softmax+FindRowMaxId", 30: "DiffXent", 31: "Randomize", 32: "Expand+Copy",
            33: "RegularizeL1", 34: "PartAddMat", 35: "CopyFromMat(window)+CopyToMat(window)", 36: "Relu+DiffRelu",
            37: "SoftRelu+DiffSoftRelu", 38: "CuVector Power/ApplyLog/ApplyExp/InvertElements/Set/Resize/CopyFromVec",
            39: "CopyToMat(window past the end)", 40: "CuStlVector Set/CopyFromVec/CopyToVec/SetZero/Resize/<<",
            41: "CuMatrix Write/Read/<< (text, binary)", 42: "CuVector Write/Read/<< (text, binary)",
            43: "sizes in bytes, Destroy, use after Destroy"}


def read_dump(path):
    recs = []
    with open(path, "rb") as f:
        while True:
            head = f.readline()
            if not head:
                break
            tag, r, c = head.decode().split()
            r, c = int(r), int(c)
            a = np.frombuffer(f.read(4 * r * c), dtype=np.float32).reshape(r, c)
            assert f.read(1) == b"\n", "dump out of step at " + tag
            recs.append((tag, a))
    return recs


def run(binary, seed, steps, path):
    p = subprocess.run([binary, str(seed), str(steps), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    if p.returncode != 0:
        raise RuntimeError("%s seed %d: exit %d\n%s" % (binary, seed, p.returncode, p.stderr.decode()[-2000:]))


def compare(seed, ref, ours, counts, bad):
    i = j = 0
    while i < len(ref) and j < len(ours):
        tag, a = ref[i]
        tag2, b = ours[j]
        if tag != tag2:
            bad.append("seed %d: record order differs: %s / %s" % (seed, tag, tag2))
            return
        op = int(tag.split("_op")[1].split("_")[0])
        counts[op] = counts.get(op, 0) + 1
        if op == 7:
            # the reference's CPU branch has no element-wise matrix Power (it dumps the input only): this build's
            # second record is checked against input ** 1.5
            want = a.astype(np.float64) ** 1.5
            if not np.array_equal(a, b):
                bad.append("seed %d %s: inputs differ" % (seed, tag))
            got = ours[j + 1][1]
            if got.shape != want.shape or not np.allclose(got, want, rtol=2e-6, atol=0):
                bad.append("seed %d %s (Power): differs from input^1.5" % (seed, tag))
            i, j = i + 1, j + 2
            continue
        if a.shape != b.shape:
            bad.append("seed %d %s (%s): shape %s / %s" % (seed, tag, OP_NAMES[op], a.shape, b.shape))
        elif tag.endswith("_ids"):
            if not np.array_equal(a, b):
                bad.append("seed %d %s: row-maximum ids differ" % (seed, tag))
        else:
            # float32 sums of up to 257 terms in a different order; everything else a few ulp
            rtol, atol = (3e-5, 3e-5) if op in (20, 22, 25, 26, 29, 30) else (4e-6, 1e-6)
            if not np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True):
                d = np.abs(a.astype(np.float64) - b)
                k = np.unravel_index(np.nanargmax(d), d.shape)
                bad.append("seed %d %s (%s): max |diff| %.3g at %s: ref %.9g ours %.9g" %
                           (seed, tag, OP_NAMES[op], d[k], k, a[k], b[k]))
        i, j = i + 1, j + 1
    if i != len(ref) or j != len(ours):
        bad.append("seed %d: %d / %d records" % (seed, len(ref), len(ours)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--first-seed", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    for b in (REF, OURS):
        if not os.path.exists(b):
            sys.exit("missing " + b + " (make -C oracle ref; make -C kaldi_amd/csrc host)")
    counts, bad, records = {}, [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(args.first_seed, args.first_seed + args.seeds):
            fr, fo = os.path.join(tmp, "ref.bin"), os.path.join(tmp, "ours.bin")
            run(REF, seed, args.steps, fr)
            run(OURS, seed, args.steps, fo)
            ref, ours = read_dump(fr), read_dump(fo)
            records += len(ref)
            compare(seed, ref, ours, counts, bad)
            print("seed %d: %d records, %d differences so far" % (seed, len(ref), len(bad)), flush=True)
    lines = ["cumatrix_walk: %d seeds x %d steps, %d records compared, %d differ" %
             (args.seeds, args.steps, records, len(bad))]
    lines += ["  %-28s %d records" % (OP_NAMES[k][:28], counts[k]) for k in sorted(counts)]
    lines += ["DIFF " + b for b in bad[:50]]
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
