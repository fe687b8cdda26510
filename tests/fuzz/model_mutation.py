"""Damaged model files through nnet-copy and rbm-convert-to-nnet of this build (on the GPU box): the tool ends with
status 0 or with a reported error; a signal (segmentation fault, bus error), a hang or a device fault is a finding.
The reference's own tools are run on the same files where they are present, for the record of how often the two agree
on accept / refuse (a damaged file that still parses may parse differently: not a finding).

usage: python tests/fuzz/model_mutation.py [--cases N] [--seed S] [--out file]"""
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
from table_mutation import mutate  # noqa: E402
import nnet_walk as W  # noqa: E402

OURS = os.path.join(ROOT, "kaldi_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref", "bin")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    counts, findings, notes, agree, both, copied_both, same_bytes = {}, [], [], 0, 0, 0, 0
    have_ref = os.path.exists(os.path.join(REF, "nnet-copy"))
    with tempfile.TemporaryDirectory() as tmp:
        for case in range(args.cases):
            tool = "nnet-copy"
            if rs.rand() < 0.3:   # one <rbm> through the converter (nnetbin/rbm-convert-to-nnet.cc)
                tool = "rbm-convert-to-nnet"
                v, h = int(rs.choice([3, 7, 16, 33])), int(rs.choice([3, 7, 16, 33]))
                layers = [O.Rbm((0.1 * rs.standard_normal((h, v))).astype(np.float32), (0.1 * rs.standard_normal(v)).astype(np.float32),
                                (0.1 * rs.standard_normal(h)).astype(np.float32), "gauss" if rs.rand() < 0.5 else "bern", "bern")]
                desc = "rbm %dx%d" % (v, h)
            else:
                layers, desc = W.random_model(rs)
            if not layers:
                continue
            binary = bool(rs.rand() < 0.5)
            data = mutate(rs, O.write_nnet(layers, binary))
            f = os.path.join(tmp, "m.nnet")
            open(f, "wb").write(data)
            res = {}
            for which, d in (("ours", OURS), ("ref", REF)):
                if which == "ref" and not have_ref:
                    continue
                if os.path.exists(os.path.join(tmp, "o_" + which)):
                    os.remove(os.path.join(tmp, "o_" + which))
                try:
                    r = subprocess.run([os.path.join(d, tool), "--binary=true", f, os.path.join(tmp, "o_" + which)],
                                       capture_output=True, timeout=100)
                    res[which] = r.returncode
                except subprocess.TimeoutExpired:
                    res[which] = "timeout"
            rc = res["ours"]
            kind = "copied" if rc == 0 else "reported error" if rc in (1, 255) else "library assertion" if rc == -6 else "status %s" % rc
            counts[kind] = counts.get(kind, 0) + 1
            if kind.startswith("status"):
                findings.append("case %d (%s, %s): %s" % (case, desc, "binary" if binary else "text", kind))
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "model_mutation_case%d.nnet" % case), "wb").write(data)
            if "ref" in res:
                both += 1
                agree += (res["ref"] == 0) == (rc == 0)
                if res["ref"] == 0 and rc == 0:
                    copied_both += 1
                    same = open(os.path.join(tmp, "o_ref"), "rb").read() == open(os.path.join(tmp, "o_ours"), "rb").read()
                    same_bytes += same
                    if not same:
                        notes.append("  case %d (%s, %s): both copy, the copies differ" % (case, desc, "binary" if binary else "text"))
                        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                        open(os.path.join(ROOT, "gpurun_out", "model_mutation_differ%d.nnet" % case), "wb").write(data)
                if (res["ref"] == 0) != (rc == 0):
                    notes.append("  case %d (%s, %s): reference %s, this build %s" % (case, desc, "binary" if binary else "text", res["ref"], rc))
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    open(os.path.join(ROOT, "gpurun_out", "model_mutation_disagree%d.nnet" % case), "wb").write(data)
    lines = ["model_mutation: %d damaged models, %d findings" % (sum(counts.values()), len(findings))]
    lines += ["  %-20s %d" % kv for kv in sorted(counts.items())]
    if both:
        lines.append("  accept / refuse as the reference's tool does: %d of %d" % (agree, both))
        lines.append("  copied by both: %d, the same bytes: %d" % (copied_both, same_bytes))
        lines += notes
    lines += findings
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
