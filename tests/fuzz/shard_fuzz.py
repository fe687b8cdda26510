#!/usr/bin/env python3
"""Random check of the utterance-sharded nnet-forward on one GPU: N rank processes (2 to 8) over random tables
(1 to 40 utterances of 1 to 900 frames) read as an archive, through a script file or from a pipe; the ranks'
archives together must be the single-process archive bit for bit, every utterance on the rank the documented
rule gives it, and no rank may read another rank's payload.     python tests/fuzz/shard_fuzz.py [--cases 12] [--seed 1]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import kaldi_amd.nnet as nn  # noqa: E402
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32
EXE = os.path.join(ROOT, "kaldi_amd", "bin", "nnet-forward")


def one_case(rs, tmp):
    dims = [int(rs.choice([13, 40, 120])), int(rs.choice([32, 100, 256])), int(rs.choice([9, 50, 201]))]
    layers = [O.BiasedLinearity((0.3 * rs.standard_normal((dims[1], dims[0]))).astype(f32), np.zeros(dims[1], f32)), O.Sigmoid(dims[1]),
              O.BiasedLinearity((0.3 * rs.standard_normal((dims[2], dims[1]))).astype(f32), np.zeros(dims[2], f32)), O.Softmax(dims[2])]
    open(tmp + "/m.nnet", "wb").write(O.write_nnet(layers, True))
    lens = [int(rs.choice([1, 2, 9, 64, 130, 300, 900])) for _ in range(rs.randint(1, 41))]
    utts = [("utt%03d" % i, rs.standard_normal((n, dims[0])).astype(f32)) for i, n in enumerate(lens)]
    ark = O.write_matrix_ark(utts)
    open(tmp + "/f.ark", "wb").write(ark)
    pos, lines = 0, []
    for k, m in utts:
        pos = ark.index(k.encode() + b" \0B", pos) + len(k) + 1
        lines.append("%s %s/f.ark:%d\n" % (k, tmp, pos))
    open(tmp + "/f.scp", "w").write("".join(lines))
    form = str(rs.choice(["ark", "scp", "pipe"]))
    rspec = {"ark": "ark:%s/f.ark" % tmp, "scp": "scp:%s/f.scp" % tmp, "pipe": "ark:cat %s/f.ark |" % tmp}[form]
    N = int(rs.choice([2, 3, 5, 8]))
    bf = int(rs.choice([0, 512, 4096]))
    tag = "%d utterances, %d ranks, %s, --batch-frames=%d" % (len(lens), N, form, bf)
    r = subprocess.run([EXE, "--batch-frames=%d" % bf, tmp + "/m.nnet", rspec, "ark:%s/single.ark" % tmp], stderr=subprocess.PIPE, timeout=100)
    if r.returncode != 0:
        return tag, "single process failed: " + r.stderr.decode()[-500:]
    single = O.read_matrix_ark(open(tmp + "/single.ark", "rb").read())
    procs = [subprocess.Popen([EXE, "--batch-frames=%d" % bf, "--num-ranks=%d" % N, "--rank=%d" % k, "--gpu-id=0", tmp + "/m.nnet", rspec,
                               "ark:%s/shard.JOB.ark" % tmp], stderr=subprocess.PIPE) for k in range(N)]
    read = []
    for p in procs:
        _, err = p.communicate(timeout=100)
        if p.returncode != 0:
            return tag, "a rank failed: " + err.decode()[-500:]
        read.append(int(re.search(r"feature payload read (\d+) bytes", err.decode()).group(1)))
    want = nn.shard_assign(N, lens)
    shards = [dict(O.read_matrix_ark(open("%s/shard.%d.ark" % (tmp, k + 1), "rb").read())) for k in range(N)]
    if sum(len(s) for s in shards) != len(single):
        return tag, "the shards hold %d utterances, the table %d" % (sum(len(s) for s in shards), len(single))
    for i, (k, m) in enumerate(single):
        if k not in shards[want[i]]:
            return tag, "%s is not on rank %d" % (k, want[i])
        if not np.array_equal(shards[want[i]][k], m):
            return tag, "%s differs between the sharded and the single run" % k
    payload = [n * dims[0] * 4 for n in lens]
    for k in range(N if form != "pipe" else 0):   # (a pipe cannot be seeked: the other ranks' payload is read and dropped)
        own = sum(b for b, w in zip(payload, want) if w == k)
        if read[k] != own:
            return tag, "rank %d read %d bytes of payload, its utterances hold %d" % (k, read[k], own)
    return tag, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad = 0
    for i in range(args.cases):
        rs = np.random.RandomState(args.seed * 1000 + i)
        tmp = tempfile.mkdtemp(prefix="shard_fuzz_")
        tag, msg = one_case(rs, tmp)
        print("case %d (%s): %s" % (i, tag, msg or "ok"), flush=True)
        bad += msg is not None
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)
    print("%d of %d cases differ" % (bad, args.cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
