#!/usr/bin/env python3
"""Soak run: the tools on long tables of utterances of very different lengths, with the host's peak resident memory per
run and the device memory the run held at most (the device library's own ledger, KALDI_HIP_MEM_REPORT).  Neither may
grow with the number of utterances.  (--poll samples rocm-smi as well; on boxes where it may not see the process it
reports 0.)

    python tools/soak.py [--out FILE]
"""
import argparse
import os
import re
import resource
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kaldi_amd", "bin")
f32 = np.float32


def w_i32(v):
    return b"\x04" + int(v).to_bytes(4, "little", signed=True)


def write_model(path, dims, rs):
    out = b"\0B"
    for i in range(len(dims) - 1):
        W = (rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(f32)
        out += b"<biasedlinearity> " + w_i32(dims[i + 1]) + w_i32(dims[i]) + b"FM " + w_i32(dims[i + 1]) + w_i32(dims[i]) + W.tobytes()
        out += b"FV " + w_i32(dims[i + 1]) + np.zeros(dims[i + 1], f32).tobytes()
        kind = b"<softmax> " if i + 2 == len(dims) else b"<sigmoid> "
        out += kind + w_i32(dims[i + 1]) + w_i32(dims[i + 1])
    open(path, "wb").write(out)


def write_rbm(path, vis, hid, rs):
    W = (0.05 * rs.standard_normal((hid, vis))).astype(f32)
    open(path, "wb").write(b"\0B<rbm> " + w_i32(hid) + w_i32(vis) + b"gauss bern FM " + w_i32(hid) + w_i32(vis) + W.tobytes() +
                           b"FV " + w_i32(vis) + np.zeros(vis, f32).tobytes() + b"FV " + w_i32(hid) + np.zeros(hid, f32).tobytes())


def write_tables(stem, n, dim, classes, rs):
    with open(stem + ".feats.ark", "wb") as f, open(stem + ".ali.ark", "wb") as a:
        frames = 0
        for i in range(n):
            rows = int(rs.choice([20, 50, 130, 400, 1200]) + rs.randint(0, 17))
            key = ("utt%06d " % i).encode()
            f.write(key + b"\0BFM " + w_i32(rows) + w_i32(dim) + rs.standard_normal((rows, dim)).astype(f32).tobytes())
            a.write(key + b"\0B" + w_i32(rows) + b"".join(w_i32(x) for x in rs.randint(0, classes, rows)))
            frames += rows
    return frames


def vram_used():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20).stdout.decode()
        m = re.findall(r"Used Memory \(B\):\s*(\d+)", out)
        return max(int(x) for x in m) if m else None
    except Exception:
        return None


POLL = False


def run(cmd):
    samples, stop = [], threading.Event()
    if not POLL:
        stop.set()

    def poll():
        while not stop.is_set():
            v = vram_used()
            if v is not None:
                samples.append(v)
            stop.wait(0.5)
    t = threading.Thread(target=poll)
    t.start()
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    t0 = time.time()
    r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, KALDI_HIP_MEM_REPORT="1"))
    dt = time.time() - t0
    stop.set()
    t.join()
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss  # (a running maximum over all children so far)
    return r.returncode, dt, rss, before, samples, r.stderr.decode(errors="replace")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--poll", action="store_true", help="sample rocm-smi while the tools run")
    ap.add_argument("--no-poll", action="store_true", help="(the default)")
    args = ap.parse_args()
    global POLL
    POLL = args.poll
    rs = np.random.RandomState(1)
    tmp = tempfile.mkdtemp(prefix="soak_")
    write_model(tmp + "/m.nnet", [40, 256, 256, 120], rs)
    write_rbm(tmp + "/r.rbm", 40, 256, rs)
    write_model(tmp + "/ae.nnet", [40, 256, 40], rs)  # (an auto-encoder shape for the MSE trainer: the targets are the features)
    lines = []
    base = vram_used()
    lines.append("GPU memory in use before anything runs: %s MiB" % (None if base is None else base >> 20))
    for n in (500, 2000, 6000):  # ascending: the children's peak RSS is a running maximum
        frames = write_tables("%s/t%d" % (tmp, n), n, 40, 120, rs)
        for tool, cmd in (("nnet-forward", [BIN + "/nnet-forward", tmp + "/m.nnet", "ark:%s/t%d.feats.ark" % (tmp, n), "ark:/dev/null"]),
                          ("nnet-train-xent-hardlab-frmshuff", [BIN + "/nnet-train-xent-hardlab-frmshuff", "--learn-rate=0.001",
                                                              "ark:%s/t%d.feats.ark" % (tmp, n), "ark:%s/t%d.ali.ark" % (tmp, n), tmp + "/m.nnet", tmp + "/o.nnet"]),
                          ("nnet-train-xent-hardlab-perutt", [BIN + "/nnet-train-xent-hardlab-perutt", "--learn-rate=0.0001",
                                                              "ark:%s/t%d.feats.ark" % (tmp, n), "ark:%s/t%d.ali.ark" % (tmp, n), tmp + "/m.nnet", tmp + "/o2.nnet"]),
                          ("nnet-train-mse-tgtmat-frmshuff", [BIN + "/nnet-train-mse-tgtmat-frmshuff", "--learn-rate=0.0001",
                                                              "ark:%s/t%d.feats.ark" % (tmp, n), "ark:%s/t%d.feats.ark" % (tmp, n), tmp + "/ae.nnet", tmp + "/o3.nnet"]),
                          ("rbm-train-cd1-frmshuff", [BIN + "/rbm-train-cd1-frmshuff", "--learn-rate=0.001", tmp + "/r.rbm",
                                                      "ark:%s/t%d.feats.ark" % (tmp, n), tmp + "/o.rbm"])):
            rc, dt, rss, before, samples, err = run(cmd)
            vr = [s >> 20 for s in samples]
            tail = [l for l in err.strip().splitlines() if "Done" in l or "device memory held" in l][-2:]
            tail = [re.sub(r"^LOG \([^)]*\) ", "", l) for l in tail]
            lines.append("%-34s %5d utterances %8d frames: rc %d, %.1f s, peak host RSS of all runs so far %d MiB (before this run %d), "
                         "%s| %s" %
                         (tool, n, frames, rc, dt, rss >> 10, before >> 10,
                          ("rocm-smi min/max %s/%s MiB over %d samples " % (min(vr), max(vr), len(vr))) if vr else "", " | ".join(tail)[:260]))
            print(lines[-1], flush=True)
        os.remove("%s/t%d.feats.ark" % (tmp, n))
        os.remove("%s/t%d.ali.ark" % (tmp, n))
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
