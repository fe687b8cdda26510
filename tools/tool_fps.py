"""Throughput of the command-line tools on a synthetic archive, host I/O and PCIe uploads included
(the `fps` the tools log themselves).  usage: python tools/tool_fps.py [swbd|rm] [frames]"""
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (model writer and the layer sizes)

which = sys.argv[1] if len(sys.argv) > 1 else "swbd"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
extra = sys.argv[3:]  # further options for both tools, e.g. --profile=true
dims = bench.WORKLOADS[which]["dims"]
tmp = tempfile.mkdtemp(prefix="tool_fps_", dir="/tmp")
rs = np.random.RandomState(777)
i32 = lambda v: b"\x04" + np.int32(v).tobytes()
feats, ali = os.path.join(tmp, "feats.ark"), os.path.join(tmp, "ali.ark")
n = 0
with open(feats, "wb") as ff, open(ali, "wb") as fa:
    u = 0
    while n < frames:
        t = int(rs.randint(300, 1201))
        key = ("utt%06d " % u).encode()
        ff.write(key + b"\0BFM " + i32(t) + i32(dims[0]) + rs.standard_normal((t, dims[0])).astype("<f4").tobytes())
        lab = rs.randint(0, dims[-1], t).astype("<i4")
        body = np.empty(t, dtype=[("s", "u1"), ("v", "<i4")])
        body["s"], body["v"] = 4, lab
        fa.write(key + b"\0B" + i32(t) + body.tobytes())
        n += t
        u += 1
model = os.path.join(tmp, "in.nnet")
bench.write_model(model, dims, seed=1)
print("%s: %d utterances, %d frames, archive %.0f MB" % (which, u, n, os.path.getsize(feats) / 1e6), flush=True)


def run(cmd):
    t0 = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.time() - t0
    err = r.stderr.decode()
    if r.returncode != 0:
        print(err[-2000:])
        sys.exit(1)
    for line in err.splitlines():
        if extra or "fps" in line or "FRAME_ACCURACY" in line or "time" in line.lower():
            print("   ", line.strip())
    print("    wall %.2f s -> %.0f frames/s including start-up" % (wall, n / wall), flush=True)


bindir = os.path.join(ROOT, "kaldi_amd", "bin")
if os.environ.get("TOOL_FPS_SKIP_TRAIN") != "1":
    print("nnet-train-xent-hardlab-frmshuff:")
    run([os.path.join(bindir, "nnet-train-xent-hardlab-frmshuff"), "--learn-rate=0.008", "--binary=true"] + extra + ["ark:" + feats,
         "ark:" + ali, model, os.path.join(tmp, "out.nnet")])
print("nnet-forward -> ark:/dev/null:")
run([os.path.join(bindir, "nnet-forward")] + extra + [model, "ark:" + feats, "ark:/dev/null"])
print("nnet-forward -> ark file in %s:" % tmp)
run([os.path.join(bindir, "nnet-forward")] + extra + [model, "ark:" + feats, "ark:" + os.path.join(tmp, "post.ark")])
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
