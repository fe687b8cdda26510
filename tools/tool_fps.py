"""Throughput of the command-line tools on a synthetic archive, host I/O and PCIe uploads included
(the `fps` the tools log themselves).  usage: python tools/tool_fps.py [swbd|rm] [frames] [tool options...]
TOOL_FPS_RAW_DIM=40: the archive holds 40-column features and the tools splice them to the network's input width with
--feature-transform (an <expand> of input/40 frames of context), with utterances of TOOL_FPS_MIN_LEN..TOOL_FPS_MAX_LEN
frames (default 300..1200) -- the shape speech recipes feed these tools."""
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
raw_dim = int(os.environ.get("TOOL_FPS_RAW_DIM", "0")) or dims[0]
lo, hi = int(os.environ.get("TOOL_FPS_MIN_LEN", "300")), int(os.environ.get("TOOL_FPS_MAX_LEN", "1200"))
assert dims[0] % raw_dim == 0 and (dims[0] // raw_dim) % 2 == 1 or raw_dim == dims[0], "input width must be an odd multiple of the raw width"
tmp = tempfile.mkdtemp(prefix="tool_fps_", dir="/tmp")
rs = np.random.RandomState(777)
i32 = lambda v: b"\x04" + np.int32(v).tobytes()
feats, ali = os.path.join(tmp, "feats.ark"), os.path.join(tmp, "ali.ark")
n = 0
with open(feats, "wb") as ff, open(ali, "wb") as fa:
    u = 0
    while n < frames:
        t = int(rs.randint(lo, hi + 1))
        key = ("utt%06d " % u).encode()
        ff.write(key + b"\0BFM " + i32(t) + i32(raw_dim) + rs.standard_normal((t, raw_dim)).astype("<f4").tobytes())
        lab = rs.randint(0, dims[-1], t).astype("<i4")
        body = np.empty(t, dtype=[("s", "u1"), ("v", "<i4")])
        body["s"], body["v"] = 4, lab
        fa.write(key + b"\0B" + i32(t) + body.tobytes())
        n += t
        u += 1
model = os.path.join(tmp, "in.nnet")
bench.write_model(model, dims, seed=1)
if raw_dim != dims[0]:
    ctx = (dims[0] // raw_dim) // 2
    transform = os.path.join(tmp, "splice.nnet")
    off = np.arange(-ctx, ctx + 1, dtype="<f8")
    open(transform, "wb").write(b"\0B<expand> " + i32(dims[0]) + i32(raw_dim) + b"DV " + i32(len(off)) + off.tobytes())
    extra = ["--feature-transform=" + transform] + extra
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
