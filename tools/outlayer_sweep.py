#!/usr/bin/env python3
"""Sweep of tile / split-K plans for the output layer of a training step (256 x 9000 x 2048 forward GEMM + the
softmax / cross-entropy row kernel that sums its slabs): microseconds per call of hipF_affine_forward_xent, weights
cycling through four copies (295 MB: streamed from HBM as in a training step)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
h.init(-1)
N, O, I = 256, 9000, 2048
if len(sys.argv) > 3:
    N, O, I = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rs = np.random.RandomState(0)
W0 = (0.05 * rs.randn(O, I)).astype(f32)
Ws = [h.DevMatrix.from_numpy(W0) for _ in range(4)]
b = h.DevMatrix.from_numpy(np.zeros(O, f32), False)
x = h.DevMatrix.from_numpy(rs.rand(N, I).astype(f32))
post, diff = h.DevMatrix(N, O), h.DevMatrix(N, O)
tgt = h.DevMatrix.from_numpy(rs.randint(0, O, N).astype(np.int32), pitched=False)
cnt = h.DevMatrix(1, 2, np.int64, pitched=False)
loss = h.DevMatrix(1, 1, np.float64, pitched=False)


def call(i):
    W = Ws[i % 4]
    h.call("hipF_affine_forward_xent", None, h.TRANS, N, O, I, x.ptr, x.stride, W.ptr, W.stride, b.ptr, post.ptr, post.stride, 0,
           tgt.ptr, diff.ptr, diff.stride, None, None, cnt.ptr, loss.ptr)


def bench():
    for i in range(400):
        call(i)
    t = h.Timer(None)
    t.start()
    for i in range(300):
        call(i)
    return t.stop() / 0.3


for i in range(600):   # clock ramp
    call(i)
print("auto: %.1f us" % bench())
res = []
for bm, bn in ((32, 32), (32, 64), (64, 64), (128, 64), (64, 128), (128, 128)):
    for s in (1, 2, 4, 8):
        h.call("hipK_gemm_force_plan", bm, bn, s)
        try:
            res.append((bench(), bm, bn, s))
        except h.HipError as e:
            print("skip", bm, bn, s, e)
h.call("hipK_gemm_force_plan", 0, 0, 0)
for t, bm, bn, s in sorted(res):
    print("%3dx%-3d S%d: %.1f us" % (bm, bn, s, t))
print("auto again: %.1f us   (ideal at 157.3 TF/s: %.1f us)" % (bench(), 2.0 * N * O * I / 157.3e6))
