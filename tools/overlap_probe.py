"""Diagnostic: does the weight-update GEMM of one layer overlap with the backward GEMM of the layer
below when they are issued on two streams?  Times 6 (bwd, upd) pairs sequentially and concurrently."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
h.init(-1)
N, O_, I = 256, 2048, 2048
rs = np.random.RandomState(0)
L = 6
x = [h.DevMatrix.from_numpy(rs.rand(N, I).astype(f32)) for _ in range(L)]
W = [h.DevMatrix.from_numpy((0.1 * rs.randn(O_, I)).astype(f32)) for _ in range(L)]
b = [h.DevMatrix.from_numpy(np.zeros(O_, f32), False) for _ in range(L)]
e = [h.DevMatrix.from_numpy((0.01 * rs.randn(N, O_)).astype(f32)) for _ in range(L + 1)]
s1 = ctypes.c_void_p()
h.call("hipK_stream_create", ctypes.byref(s1))


def bwd(i, s):
    h.call("hipF_affine_backward", s, N, O_, I, e[i + 1].ptr, e[i + 1].stride, W[i].ptr, W[i].stride, x[i].ptr, x[i].stride,
           e[i].ptr, e[i].stride)


def upd(i, s):
    h.call("hipF_affine_update", s, N, O_, I, e[i + 1].ptr, e[i + 1].stride, x[i].ptr, x[i].stride, f32(1), f32(0),
           f32(1e-9), f32(0), W[i].ptr, W[i].stride, None, 0, b[i].ptr, None)


def run(concurrent, iters=20):
    h.synchronize()
    h.call("hipK_stream_synchronize", s1)
    t0 = time.perf_counter()
    for _ in range(iters):
        for i in reversed(range(L)):
            bwd(i, None)
            upd(i, s1 if concurrent else None)
    h.synchronize()
    h.call("hipK_stream_synchronize", s1)
    return (time.perf_counter() - t0) / iters * 1e6


for plan in [(0, 0, 0), (64, 64, 1), (64, 64, 2), (64, 64, 4), (128, 64, 1), (128, 128, 1), (128, 128, 2)]:
    h.call("hipK_gemm_force_plan", *plan)
    run(False, 3)
    a = run(False)
    run(True, 3)
    c = run(True)
    print("plan %s: 6x(bwd+upd) sequential %.1f us, two streams %.1f us  (ideal MFMA time %.1f us)" %
          (plan, a, c, 12 * 2.0 * N * O_ * I / 133e12 * 1e6))
