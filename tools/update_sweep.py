#!/usr/bin/env python3
"""Sweep of tile plans for the weight-update GEMM (W -= lr * e^T * in, K = bunch 256) at the Switchboard layer
shapes: microseconds per call of hipF_affine_update_fused with weights cycling through several copies (streamed
from HBM as in a training step) or staying on one copy (cache-resident)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
h.init(-1)
N = 256
rs = np.random.RandomState(0)
for (O, I, ncopy) in ((2048, 2048, 16), (2048, 2048, 1), (9000, 2048, 4), (9000, 2048, 1)):
    W0 = (0.05 * rs.randn(O, I)).astype(f32)
    Ws = [h.DevMatrix.from_numpy(W0) for _ in range(ncopy)]
    b = h.DevMatrix.from_numpy(np.zeros(O, f32), False)
    x = h.DevMatrix.from_numpy(rs.rand(N, I).astype(f32))
    e = h.DevMatrix.from_numpy((0.01 * rs.randn(N, O)).astype(f32))

    fused = True

    def call(i):
        W = Ws[i % ncopy]
        if not fused:
            h.call("hipF_affine_update_weights", None, N, O, I, e.ptr, e.stride, x.ptr, x.stride, f32(1), f32(0), f32(1e-6), f32(0),
                   W.ptr, W.stride, W.ptr, W.stride, None, 0)
            return
        h.call("hipF_affine_update_fused", None, N, O, I, e.ptr, e.stride, x.ptr, x.stride, f32(1), f32(0), f32(1e-6), f32(0),
               W.ptr, W.stride, W.ptr, W.stride, None, 0, b.ptr, None)

    def bench():
        for i in range(300):
            call(i)
        t = h.Timer(None)
        t.start()
        for i in range(300):
            call(i)
        return t.stop() / 0.3

    for i in range(1500):   # clock ramp
        call(i)
    ideal = 2.0 * N * O * I / 157.3e6
    res = [("auto", bench())]
    for bm, bn in ((64, 64), (128, 64), (64, 128), (128, 128)):
        h.call("hipK_gemm_force_plan", bm, bn, 1)
        res.append(("%dx%d" % (bm, bn), bench()))
    h.call("hipK_gemm_force_plan", 0, 0, 0)
    fused = False
    res.append(("weights only (no bias workgroups)", bench()))
    print("O=%d I=%d, %d weight copies: " % (O, I, ncopy) + "  ".join("%s %.1f us (%.0f%%)" % (n, t, 100 * ideal / t) for n, t in res), flush=True)
    del Ws
