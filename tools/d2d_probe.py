#!/usr/bin/env python3
"""Device -> device copies of pitched matrices as CuMatrix::CopyFromMat / CopyRowsFromMat / Cache::AddData make them
(hipK_memcpy_2d, kind 2): microseconds of GPU time per copy and the bandwidth reached, for utterance- and cache-like
shapes.  A diagnostic; run on the GPU box.    python tools/d2d_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

h.init()
print("%8s %6s %12s %12s %12s" % ("rows", "cols", "bytes", "us per copy", "GB/s (r+w)"))
for rows, cols in ((366, 40), (366, 440), (1200, 440), (256, 2048), (4096, 440), (32768, 440), (32768, 1320), (256, 9000)):
    a = h.DevMatrix.from_numpy(np.random.standard_normal((rows, cols)).astype(np.float32))
    b = h.DevMatrix(rows, cols)
    es = 4
    def copy():
        h.call("hipK_memcpy_2d", None, b.ptr, b.stride * es, a.ptr, a.stride * es, cols * es, rows, 2)
    for _ in range(5):
        copy()
    h.synchronize()
    n = 100
    t = h.Timer()
    t.start()
    for _ in range(n):
        copy()
    us = t.stop() * 1e3 / n
    assert np.array_equal(a.numpy(), b.numpy())
    print("%8d %6d %12d %12.2f %12.1f" % (rows, cols, rows * cols * es, us, 2 * rows * cols * es / us / 1e3), flush=True)
