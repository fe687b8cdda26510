#!/usr/bin/env python3
"""Host -> device copies of feature matrices as the tools make them (hipK_memcpy_2d, pageable source, device rows at a
256-byte pitch): microseconds per call for utterance-like shapes.  A diagnostic; run on the GPU box.

    python tools/h2d_probe.py
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

h.init()
print("%8s %6s %10s %12s %12s %10s %14s" % ("rows", "cols", "bytes", "us per call", "GB/s", "us per row", "back: us/call"))
for cols in (13, 40, 120, 440, 1320):
    for rows in (30, 100, 366, 1000, 3000):
        a = np.random.standard_normal((rows, cols)).astype(np.float32)
        d = h.DevMatrix(rows, cols)
        for _ in range(3):
            d.set(a)
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            d.set(a)
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            b = d.numpy()
        db = (time.perf_counter() - t0) / n
        assert np.array_equal(a, b)
        print("%8d %6d %10d %12.1f %12.3f %10.3f %14.1f" % (rows, cols, a.nbytes, dt * 1e6, a.nbytes / dt / 1e9, dt * 1e6 / rows, db * 1e6), flush=True)
