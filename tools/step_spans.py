#!/usr/bin/env python3
"""Timeline of the GEMM launches of one steady-state training step, from HIP events on their own streams
(no tracer: rocprofv3 lengthens the gaps between dependent kernels).  usage: python tools/step_spans.py [warm_steps]"""
import ctypes
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kaldi_amd import hip as h  # noqa: E402
from kaldi_amd import nnet as nn  # noqa: E402

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 60
h.init(0)
dims = bench.WORKLOADS["swbd"]["dims"]
tmp = tempfile.mkdtemp(prefix="step_spans_")
model = os.path.join(tmp, "m.nnet")
bench.write_model(model, dims, seed=1)
tr = nn.Trainer(model)
tr.set_train_opts(0.008)
tr.keep_posteriors(False)
rs = np.random.RandomState(100)
B = bench.BUNCH
cache = h.DevMatrix.from_numpy(rs.standard_normal((16 * B, dims[0])).astype(np.float32))
labels = h.DevMatrix.from_numpy(rs.randint(0, dims[-1], 16 * B).astype(np.int32), pitched=False)


def step(i):
    b = i % 16
    tr.train_bunch(ctypes.c_void_p(cache.ptr.value + b * B * cache.stride * 4), cache.stride, B, dims[0],
                   ctypes.c_void_p(labels.ptr.value + b * B * 4))


for i in range(warm):
    step(i)
nn.synchronize()
h.call("hipK_gemm_timing", 1)
for i in range(4):
    step(i)
cap = 256
st, du, gf = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
n = ctypes.c_int()
h.call("hipK_gemm_timing_spans", st, du, gf, cap, ctypes.byref(n))
h.call("hipK_gemm_timing", 0)
per = n.value // 4
print("%d launches, %d per step; step 3 (us; start, end, duration, GFLOP, TFLOP/s):" % (n.value, per))
t0 = st[2 * per] * 1e3
for k in range(2 * per, 3 * per + 2):
    if k >= n.value:
        break
    s_us, d_us = st[k] * 1e3 - t0, du[k] * 1e3
    print("%8.1f %8.1f %7.1f  %6.2f GF %6.1f TF/s" % (s_us, s_us + d_us, d_us, gf[k], gf[k] / d_us * 1e-3 if d_us > 0 else 0))
tr.close()
