"""Runs one hot-path GEMM shape a few times (for rocprofv3 --pmc passes).
usage: python tools/gemm_one.py fwd|bwd|upd O I [bm bn s] [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
op, O_, I = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
plan = [int(a) for a in sys.argv[4:7]] if len(sys.argv) >= 7 else [0, 0, 0]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
h.init(-1)
N = 256
rs = np.random.RandomState(0)
x = h.DevMatrix.from_numpy(rs.randn(N, I).astype(f32))
W = h.DevMatrix.from_numpy((0.1 * rs.randn(O_, I)).astype(f32))
b = h.DevMatrix.from_numpy(np.zeros(O_, f32), False)
y = h.DevMatrix(N, O_)
e = h.DevMatrix.from_numpy((0.01 * rs.randn(N, O_)).astype(f32))
eo = h.DevMatrix(N, I)
h.call("hipK_gemm_force_plan", *plan)
for _ in range(iters):
    if op == "fwd":
        h.call("hipF_affine_forward", None, h.ACT_SIGMOID, h.TRANS, N, O_, I, x.ptr, x.stride, W.ptr, W.stride, b.ptr,
               y.ptr, y.stride)
    elif op == "bwd":
        h.call("hipF_affine_backward", None, N, O_, I, e.ptr, e.stride, W.ptr, W.stride, x.ptr, x.stride, eo.ptr, eo.stride)
    else:
        h.call("hipF_affine_update", None, N, O_, I, e.ptr, e.stride, x.ptr, x.stride, f32(1), f32(0), f32(1e-9), f32(0),
               W.ptr, W.stride, None, 0, b.ptr, None)
h.synchronize()
