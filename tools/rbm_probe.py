"""Times the reconstruction product of an RBM CD-1 step (256 x V = 256 x H * H x V, bias in the epilogue) under every
tile / split-K plan, and the two propagation products, with HIP events.  usage: python tools/rbm_probe.py [V H]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
V, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (440, 2048)
N = 256
h.init(-1)
rs = np.random.RandomState(0)
hid = h.DevMatrix.from_numpy(rs.rand(N, H).astype(f32))
vis = h.DevMatrix.from_numpy(rs.randn(N, V).astype(f32))
W = h.DevMatrix.from_numpy((0.1 * rs.randn(H, V)).astype(f32))
vb = h.DevMatrix.from_numpy(np.zeros(V, f32), False)
hb = h.DevMatrix.from_numpy(np.zeros(H, f32), False)
out_v, out_h = h.DevMatrix(N, V), h.DevMatrix(N, H)


def bench(fn, iters=200):
    for _ in range(20):
        fn()
    t = h.Timer()
    t.start()
    for _ in range(iters):
        fn()
    return t.stop() / iters * 1e3


rec = lambda: h.call("hipF_affine_forward", None, h.ACT_NONE, h.NO_TRANS, N, V, H, hid.ptr, hid.stride, W.ptr, W.stride, vb.ptr, out_v.ptr, out_v.stride)
prop = lambda: h.call("hipF_affine_forward", None, h.ACT_SIGMOID, h.TRANS, N, H, V, vis.ptr, vis.stride, W.ptr, W.stride, hb.ptr, out_h.ptr, out_h.stride)
a = h.DevMatrix(4096, 4096)
for _ in range(300):
    h.call("hipF_scale", None, a.ptr, f32(1.0), a.dim)
print("reconstruct %dx%dx%d, planner's choice: %.2f us" % (N, V, H, bench(rec)))
print("propagate   %dx%dx%d, planner's choice: %.2f us" % (N, H, V, bench(prop)))
for name, fn in (("reconstruct", rec), ("propagate", prop)):
    for bm, bn in ((64, 64), (32, 64), (32, 32)):
        for s in (1, 2, 4, 8):
            h.call("hipK_gemm_force_plan", bm, bn, s)
            try:
                print("%s %dx%d split %d: %.2f us" % (name, bm, bn, s, bench(fn)))
            except h.HipError as e:
                print("%s %dx%d split %d: %s" % (name, bm, bn, s, e))
h.call("hipK_gemm_force_plan", 0, 0, 0)
