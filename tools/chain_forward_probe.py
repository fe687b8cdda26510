"""hipF_affine_forward_chain against one hipF_affine_forward launch per layer: same bits, and the time of both (HIP events).
usage: python tools/chain_forward_probe.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
h.init(-1)
rs = np.random.RandomState(0)
a = h.DevMatrix(4096, 4096)
for _ in range(300):
    h.call("hipF_scale", None, a.ptr, f32(1.0), a.dim)
N = 256
for width, nl in ((2048, 6), (1024, 3), (2048, 2)):
    x = h.DevMatrix.from_numpy(rs.randn(N, width).astype(f32))
    Ws = [h.DevMatrix.from_numpy((0.05 * rs.randn(width, width)).astype(f32)) for _ in range(nl)]
    bs = [h.DevMatrix.from_numpy((0.1 * rs.randn(width)).astype(f32), False) for _ in range(nl)]
    act_a = [x] + [h.DevMatrix(N, width) for _ in range(nl)]
    act_b = [x] + [h.DevMatrix(N, width) for _ in range(nl)]
    items = (h.AffineForward * nl)()
    for i in range(nl):
        items[i] = h.AffineForward(width, width, act_b[i]._p.value, act_b[i].stride, Ws[i]._p.value, Ws[i].stride, bs[i]._p.value,
                                   act_b[i + 1]._p.value, act_b[i + 1].stride)

    def separate():
        for i in range(nl):
            h.call("hipF_affine_forward", None, h.ACT_SIGMOID, h.TRANS, N, width, width, act_a[i].ptr, act_a[i].stride, Ws[i].ptr,
                   Ws[i].stride, bs[i].ptr, act_a[i + 1].ptr, act_a[i + 1].stride)

    def chain():
        h.call("hipF_affine_forward_chain", None, h.ACT_SIGMOID, N, nl, ctypes.cast(items, ctypes.c_void_p))

    for rep in range(3):
        x.set(rs.randn(N, width).astype(f32))
        separate()
        chain()
        for i in range(1, nl + 1):
            assert np.array_equal(act_a[i].numpy(), act_b[i].numpy()), (width, rep, i)

    def bench(fn, iters=300):
        for _ in range(30):
            fn()
        t = h.Timer()
        t.start()
        for _ in range(iters):
            fn()
        return t.stop() / iters * 1e3

    for rep in range(2):
        print("%d layers %d x %d, bunch %d: separate %.2f us, chain %.2f us" % (nl, width, width, N, bench(separate), bench(chain)))
print("chain probe ok")
