"""Diagnostic: a chain of 6 dependent 2048x2048 forward layers on a 256-row bunch, run (a) as one chain,
(b) as two independent half-bunch chains on two streams, (c) four quarter-bunch chains."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h
f32 = np.float32
h.init(-1)
N, D, L = 256, 2048, 6
rs = np.random.RandomState(0)
W = [h.DevMatrix.from_numpy((0.05 * rs.randn(D, D)).astype(f32)) for _ in range(L)]
b = [h.DevMatrix.from_numpy(np.zeros(D, f32), False) for _ in range(L)]
x = [h.DevMatrix.from_numpy(rs.rand(N, D).astype(f32)) for _ in range(L + 1)]
streams = [None]
for _ in range(3):
    s = ctypes.c_void_p()
    h.call("hipK_stream_create", ctypes.byref(s))
    streams.append(s)

def layer(i, r0, n, s, bwd=False):
    off = r0 * x[0].stride * 4
    a = ctypes.c_void_p(x[i].ptr.value + off)
    o = ctypes.c_void_p(x[i + 1].ptr.value + off)
    if bwd:
        h.call("hipF_affine_backward", s, n, D, D, o, x[0].stride, W[i].ptr, W[i].stride, a, x[0].stride, a, x[0].stride)
    else:
        h.call("hipF_affine_forward", s, h.ACT_SIGMOID, h.TRANS, n, D, D, a, x[0].stride, W[i].ptr, W[i].stride, b[i].ptr, o, x[0].stride)

def run(parts, iters=30, bwd=False):
    n = N // parts
    def once():
        order = range(L - 1, -1, -1) if bwd else range(L)
        for i in order:
            for pp in range(parts):
                layer(i, pp * n, n, streams[pp], bwd)
    def sync():
        h.synchronize()
        for s in streams[1:]:
            h.call("hipK_stream_synchronize", s)
    for _ in range(3): once()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters): once()
    sync()
    return (time.perf_counter() - t0) / iters * 1e6

for bwd in (False, True):
    for parts in (1, 2, 4):
        print("%s chain of %d layers, %d part(s): %.1f us" % ("backward" if bwd else "forward", L, parts, run(parts, bwd=bwd)))
