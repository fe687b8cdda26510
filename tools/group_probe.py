#!/usr/bin/env python3
"""The eight weight updates of a Switchboard-sized step (bunch 256) launched one by one
(hipF_affine_update_fused) and as one grouped launch (hipF_affine_update_group): microseconds per
set, and a bit-for-bit comparison of the results.
Weights cycle through two copies of the net (370 MB: not cache-resident, as in a training step)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
h.init(-1)
N = 256
dims = [1320] + [2048] * 7 + [9000]
rs = np.random.RandomState(0)
ncopy = 2
layers = []
for li in range(len(dims) - 1):
    I, O = dims[li], dims[li + 1]
    W0 = (0.05 * rs.randn(O, I)).astype(f32)
    layers.append(dict(
        O=O, I=I, W0=W0,
        Ws=[h.DevMatrix.from_numpy(W0) for _ in range(ncopy)],
        bs=[h.DevMatrix.from_numpy(np.zeros(O, f32), False) for _ in range(ncopy)],
        x=h.DevMatrix.from_numpy(rs.rand(N, I).astype(f32)),
        e=h.DevMatrix.from_numpy((0.01 * rs.randn(N, O)).astype(f32))))
order = list(reversed(range(len(layers))))  # the order of a backward pass


def one_by_one(c):
    for li in order:
        L = layers[li]
        W, b = L["Ws"][c], L["bs"][c]
        h.call("hipF_affine_update_fused", None, N, L["O"], L["I"], L["e"].ptr, L["e"].stride, L["x"].ptr, L["x"].stride,
               f32(1), f32(0), f32(1e-3), f32(0), W.ptr, W.stride, W.ptr, W.stride, None, 0, b.ptr, None)


items = []
for c in range(ncopy):
    arr = (h.AffineUpdate * len(order))()
    for k, li in enumerate(order):
        L = layers[li]
        W, b = L["Ws"][c], L["bs"][c]
        arr[k] = h.AffineUpdate(N, L["O"], L["I"], L["e"]._p.value, L["e"].stride, L["x"]._p.value, L["x"].stride, 1.0, 0.0, 1e-3, 0.0,
                                W._p.value, W.stride, W._p.value, W.stride, None, 0, b._p.value, None)
    items.append(arr)


def grouped(c):
    h.call("hipF_affine_update_group", None, len(order), ctypes.cast(items[c], ctypes.c_void_p))


def reset():
    for L in layers:
        for c in range(ncopy):
            L["Ws"][c].set(L["W0"])
            L["bs"][c].set(np.zeros(L["O"], f32))


# results: copy 0 one by one, copy 1 grouped, three steps each
reset()
for _ in range(3):
    one_by_one(0)
    grouped(1)
for li, L in enumerate(layers):
    a, b = L["Ws"][0].numpy(), L["Ws"][1].numpy()
    ba, bb = L["bs"][0].numpy(), L["bs"][1].numpy()
    assert np.array_equal(a, b) and np.array_equal(ba, bb), "layer %d differs" % li
    assert not np.array_equal(a, L["W0"])
print("grouped == one by one, bit for bit, all 8 layers", flush=True)


def bench(fn, reps=100):
    for i in range(20):
        fn(i % ncopy)
    t = h.Timer(None)
    t.start()
    for i in range(reps):
        fn(i % ncopy)
    return t.stop() / reps * 1e3


for i in range(300):  # clock ramp
    one_by_one(i % ncopy)
flops = sum(2.0 * N * L["O"] * L["I"] for L in layers)
ideal = flops / 157.3e6
print("MFMA time of the set at peak: %.1f us" % ideal)
for rnd in range(3):
    t = bench(one_by_one)
    print("one by one: %.1f us (%.0f%%)" % (t, 100 * ideal / t), flush=True)
    t = bench(grouped)
    print("grouped:    %.1f us (%.0f%%)" % (t, 100 * ideal / t), flush=True)
