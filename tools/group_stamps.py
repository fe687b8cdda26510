"""Diagnostic: in-kernel stamps of one grouped weight-update launch (hipF_affine_update_group, one workgroup per
tile) taken after a long run of back-to-back launches (clock settled).  Needs the stamps build:
usage: KALDI_AMD_HIP_LIB=tools/dbg/libkaldi_hip.so python tools/group_stamps.py"""
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
layers = []
for li in range(len(dims) - 1):
    I, O = dims[li], dims[li + 1]
    layers.append(dict(O=O, I=I, W=h.DevMatrix.from_numpy((0.05 * rs.randn(O, I)).astype(f32)),
                       b=h.DevMatrix.from_numpy(np.zeros(O, f32), False), x=h.DevMatrix.from_numpy(rs.rand(N, I).astype(f32)),
                       e=h.DevMatrix.from_numpy((0.01 * rs.randn(N, O)).astype(f32))))
order = list(reversed(range(len(layers))))
arr = (h.AffineUpdate * len(order))()
nblk = 0
for k, li in enumerate(order):
    L = layers[li]
    arr[k] = h.AffineUpdate(N, L["O"], L["I"], L["e"]._p.value, L["e"].stride, L["x"]._p.value, L["x"].stride, 1.0, 0.0, 1e-6, 0.0,
                            L["W"]._p.value, L["W"].stride, L["W"]._p.value, L["W"].stride, None, 0, L["b"]._p.value, None)
    nblk += ((L["O"] + 63) // 64 * ((L["I"] + 63) // 64) + 7) & ~7  # (upper bound for either tile size)
for _ in range(300):
    h.call("hipF_affine_update_group", None, len(order), ctypes.cast(arr, ctypes.c_void_p))
buf = np.zeros(nblk * 16, np.uint64)
fn = h.lib.hipK_gemm_stamps_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.size) == 0
full = buf.reshape(nblk, 16).astype(np.int64)
full = full[full[:, 9] > 0]   # (padding blocks leave no stamps)
rt0, rt1 = full[:, 8], full[:, 9]
cyc = full[:, 7] - full[:, 0]
life = (rt1 - rt0) / 100.0
print("%d tiles; launch %.1f us (first entry -> last exit)" % (full.shape[0], (rt1.max() - rt0.min()) / 100.0))
print("workgroup lifetime us: median %.2f, p10 %.2f, p90 %.2f; cycles median %d; clock (cycles / real time) median %.3f GHz" %
      (np.median(life), np.percentile(life, 10), np.percentile(life, 90), np.median(cyc), np.median(cyc / (life * 1e3))))
names = ["entry", "loads issued", "tile0 in LDS", "first barrier", "loop end", "drain end", "stores issued", "stores done"]
d = np.diff(full[:, :8], axis=1)
print("per-workgroup phase lengths, cycles (p10 / median / p90):")
for i in range(7):
    print("  %-14s -> %-14s %8d %8d %8d" % (names[i], names[i + 1], np.percentile(d[:, i], 10), np.median(d[:, i]), np.percentile(d[:, i], 90)))
ok = (full[:, 14] > 0) & (full[:, 15] > 0)
if ok.any():
    a, b2, c = full[ok, 14] - full[ok, 5], full[ok, 15] - full[ok, 14], full[ok, 6] - full[ok, 15]
    print("epilogue of wave 0, cycles (p10 / median / p90): barrier after the K loop %d %d %d; accumulators into LDS %d %d %d; "
          "LDS reads, arithmetic, store issue %d %d %d" % (np.percentile(a, 10), np.median(a), np.percentile(a, 90), np.percentile(b2, 10),
                                                          np.median(b2), np.percentile(b2, 90), np.percentile(c, 10), np.median(c), np.percentile(c, 90)))
# how many workgroups are resident over time, and how many of them are inside their K loop
t0 = rt0.min()
edges = np.arange(0, (rt1.max() - t0) / 100.0 + 2.0, 2.0)
starts = np.histogram((rt0 - t0) / 100.0, edges)[0]
print("workgroups entering per 2-us bin: " + " ".join("%d" % v for v in starts))
