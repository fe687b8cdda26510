"""Diagnostic: per-workgroup cycle stamps of one GEMM launch (needs `make -C kaldi_amd/csrc stamps`).
usage: KALDI_AMD_HIP_LIB=tools/dbg/libkaldi_hip.so python tools/gemm_stamps.py fwd|bwd|upd O I bm bn s"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
op, O_, I = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bm, bn, sk = [int(a) for a in sys.argv[4:7]]
h.init(-1)
N = 256
rs = np.random.RandomState(0)
x = h.DevMatrix.from_numpy(rs.randn(N, I).astype(f32))
W = h.DevMatrix.from_numpy((0.1 * rs.randn(O_, I)).astype(f32))
b = h.DevMatrix.from_numpy(np.zeros(O_, f32), False)
y = h.DevMatrix(N, O_)
e = h.DevMatrix.from_numpy((0.01 * rs.randn(N, O_)).astype(f32))
eo = h.DevMatrix(N, I)
h.lib.hipK_gemm_force_plan(bm, bn, sk)


def run():
    if op == "fwd":
        h.call("hipF_affine_forward", None, h.ACT_SIGMOID, h.TRANS, N, O_, I, x.ptr, x.stride, W.ptr, W.stride, b.ptr,
               y.ptr, y.stride)
        return (N + bm - 1) // bm * ((O_ + bn - 1) // bn) * sk
    if op == "bwd":
        h.call("hipF_affine_backward", None, N, O_, I, e.ptr, e.stride, W.ptr, W.stride, x.ptr, x.stride, eo.ptr, eo.stride)
        return (N + bm - 1) // bm * ((I + bn - 1) // bn) * sk
    h.call("hipF_affine_update", None, N, O_, I, e.ptr, e.stride, x.ptr, x.stride, f32(1), f32(0), f32(1e-9), f32(0),
           W.ptr, W.stride, None, 0, b.ptr, None)
    return (O_ + bm - 1) // bm * ((I + bn - 1) // bn) * sk


for _ in range(20):
    nblk = run()
buf = np.zeros(nblk * 16, np.uint64)
fn = h.lib.hipK_gemm_stamps_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = fn(buf.ctypes.data, buf.size)
assert rc == 0, rc
full = buf.reshape(nblk, 16).astype(np.int64)
car = full[full[:, 10] > 0]
if car.shape[0]:
    t_first = full[:, 8].min()
    c = (car[:, 10:14] - t_first) / 100.0
    print("%d carriers of the bias sums (us after the first tile entry, median): start %.2f, sums done %.2f, exchanged %.2f, "
          "bias written %.2f" % (car.shape[0], np.median(c[:, 0]), np.median(c[:, 1]), np.median(c[:, 2]), np.median(c[:, 3])))
st = full[:, :8]
nblk = st.shape[0]
rt0, rt1 = full[:, 8], full[:, 9]
print("real time (100 MHz counter, common to all workgroups): first entry -> last exit %.2f us; entries spread over %.2f us, "
      "exits over %.2f us; median workgroup lifetime %.2f us" % ((rt1.max() - rt0.min()) / 100.0, (rt0.max() - rt0.min()) / 100.0,
                                                                 (rt1.max() - rt1.min()) / 100.0, np.median(rt1 - rt0) / 100.0))
late = rt0 - rt0.min()
ids = np.arange(full.shape[0])
print("entry delay percentiles (us) 50/90/99/100: %s; workgroups entering > 1 us late: %d" %
      ("/".join("%.2f" % (np.percentile(late, q) / 100.0) for q in (50, 90, 99, 100)), int((late > 100).sum())))
if (late > 100).any():
    w = np.nonzero(late > 100)[0]
    print("  late workgroups (grid index: delay us, lifetime us): " +
          ", ".join("%d: %.1f, %.1f" % (ids[i], late[i] / 100.0, (rt1[i] - rt0[i]) / 100.0) for i in w[:24]))
tm = h.Timer(None)
tm.start()
for _ in range(50):
    run()
print("back-to-back launches of this shape: %.2f us each" % (tm.stop() / 50 * 1e3))
t0 = st[:, 0].min()
names = ["entry", "loads issued", "tile0 in LDS", "first barrier", "loop end", "drain end", "stores issued", "stores done"]
print("%s O=%d I=%d plan %dx%d/S%d: %d workgroups; span (first entry -> last done) %d cycles" %
      (op, O_, I, bm, bn, sk, nblk, st[:, 7].max() - t0))
print("%-14s %10s %10s %10s   (cycles since the first workgroup's entry)" % ("stamp", "min", "median", "max"))
for i, n in enumerate(names):
    c = st[:, i] - t0
    print("%-14s %10d %10d %10d" % (n, c.min(), np.median(c), c.max()))
d = np.diff(st, axis=1)
print("per-workgroup phase lengths (median / max):")
for i in range(7):
    print("  %-14s -> %-14s %8d %8d" % (names[i], names[i + 1], np.median(d[:, i]), d[:, i].max()))
