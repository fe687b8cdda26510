"""Diagnostic: host time to queue training steps vs GPU time to run them (is the host ahead of the GPU?)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kaldi_amd import hip as h, nnet as nn
import tempfile
h.init(0)
dims = bench.WORKLOADS["swbd"]["dims"]
tmp = tempfile.mkdtemp()
mp = os.path.join(tmp, "m.nnet")
bench.write_model(mp, dims, 1)
t = nn.Trainer(mp)
t.set_train_opts(0.008)
B = 256
rs = np.random.RandomState(0)
cache = h.DevMatrix.from_numpy(rs.standard_normal((4 * B, dims[0])).astype(np.float32))
lab = h.DevMatrix.from_numpy(rs.randint(0, dims[-1], 4 * B).astype(np.int32), pitched=False)
def step(i):
    b = i % 4
    t.train_bunch(ctypes.c_void_p(cache.ptr.value + b * B * cache.stride * 4), cache.stride, B, dims[0], ctypes.c_void_p(lab.ptr.value + b * B * 4))
for i in range(20): step(i)
nn.synchronize()
for n in (1, 5, 50):
    t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter()
    nn.synchronize()
    t2 = time.perf_counter()
    print("steps %3d: host queueing %.1f us/step, until GPU done %.1f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
