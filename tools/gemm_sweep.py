"""Sweeps tile / split-K plans for the hot-path GEMM shapes (timed with HIP events).
usage: python tools/gemm_sweep.py [swbd|rm]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
PEAK = 157.3e12


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    t = h.Timer()
    t.start()
    for _ in range(iters):
        fn()
    return t.stop() / iters * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "swbd"
    h.init(-1)
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    shapes = [(2048, 2048), (2048, 1320), (9000, 2048)] if which == "swbd" else [(1024, 1024), (1024, 440), (1483, 1024)]
    rs = np.random.RandomState(0)
    for (O_, I) in shapes:
        x = h.DevMatrix.from_numpy(rs.randn(N, I).astype(f32))
        W = h.DevMatrix.from_numpy((0.1 * rs.randn(O_, I)).astype(f32))
        b = h.DevMatrix.from_numpy(np.zeros(O_, f32), False)
        y = h.DevMatrix(N, O_)
        e = h.DevMatrix.from_numpy((0.01 * rs.randn(N, O_)).astype(f32))
        eo = h.DevMatrix(N, I)
        fl = 2.0 * N * O_ * I
        ops = {
            "fwd(NT)+sigm": lambda: h.call("hipF_affine_forward", None, h.ACT_SIGMOID, h.TRANS, N, O_, I, x.ptr, x.stride,
                                           W.ptr, W.stride, b.ptr, y.ptr, y.stride),
            "bwd(NN)+dsig": lambda: h.call("hipF_affine_backward", None, N, O_, I, e.ptr, e.stride, W.ptr, W.stride, x.ptr,
                                           x.stride, eo.ptr, eo.stride),
            "upd(TN)+sgd ": lambda: h.call("hipF_affine_update", None, N, O_, I, e.ptr, e.stride, x.ptr, x.stride, f32(1),
                                           f32(0), f32(1e-9), f32(0), W.ptr, W.stride, None, 0, b.ptr, None),
        }
        for name, fn in ops.items():
            res = []
            for (bm, bn) in [(64, 64), (128, 64), (64, 128), (128, 128), (32, 64), (32, 32)]:
                if name.startswith("upd") and bm == 32:
                    continue  # the update epilogue has no form with K split over waves
                for s in ([1] if name.startswith("upd") else [1, 2, 4, 8]):
                    h.call("hipK_gemm_force_plan", bm, bn, s)
                    try:
                        t = bench(fn)
                    except h.HipError:
                        continue
                    res.append((t, bm, bn, s))
            h.call("hipK_gemm_force_plan", 0, 0, 0)
            auto = bench(fn)
            res.sort()
            print("O=%5d I=%5d %s auto %6.1f us (%4.1f%%) | " % (O_, I, name, auto, fl / auto / PEAK * 1e8) +
                  "  ".join("%dx%d/S%d %.1f" % (bm, bn, s, t) for t, bm, bn, s in res[:8]), flush=True)


if __name__ == "__main__":
    main()
