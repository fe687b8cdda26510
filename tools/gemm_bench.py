"""Times the hot-path GEMM shapes through the C ABI (HIP events, default stream).
usage: python tools/gemm_bench.py [rm|swbd] [iters]   (KALDI_HIP_GEMM_FORCE=BM,BN,S to pin a plan)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
PEAK = 157.3e12


def bench(fn, iters):
    for _ in range(3):
        fn()
    t = h.Timer()
    t.start()
    for _ in range(iters):
        fn()
    return t.stop() / iters * 1e3  # us


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "swbd"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    h.init(-1)
    N = 256
    dims = [1320] + [2048] * 7 + [9000] if which == "swbd" else [440, 1024, 1024, 1024, 1024, 1483]
    rs = np.random.RandomState(0)
    shapes = sorted(set((dims[i + 1], dims[i]) for i in range(len(dims) - 1)))
    tot_us, tot_fl = 0.0, 0.0
    for (O_, I) in shapes:
        cnt = sum(1 for i in range(len(dims) - 1) if (dims[i + 1], dims[i]) == (O_, I))
        first = dims[0] == I and dims[1] == O_
        x = h.DevMatrix.from_numpy(rs.randn(N, I).astype(f32))
        W = h.DevMatrix.from_numpy((0.1 * rs.randn(O_, I)).astype(f32))
        b = h.DevMatrix.from_numpy(np.zeros(O_, f32), False)
        y = h.DevMatrix(N, O_)
        e = h.DevMatrix.from_numpy((0.01 * rs.randn(N, O_)).astype(f32))
        eo = h.DevMatrix(N, I)
        fl = 2.0 * N * O_ * I
        act = h.ACT_SOFTMAX if O_ == dims[-1] else h.ACT_SIGMOID
        t_f = bench(lambda: h.call("hipF_affine_forward", None, act, h.TRANS, N, O_, I, x.ptr, x.stride, W.ptr,
                                   W.stride, b.ptr, y.ptr, y.stride), iters)
        t_b = bench(lambda: h.call("hipF_affine_backward", None, N, O_, I, e.ptr, e.stride, W.ptr, W.stride, x.ptr,
                                   x.stride, eo.ptr, eo.stride), iters)
        t_u = bench(lambda: h.call("hipF_affine_update", None, N, O_, I, e.ptr, e.stride, x.ptr, x.stride, f32(1),
                                   f32(0), f32(1e-9), f32(0), W.ptr, W.stride, None, 0, b.ptr, None), iters)
        t_g = bench(lambda: h.call("hipF_gemm", None, h.NO_TRANS, h.TRANS, N, O_, I, f32(1), x.ptr, x.stride, W.ptr,
                                   W.stride, f32(0), y.ptr, y.stride), iters)
        print("O=%5d I=%5d x%d  fwd %7.1f us (%5.1f%%)  bwd %7.1f us (%5.1f%%)  upd %7.1f us (%5.1f%%)  gemmNT %7.1f us (%5.1f%%)"
              % (O_, I, cnt, t_f, fl / t_f / PEAK * 1e8, t_b, fl / t_b / PEAK * 1e8, t_u, fl / t_u / PEAK * 1e8, t_g,
                 fl / t_g / PEAK * 1e8), flush=True)
        tot_us += cnt * (t_f + t_u) + (0 if first else cnt * t_b)
        tot_fl += cnt * 2 * fl + (0 if first else cnt * fl)
    print("sum of GEMM-side kernels per minibatch: %.1f us -> %.0f frames/s, %.1f%% of fp32 MFMA peak"
          % (tot_us, N / tot_us * 1e6, tot_fl / tot_us / PEAK * 1e8))


if __name__ == "__main__":
    main()
