#!/usr/bin/env python3
"""Random check of the data-parallel training paths on one GPU (loopback communicator: R stand-in ranks that all hold
the bunch): for random networks (odd widths, 1 to 5 layers), bunch sizes, rank counts, hyper-parameters, exchanges
and chunk counts, R ranks on a bunch must train like ONE process on R stacked copies of it (to rounding), and the
sharded exchange must give the bits of the summed one.     python tests/fuzz/dp_fuzz.py [--cases 60] [--seed 1]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaldi_amd import hip as h  # noqa: E402
import kaldi_amd.nnet as nn  # noqa: E402
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32


def train(model, opts, X, T, dims, ranks, env):
    """-> [(W, b) per layer] after training on the bunches of X; ranks > 1: through the data-parallel path"""
    for k in ("KALDI_AMD_DP_EXCHANGE", "KALDI_AMD_DP_CHUNKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if ranks > 1:
        h.call("hipK_comm_init_loopback", ranks)
    try:
        tr = nn.Trainer(model)
        tr.set_train_opts(*opts)
        if ranks > 1:
            tr.set_data_parallel(ranks)
        for s in range(X.shape[0]):
            x = h.DevMatrix.from_numpy(X[s])
            t = h.DevMatrix.from_numpy(T[s], pitched=False)
            tr.train_bunch(x.ptr, x.stride, X.shape[1], dims[0], t.ptr)
        out = [tr.layer_params(2 * k, dims[k + 1], dims[k]) for k in range(len(dims) - 1)]
        tr.close()
    finally:
        if ranks > 1:
            h.call("hipK_comm_destroy")
    return out


def one_case(rs, tmp):
    dims = [int(rs.choice([13, 40, 120, 440]))] + [int(rs.choice([17, 32, 100, 256, 333, 1024])) for _ in range(rs.randint(0, 5))] + \
           [int(rs.choice([9, 50, 201, 1483]))]
    layers = []
    for i in range(len(dims) - 1):
        layers.append(O.BiasedLinearity((rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(f32),
                                        (0.1 * rs.standard_normal(dims[i + 1])).astype(f32)))
        layers.append(O.Sigmoid(dims[i + 1]) if i + 2 < len(dims) else O.Softmax(dims[i + 1]))
    model = os.path.join(tmp, "m.nnet")
    open(model, "wb").write(O.write_nnet(layers, True))
    N, R, steps = int(rs.choice([1, 7, 64, 100, 256])), int(rs.choice([2, 3, 4, 5, 8])), int(rs.randint(1, 4))
    X = rs.standard_normal((steps, N, dims[0])).astype(f32)
    T = rs.randint(0, dims[-1], (steps, N)).astype(np.int32)
    avg = bool(rs.rand() < 0.6)
    lr = float(rs.choice([0.01, 0.05])) if avg else float(rs.choice([0.0005, 0.002]))
    opts = (lr, float(rs.choice([0.0, 0.5, 0.9])), float(rs.choice([0.0, 0.0, 1e-4])), 0.0,
            float(rs.choice([0.0, 0.0, 1.5])), avg, True)
    tag = "dims %s bunch %d ranks %d steps %d opts %s" % (dims, N, R, steps, opts[:5] + (avg,))
    got = {}
    for ex, ch in (("sum", "3"), ("shard", str(rs.choice([1, 2, 3, 8]))), ("gather", "3")):
        got[ex] = train(model, opts, X, T, dims, R, {"KALDI_AMD_DP_EXCHANGE": ex, "KALDI_AMD_DP_CHUNKS": ch})
    for (Wa, ba), (Wb, bb) in zip(got["sum"], got["shard"]):
        if not (np.array_equal(Wa, Wb) and np.array_equal(ba, bb)):
            return tag, "shard differs from sum in bits (max abs %.3g)" % np.abs(Wa - Wb).max()
    one = train(model, opts, np.concatenate([X] * R, axis=1), np.concatenate([T] * R, axis=1), dims, 1, {})
    for ex in ("sum", "gather"):
        for k, ((Wa, ba), (Wb, bb)) in enumerate(zip(got[ex], one)):
            scale = max(float(np.abs(Wb).max()), 1e-3)
            if not np.isfinite(Wb).all() or scale > 1e3:
                return tag, None                      # the random step sizes made this training diverge: nothing to compare
            if not (np.allclose(Wa, Wb, rtol=2e-3, atol=1e-4 * scale) and np.allclose(ba, bb, rtol=2e-3, atol=1e-4 * max(scale, float(np.abs(bb).max())))):
                return tag, "%s differs from one process on the stacked bunch: layer %d max abs %.3g of %.3g" % (ex, k, np.abs(Wa - Wb).max(), scale)
    return tag, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    h.init(-1)
    tmp = tempfile.mkdtemp(prefix="dp_fuzz_")
    bad = 0
    for i in range(args.cases):
        rs = np.random.RandomState(args.seed * 100000 + i)
        tag, msg = one_case(rs, tmp)
        if msg:
            bad += 1
        print("case %d %s: %s" % (i, tag, msg or "ok"), flush=True)
    os.remove(os.path.join(tmp, "m.nnet"))
    os.rmdir(tmp)
    print("%d of %d cases differ" % (bad, args.cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
