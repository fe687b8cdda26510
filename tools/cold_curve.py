#!/usr/bin/env python3
"""Per-step GPU time of the first training steps of a cold process (Switchboard-sized net, bunch 256).

Mimics bench.py's start (torch import, set_device, same model and cache), then records one HIP event on
the main stream after every step and prints the deltas:
  phase A: steps 0..N-1 of the cold process, a fence after step 4 (bench.py's --warmup 5)
  phase B: after `idle` seconds of an idle GPU, M more steps (everything loaded and touched by then:
           what is left is the clock ramp from idle)
Usage: python tools/cold_curve.py [--steps 120] [--idle 0.5] [--again 60]
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--idle", type=float, default=0.5)
    ap.add_argument("--again", type=int, default=60)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    torch.cuda.set_device(0)
    import bench
    from kaldi_amd import hip as h
    from kaldi_amd import nnet as nn
    h.init(0)
    dims = bench.WORKLOADS["swbd"]["dims"]
    tmp = tempfile.mkdtemp(prefix="cold_curve_")
    model = os.path.join(tmp, "m.nnet")
    bench.write_model(model, dims, seed=1)
    t_create = time.perf_counter()
    trainer = nn.Trainer(model)
    trainer.set_train_opts(0.008)
    trainer.keep_posteriors(False)
    t_create = time.perf_counter() - t_create
    rs = np.random.RandomState(100)
    B = bench.BUNCH
    rows = 16 * B
    cache = h.DevMatrix.from_numpy(rs.standard_normal((rows, dims[0])).astype(np.float32))
    labels = h.DevMatrix.from_numpy(rs.randint(0, dims[-1], rows).astype(np.int32), pitched=False)

    def step(i):
        b = i % 16
        trainer.train_bunch(ctypes.c_void_p(cache.ptr.value + b * B * cache.stride * 4), cache.stride, B, dims[0],
                            ctypes.c_void_p(labels.ptr.value + b * B * 4))

    def events(n):
        ev = []
        for _ in range(n):
            e = ctypes.c_void_p()
            h.call("hipK_event_create", ctypes.byref(e))
            ev.append(e)
        return ev

    def run(n, fence_after=None):
        ev = events(n + 1)
        host = []
        nn.synchronize()
        h.call("hipK_event_record", ev[0], None)
        for i in range(n):
            t0 = time.perf_counter()
            step(i)
            h.call("hipK_event_record", ev[i + 1], None)
            host.append((time.perf_counter() - t0) * 1e3)
            if fence_after is not None and i == fence_after:
                nn.synchronize()
                torch.cuda.synchronize()
        nn.synchronize()
        ms = []
        for i in range(n):
            v = ctypes.c_float()
            h.call("hipK_event_elapsed_ms", ctypes.byref(v), ev[i], ev[i + 1])
            ms.append(v.value)
        return ms, host

    a_ms, a_host = run(args.steps, fence_after=4)
    time.sleep(args.idle)
    b_ms, b_host = run(args.again)
    out = {"create_s": round(t_create, 3), "A_gpu_ms": [round(x, 4) for x in a_ms], "A_host_ms": [round(x, 4) for x in a_host],
           "B_gpu_ms": [round(x, 4) for x in b_ms], "B_host_ms": [round(x, 4) for x in b_host]}

    def show(tag, v):
        print(tag, " ".join("%.3f" % x for x in v))
    show("A gpu ms :", a_ms)
    show("A host ms:", a_host)
    show("B gpu ms :", b_ms)
    show("B host ms:", b_host)
    print("A: steps 5..24 sum %.3f ms (bench 20/5 window), steps 100..119 sum %.3f ms" %
          (sum(a_ms[5:25]), sum(a_ms[100:120]) if len(a_ms) >= 120 else float("nan")))
    print("B: steps 0..19 sum %.3f ms, last 20 sum %.3f ms" % (sum(b_ms[:20]), sum(b_ms[-20:])))
    if args.out:
        json.dump(out, open(args.out, "w"))
    trainer.close()


if __name__ == "__main__":
    main()
