#!/usr/bin/env python3
"""The HBM-bound kernels of the hot path on their own: time per launch (HIP events, after the clock has left its idle
state) and achieved GB/s = algorithmic bytes / time, against the MI355X peak of 8 TB/s, at a bunch-sized hidden
layer (256 x 2048), a bunch-sized output layer (256 x 9000) and a frame cache (32768 x 1320).

    python tools/hbm_kernels.py                 timing table (JSON lines on stdout)
    python tools/hbm_kernels.py --pmc           three launches per case and nothing else: run under
                                                rocprofv3 --pmc FETCH_SIZE (or WRITE_SIZE); tools/hbm_summary.py
                                                matches the dispatches to the cases by order
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402

f32 = np.float32
PMC = "--pmc" in sys.argv
h.init(-1)
rs = np.random.RandomState(0)
SHAPES = [(256, 2048), (256, 9000), (32768, 1320), (4321, 19)]   # the last: the reference unit test's AddRowSumMat shape


def cases(R, C):
    x = h.DevMatrix.from_numpy(rs.standard_normal((R, C)).astype(f32))
    y = h.DevMatrix(R, C)
    z = h.DevMatrix.from_numpy(rs.random_sample((R, C)).astype(f32))
    idx = h.DevMatrix.from_numpy(rs.permutation(R).astype(np.int32), pitched=False)
    tgt = h.DevMatrix.from_numpy(rs.randint(0, C, R).astype(np.int32), pitched=False)
    vec = h.DevMatrix(1, C, f32, pitched=False)
    ids = h.DevMatrix(1, R, np.int32, pitched=False)
    lp = h.DevMatrix(1, R, f32, pitched=False)
    cnt = h.DevMatrix(1, 2, np.int64, pitched=False)
    loss = h.DevMatrix(1, 1, np.float64, pitched=False)
    keep = (x, y, z, idx, tgt, vec, ids, lp, cnt, loss)
    e = R * C
    return keep, [
        # name, kernel, algorithmic bytes (reads + writes of the matrices), call
        ("sigmoid", "ew_", 8 * e, lambda: h.call("hipF_sigmoid", None, y.ptr, x.ptr, x.dim)),
        ("diff_sigmoid", "ew_", 12 * e, lambda: h.call("hipF_diff_sigmoid", None, y.ptr, x.ptr, z.ptr, x.dim)),
        ("add_mat (SGD step W -= lr*corr)", "ew_", 12 * e, lambda: h.call("hipF_add_mat", None, f32(-0.008), x.ptr, f32(1.0), y.ptr, x.dim)),
        ("softmax", "softmax", 8 * e, lambda: h.call("hipF_softmax", None, y.ptr, x.ptr, x.dim)),
        ("xent_eval (argmax + diff + loss)", "xent_rows", 8 * e, lambda: h.call("hipF_xent_eval", None, z.ptr, tgt.ptr, y.ptr, lp.ptr, ids.ptr, cnt.ptr, loss.ptr, z.dim)),
        ("find_row_max_id", "find_row_max_id", 4 * e, lambda: h.call("hipF_find_row_max_id", None, x.ptr, None, ids.ptr, x.dim)),
        ("add_row_sum_mat (bias gradient)", "col_sums", 4 * e, lambda: h.call("hipF_add_row_sum_mat", None, f32(1.0), x.ptr, x.dim, f32(0.0), vec.ptr)),
        ("randomize (frame shuffle)", "randomize", 8 * e, lambda: h.call("hipF_randomize", None, y.ptr, x.ptr, idx.ptr, x.dim, x.dim)),
    ]


def ramp():
    a = h.DevMatrix(4096, 4096)
    for _ in range(400):   # ~40 ms of streaming load
        h.call("hipF_scale", None, a.ptr, f32(1.0), a.dim)
    h.synchronize()


if not PMC:
    ramp()
for R, C in SHAPES:
    keep, cs = cases(R, C)
    for name, kernel, nbytes, fn in cs:
        if PMC:
            for _ in range(3):
                fn()
            h.synchronize()
            # (a tall add_row_sum_mat is two kernels per call: slab sums, then the sum of the slabs)
            per_call = 2 if (kernel == "col_sums" and R >= 2048) else 1
            print(json.dumps({"case": name, "shape": [R, C], "kernel": kernel, "launches": 3 * per_call, "calls": 3, "bytes": nbytes}), flush=True)
            continue
        n = 200 if R * C < 4e6 else 40
        for _ in range(n // 4):
            fn()
        t = h.Timer(None)
        t.start()
        for _ in range(n):
            fn()
        us = t.stop() / n * 1e3
        print(json.dumps({"case": name, "shape": [R, C], "kernel": kernel, "us": round(us, 2), "bytes": nbytes,
                          "GBps": round(nbytes / us * 1e-3, 1), "frac_of_8TBps": round(nbytes / us * 1e-3 / 8000.0, 4)}), flush=True)
    del keep, cs
