"""hipF_xent_eval_static -- the entry point that allocates nothing (its arrival counter and per-row scratch
live in a __device__ array of the library) -- against hipF_xent_eval, through the C ABI, bit for bit, on
two calls in a row, the second one smaller; also against the oracle, and the size limit.
Collected last (file name) so that a failure here hides no other test."""
import ctypes

import numpy as np
import pytest

from oracle import nnet1_oracle as O

f32 = np.float32
pytestmark = pytest.mark.gpu


def _case(h, name, R, C, seed, with_row_outputs):
    rs = np.random.RandomState(seed)
    post = O.softmax((2.0 * rs.standard_normal((R, C))).astype(f32))
    tgt = rs.randint(0, C, R).astype(np.int32)
    tgt[: R // 3] = post[: R // 3].argmax(1)
    post[0, tgt[0]] = 0.0  # the 1e-20 floor
    dp = h.DevMatrix.from_numpy(post)
    dt = h.DevMatrix.from_numpy(tgt, pitched=False)
    dd = h.DevMatrix(R, C)
    cnt = h.DevMatrix(1, 2, np.int64, pitched=False)
    loss = h.DevMatrix(1, 1, np.float64, pitched=False)
    lp = h.DevMatrix(1, R, f32, pitched=False) if with_row_outputs else None
    mid = h.DevMatrix(1, R, np.int32, pitched=False) if with_row_outputs else None
    h.call(name, None, dp.ptr, dt.ptr, dd.ptr, lp.ptr if lp else None, mid.ptr if mid else None, cnt.ptr, loss.ptr,
           dp.dim)
    out = dict(diff=dd.numpy(), cnt=cnt.numpy()[0].tolist(), loss=float(loss.numpy()[0, 0]))
    if with_row_outputs:
        out["lp"], out["mid"] = lp.numpy()[0], mid.numpy()[0]
    return out, post, tgt


def test_zz_xent_eval_static_matches_xent_eval(hip):
    h = hip
    assert hasattr(h.lib, "hipF_xent_eval_static")
    # first call large, second call smaller (the scratch keeps the first call's contents in between)
    for R, C, seed, rows_out in ((300, 9000, 5, False), (77, 1483, 6, False), (256, 2048, 7, True), (3, 17, 8, True)):
        ref, post, tgt = _case(h, "hipF_xent_eval", R, C, seed, rows_out)
        got, _, _ = _case(h, "hipF_xent_eval_static", R, C, seed, rows_out)
        assert np.array_equal(got["diff"], ref["diff"]), (R, C)
        assert got["cnt"] == ref["cnt"], (R, C)
        assert got["loss"] == ref["loss"], (R, C)   # same kernels, same summation order: identical bits
        if rows_out:
            assert np.array_equal(got["lp"], ref["lp"]) and np.array_equal(got["mid"], ref["mid"])
        x = O.Xent()
        rdiff, _, _, rcorrect = x.eval_vec(post, tgt)
        np.testing.assert_allclose(got["diff"], rdiff, rtol=1e-6, atol=1e-7)
        assert got["cnt"] == [rcorrect, R]                      # frame-accuracy counts exactly
        assert abs(got["loss"] - x.loss) <= 1e-5 * abs(x.loss)


def test_zz_xent_eval_static_accumulates_and_rejects_oversize(hip):
    h = hip
    R, C = 64, 500
    rs = np.random.RandomState(9)
    post = O.softmax(rs.standard_normal((R, C)).astype(f32))
    tgt = rs.randint(0, C, R).astype(np.int32)
    dp, dt, dd = h.DevMatrix.from_numpy(post), h.DevMatrix.from_numpy(tgt, pitched=False), h.DevMatrix(R, C)
    cnt = h.DevMatrix(1, 2, np.int64, pitched=False)
    loss = h.DevMatrix(1, 1, np.float64, pitched=False)
    for _ in range(3):   # back to back on one stream: counters and loss accumulate like hipF_xent_eval's
        h.call("hipF_xent_eval_static", None, dp.ptr, dt.ptr, dd.ptr, None, None, cnt.ptr, loss.ptr, dp.dim)
    x = O.Xent()
    _, _, _, rcorrect = x.eval_vec(post, tgt)
    assert cnt.numpy()[0].tolist() == [3 * rcorrect, 3 * R]
    assert abs(loss.numpy()[0, 0] - 3 * x.loss) <= 1e-5 * abs(3 * x.loss)
    # beyond the stated maximum: a clear error, nothing launched
    big = h.MatrixDim(65537, 4, 4)
    rc = h.lib.hipF_xent_eval_static(None, dp.ptr, dt.ptr, dd.ptr, None, None, cnt.ptr, loss.ptr, big)
    assert rc == 100004   # KH_ERR_TOO_LARGE
    assert b"65536" in h.lib.hipK_error_string(rc)
    assert cnt.numpy()[0].tolist() == [3 * rcorrect, 3 * R]
