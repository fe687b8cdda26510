"""GPU parity tests of the C-ABI kernels (include/kaldi_hip.h) against the CPU oracle.

Shapes follow the reference's own cudamatrix unit test (src/cudamatrix/cuda-matrix-test.cc:
100x111 element-wise, 4321x19 / 1000x990 reductions, 200x100 GEMMs) plus the RM/SWBD layer
shapes of the hot path.  Tolerances: the reference test uses 1e-3 relative
(cuda-matrix-test.cc:77-86); here element-wise ops are held to 1e-6, reductions/GEMMs to 1e-5
of the row scale, integer outputs exactly.
"""
import ctypes

import numpy as np
import pytest

from oracle import nnet1_oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32
P = ctypes.c_void_p


def dev(h, a, pitched=True):
    return h.DevMatrix.from_numpy(np.asarray(a), pitched)


def rnd(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(*shape) * scale).astype(f32)


def close(a, b, rtol=1e-6, atol=1e-6):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(100, 111), (256, 2048), (7, 3), (1, 1)])
def test_elementwise_unary_binary(hip, shape):
    h = hip
    x = rnd(shape, 1)
    y = rnd(shape, 2)
    # sigmoid / diff_sigmoid
    dx, dy = dev(h, x), dev(h, np.zeros(shape, f32))
    h.call("hipF_sigmoid", None, dy.ptr, dx.ptr, dx.dim)
    s = O.sigmoid(x)
    close(dy.numpy(), s, 2e-6, 1e-7)
    de, dout = dev(h, y), dev(h, np.zeros(shape, f32))
    h.call("hipF_diff_sigmoid", None, dout.ptr, de.ptr, dy.ptr, dy.dim)
    close(dout.numpy(), O.diff_sigmoid(y, dy.numpy()), 2e-6, 1e-7)
    # add_mat with beta (SGD update shape) and beta == 0 on NaN-filled dst
    d1 = dev(h, x)
    h.call("hipF_add_mat", None, f32(-0.008), dev(h, y).ptr, f32(1.0), d1.ptr, d1.dim)
    close(d1.numpy(), (f32(-0.008) * y + x).astype(f32))
    dn = dev(h, np.full(shape, np.nan, f32))
    h.call("hipF_add_mat", None, f32(2.0), dev(h, y).ptr, f32(0.0), dn.ptr, dn.dim)
    close(dn.numpy(), 2 * y)
    # scale, add_const, apply_exp, apply_log, invert, power, floor, truncate, binarize, round
    for name, args, ref in [
        ("hipF_scale", (f32(0.5),), x * f32(0.5)),
        ("hipF_add_const", (f32(1.5),), x + f32(1.5)),
        ("hipF_apply_exp", (), np.exp(x)),
        ("hipF_binarize", (f32(0.1),), (x > 0.1).astype(f32)),
        ("hipF_round", (), np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))),
        ("hipF_apply_floor", (f32(-0.2),), np.maximum(x, f32(-0.2))),
        ("hipF_apply_truncate", (f32(-0.3), f32(0.4)), np.clip(x, f32(-0.3), f32(0.4))),
    ]:
        d = dev(h, x)
        h.call(name, None, d.ptr, *args, d.dim)
        close(d.numpy(), ref.astype(f32), 2e-6, 1e-7)
    pos = np.abs(x) + f32(0.1)
    d = dev(h, pos)
    h.call("hipF_apply_log", None, d.ptr, d.dim)
    close(d.numpy(), np.log(pos), 2e-6, 2e-7)
    d = dev(h, pos)
    h.call("hipF_invert_elements", None, d.ptr, d.dim)
    close(d.numpy(), 1 / pos, 2e-6)
    d = dev(h, pos)
    h.call("hipF_power", None, d.ptr, f32(1.7), d.dim)
    close(d.numpy(), np.power(pos, f32(1.7)), 5e-6)
    # mul / div elements
    d = dev(h, x)
    h.call("hipF_mul_elements", None, d.ptr, dev(h, y).ptr, d.dim)
    close(d.numpy(), x * y)
    d = dev(h, x)
    h.call("hipF_div_elements", None, d.ptr, dev(h, pos).ptr, d.dim)
    close(d.numpy(), x / pos, 2e-6)
    # relu family
    d = dev(h, np.zeros(shape, f32))
    h.call("hipF_relu", None, d.ptr, dx.ptr, dx.dim)
    close(d.numpy(), np.maximum(x, 0))
    h.call("hipF_diff_relu", None, d.ptr, de.ptr, dx.ptr, dx.dim)
    close(d.numpy(), np.where(x > 0, y, 0))
    h.call("hipF_softrelu", None, d.ptr, dx.ptr, dx.dim)
    close(d.numpy(), np.where(x > 4, x, np.log1p(np.exp(x))), 3e-6, 1e-6)
    h.call("hipF_diff_softrelu", None, d.ptr, de.ptr, dx.ptr, dx.dim)
    close(d.numpy(), np.where(x > 4, y, y * np.exp(x) / (1 + np.exp(x))), 3e-6, 1e-6)
    # log_add_exp_mat
    d = dev(h, x)
    h.call("hipF_log_add_exp_mat", None, dev(h, y).ptr, d.ptr, d.dim)
    close(d.numpy(), np.logaddexp(x.astype(np.float64), y.astype(np.float64)).astype(f32), 2e-6, 1e-7)


@pytest.mark.parametrize("shape", [(100, 111), (256, 1483), (33, 4)])
def test_vector_broadcast_ops(hip, shape):
    h = hip
    R, C = shape
    x = rnd(shape, 3)
    col = rnd((C,), 4)
    row = rnd((R,), 5)
    dcol, drow = dev(h, col, False), dev(h, row, False)
    for name, vec, ref in [
        ("hipF_mul_cols_vec", dcol, x * col[None, :]),
        ("hipF_mul_rows_vec", drow, x * row[:, None]),
        ("hipF_div_cols_vec", dcol, x * (f32(1) / col)[None, :]),
        ("hipF_div_rows_vec", drow, x * (f32(1) / row)[:, None]),
    ]:
        d = dev(h, x)
        h.call(name, None, d.ptr, vec.ptr, d.dim)
        close(d.numpy(), ref.astype(f32), 2e-6, 1e-7)
    d = dev(h, x)
    h.call("hipF_add_vec_to_rows", None, f32(0.5), dcol.ptr, f32(2.0), d.ptr, d.dim)
    close(d.numpy(), f32(0.5) * col[None, :] + f32(2) * x)
    d = dev(h, np.full(shape, np.nan, f32))
    h.call("hipF_add_vec_to_rows", None, f32(1.0), dcol.ptr, f32(0.0), d.ptr, d.dim)  # bias pre-copy
    close(d.numpy(), np.broadcast_to(col, shape))
    d = dev(h, x)
    h.call("hipF_add_vec_to_cols", None, f32(0.5), drow.ptr, f32(2.0), d.ptr, d.dim)
    close(d.numpy(), f32(0.5) * row[:, None] + f32(2) * x)
    # partial ops
    d = dev(h, x)
    h.call("hipF_part_set_const", None, d.ptr, f32(9.0), d.dim, 1, 2, 1, 2)
    ref = x.copy()
    ref[1:3, 1:3] = 9
    close(d.numpy(), ref)
    if C > 8:
        d = dev(h, x)
        sub = rnd((5,), 9)
        h.call("hipF_add_vec_to_partial_rows", None, f32(2.0), 3, dev(h, sub, False).ptr, 5, f32(0.5), d.ptr, d.dim)
        ref = x.copy()
        ref[:, 3:8] = f32(2) * sub[None, :] + f32(0.5) * x[:, 3:8]
        close(d.numpy(), ref)
        a = rnd((4, 6), 10)
        d = dev(h, x)
        da = dev(h, a)
        h.call("hipF_part_add_mat", None, f32(2.0), da.ptr, da.dim, 2, 1, f32(0.5), d.ptr, d.dim)
        ref = x.copy()
        ref[2:6, 1:7] = f32(2) * a + f32(0.5) * x[2:6, 1:7]
        close(d.numpy(), ref)
    di = h.DevMatrix(1, R, np.int32, pitched=False)
    h.call("hipI32_set_const", None, di.ptr, -1, di.dim)
    assert (di.numpy() == -1).all()


@pytest.mark.parametrize("shape", [(4321, 19), (1000, 990), (256, 2048), (256, 9000), (5, 1), (5000, 1320), (2048, 257), (9001, 1483)])
def test_row_col_sums(hip, shape):
    h = hip
    R, C = shape
    x = rnd(shape, 6)
    v0 = rnd((C,), 7)
    d = dev(h, x)
    dv = dev(h, v0, False)
    h.call("hipF_add_row_sum_mat", None, f32(0.7), d.ptr, d.dim, f32(0.5), dv.ptr)
    ref = O.add_row_sum_mat(0.7, x, 0.5, v0)
    close(dv.numpy()[0], ref, 1e-5, 1e-5 * np.sqrt(R))
    v1 = rnd((R,), 8)
    dv = dev(h, v1, False)
    h.call("hipF_add_col_sum_mat", None, f32(0.7), d.ptr, d.dim, f32(0.5), dv.ptr)
    close(dv.numpy()[0], O.add_col_sum_mat(0.7, x, 0.5, v1), 1e-5, 1e-5 * np.sqrt(C))
    dv = dev(h, np.full((R,), np.nan, f32), False)
    h.call("hipF_add_col_sum_mat", None, f32(1.0), d.ptr, d.dim, f32(0.0), dv.ptr)
    close(dv.numpy()[0], O.add_col_sum_mat(1.0, x, 0.0, np.zeros(R, f32)), 1e-5, 1e-5 * np.sqrt(C))


@pytest.mark.parametrize("shape", [(100, 111), (256, 1483), (256, 9000), (3, 13001), (2, 1), (4100, 1320), (3001, 19), (5, 2048), (7, 2049)])
def test_softmax(hip, shape):
    h = hip
    x = rnd(shape, 11, 3.0)
    d = dev(h, x)
    y = dev(h, np.zeros(shape, f32))
    h.call("hipF_softmax", None, y.ptr, d.ptr, d.dim)
    got = y.numpy()
    ref = O.softmax(x)
    # north_star: posteriors within 1e-4 relative
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-30)
    close(got.sum(1), np.ones(shape[0]), 1e-5)
    h.call("hipF_softmax", None, d.ptr, d.ptr, d.dim)  # in place
    np.testing.assert_allclose(d.numpy(), ref, rtol=1e-4, atol=1e-30)
    # a row's bits do not depend on how many rows share the call
    if shape[0] > 2:
        sub = dev(h, x[1:3])
        h.call("hipF_softmax", None, sub.ptr, sub.ptr, sub.dim)
        assert np.array_equal(sub.numpy(), got[1:3])


def test_softmax_part_and_argmax_ties(hip):
    h = hip
    x = rnd((37, 600), 12)
    x[3, 10] = x[3, 500] = 10.0      # tie: the first maximum wins
    x[4, 0] = x[4, 599] = 11.0
    x[5, :] = 2.5                    # all equal -> index 0
    x[6, :] = -2e21                  # nothing exceeds the -1e21 start value -> -1
    d = dev(h, x)
    ids = h.DevMatrix(1, 37, np.int32, pitched=False)
    vals = h.DevMatrix(1, 37, f32, pitched=False)
    h.call("hipF_find_row_max_id", None, d.ptr, vals.ptr, ids.ptr, d.dim)
    ref = O.find_row_max_id(x)
    got = ids.numpy()[0]
    assert (got == ref).all()
    assert got[3] == 10 and got[4] == 0 and got[5] == 0 and got[6] == -1
    ok = ref >= 0
    close(vals.numpy()[0][ok], x[np.arange(37)[ok], ref[ok]])
    ids2 = dev(h, np.where(ref < 0, 0, ref).astype(np.int32), False)
    y = dev(h, np.zeros_like(x))
    h.call("hipF_softmax_part", None, d.ptr, ids2.ptr, y.ptr, d.dim)
    xx = x[:6]
    close(y.numpy()[:6], np.exp(xx - xx.max(1, keepdims=True)), 3e-6, 1e-30)
    # the same rows inside a tall matrix (one wave per row there): same answers
    for cols in (600, 1320, 2048):
        tall = rnd((3000, cols), 15)
        tall[:37, :600] = x
        tall[:37, 600:] = -3e21
        tall[2999, cols - 1] = tall[2999, 5] = 50.0
        dtall = dev(h, tall)
        idt, vt = h.DevMatrix(1, 3000, np.int32, pitched=False), h.DevMatrix(1, 3000, f32, pitched=False)
        h.call("hipF_find_row_max_id", None, dtall.ptr, vt.ptr, idt.ptr, dtall.dim)
        reft = O.find_row_max_id(tall)
        assert (idt.numpy()[0] == reft).all() and idt.numpy()[0][2999] == 5 and idt.numpy()[0][6] == -1


@pytest.mark.parametrize("shape", [(256, 1483), (256, 9000), (16, 17), (4100, 1320), (1026, 2048), (3000, 21)])
def test_xent(hip, shape):
    h = hip
    R, C = shape
    rs = np.random.RandomState(13)
    post = O.softmax(rnd(shape, 14, 2.0))
    tgt = rs.randint(0, C, R).astype(np.int32)
    tgt[: R // 4] = post[: R // 4].argmax(1)   # make a known share of the frames correct
    post[1, tgt[1]] = 0.0                       # exercises the 1e-20 floor
    # separate kernel
    d = dev(h, post)
    dt = dev(h, tgt, False)
    lp = h.DevMatrix(1, R, f32, pitched=False)
    h.call("hipF_diff_xent", None, dt.ptr, d.ptr, lp.ptr, d.dim)
    rdiff, rlp = O.diff_xent(tgt, post)
    close(d.numpy(), rdiff)
    close(lp.numpy()[0], rlp, 2e-6)
    # fused evaluation, out of place and in place, two calls accumulate
    x = O.Xent()
    rdiff, rlp, rmax, rcorrect = x.eval_vec(post, tgt)
    cnt = h.DevMatrix(1, 2, np.int64, pitched=False)
    loss = h.DevMatrix(1, 1, np.float64, pitched=False)
    dp, dd = dev(h, post), dev(h, np.zeros(shape, f32))
    mid = h.DevMatrix(1, R, np.int32, pitched=False)
    h.call("hipF_xent_eval", None, dp.ptr, dt.ptr, dd.ptr, lp.ptr, mid.ptr, cnt.ptr, loss.ptr, dp.dim)
    close(dd.numpy(), rdiff)
    assert (mid.numpy()[0] == rmax).all()
    assert cnt.numpy()[0].tolist() == [rcorrect, R]           # frame-accuracy counts exactly
    assert abs(loss.numpy()[0, 0] - x.loss) <= 1e-5 * abs(x.loss)
    h.call("hipF_xent_eval", None, dp.ptr, dt.ptr, dp.ptr, None, None, cnt.ptr, loss.ptr, dp.dim)  # in place
    close(dp.numpy(), rdiff)
    assert cnt.numpy()[0].tolist() == [2 * rcorrect, 2 * R]
    assert abs(loss.numpy()[0, 0] - 2 * x.loss) <= 1e-5 * abs(2 * x.loss)


@pytest.mark.parametrize("rows,cols", [(256, 440), (1, 7), (3, 2048), (1000, 33)])
def test_mse_eval(hip, rows, cols):
    """Mse::Eval (nnet-loss.cc:123-158) in one launch: row sums, then the workgroup that arrives last adds them up.
    One row, more rows than that workgroup has threads, ragged widths; the loss accumulates over calls (the arrival
    counter is back at zero after each) and the optional outputs may be left out."""
    h = hip
    a, b = rnd((rows, cols), 15), rnd((rows, cols), 16)
    rdiff, rrow, rloss = O.mse_eval(a, b)
    da, db, dd = dev(h, a), dev(h, b), dev(h, np.zeros_like(a))
    row = h.DevMatrix(1, rows, f32, pitched=False)
    loss = h.DevMatrix(1, 1, np.float64, pitched=False)
    h.call("hipF_mse_eval", None, da.ptr, db.ptr, dd.ptr, row.ptr, loss.ptr, da.dim)
    close(dd.numpy(), rdiff)
    close(row.numpy()[0], rrow, 1e-5)
    assert abs(loss.numpy()[0, 0] - rloss) < 1e-5 * rloss
    first = loss.numpy()[0, 0]
    for _ in range(2):
        h.call("hipF_mse_eval", None, da.ptr, db.ptr, None, None, loss.ptr, da.dim)
    assert abs(loss.numpy()[0, 0] - 3 * first) <= 1e-12 * 3 * first


def test_gathers(hip):
    h = hip
    rs = np.random.RandomState(17)
    src = rnd((1000, 440), 18)
    idx = rs.permutation(1000)[:777].astype(np.int32)
    d, di = dev(h, src), dev(h, idx, False)
    out = dev(h, np.zeros((1000, 440), f32))
    dout = out.dim
    dout.rows = 777
    h.call("hipF_randomize", None, out.ptr, d.ptr, di.ptr, dout, d.dim)
    assert (out.numpy()[:777] == O.randomize(src, idx)).all()     # bit exact: it is a copy
    assert (out.numpy()[777:] == 0).all()
    # unaligned width takes the scalar path
    src2 = rnd((50, 39), 19)
    idx2 = rs.permutation(50).astype(np.int32)
    d2, out2 = dev(h, src2, False), dev(h, np.zeros((50, 39), f32), False)
    h.call("hipF_randomize", None, out2.ptr, d2.ptr, dev(h, idx2, False).ptr, out2.dim, d2.dim)
    assert (out2.numpy() == src2[idx2]).all()
    # splice +-5 like the recipes (40 -> 440)
    feats = rnd((123, 40), 20)
    offs = np.arange(-5, 6).astype(np.int32)
    df, do = dev(h, feats), dev(h, np.zeros((123, 440), f32))
    h.call("hipF_expand", None, do.ptr, df.ptr, dev(h, offs, False).ptr, do.dim, df.dim)
    assert (do.numpy() == O.expand(feats, offs)).all()
    # column gather with one bad index -> +inf
    cidx = rs.randint(0, 40, 25).astype(np.int32)
    cidx[7] = 40
    dc = dev(h, np.zeros((123, 25), f32))
    h.call("hipF_copy", None, dc.ptr, df.ptr, dev(h, cidx, False).ptr, dc.dim, df.dim)
    got = dc.numpy()
    good = np.arange(25) != 7
    assert (got[:, good] == feats[:, cidx[good]]).all() and np.isposinf(got[:, 7]).all()


def test_host_copies_of_narrow_matrices_do_not_go_row_by_row(hip):
    """A pitched matrix copied from or to pageable host memory costs the runtime 9 us per ROW (3.4 ms for a 366 x 40
    utterance); hipK_memcpy_2d sends every copy of more than one row through its pinned staging buffers instead
    (about 20 us).  Values round-trip; the time is checked with a wide margin."""
    import time
    rs = np.random.RandomState(0)
    for rows, cols in ((366, 40), (100, 13), (30, 120), (1, 40), (400, 1)):
        a = rs.standard_normal((rows, cols)).astype(np.float32)
        d = hip.DevMatrix(rows, cols)
        d.set(a)
        np.testing.assert_array_equal(d.numpy(), a)
        times = []
        for _ in range(15):
            t0 = time.perf_counter()
            d.set(a)
            times.append(time.perf_counter() - t0)
        if rows >= 100:  # (row by row these took 0.9 and 3.4 ms; through the staging buffers about 0.02 ms)
            assert sorted(times)[len(times) // 2] < 4e-4, (rows, cols, sorted(times))


def test_cache_sized_streaming_kernels(hip):
    """The flat forms that element-wise passes and the frame shuffle take from 2^20 16-byte elements up (a whole
    frame cache): the shuffle bit for bit, the element-wise results against numpy,
    a row count that does not fill the last thread's four elements, an index outside the source (row left alone)."""
    h = hip
    rs = np.random.RandomState(23)
    R, C = 3301, 1320                                   # 3301 * 330 = 1,089,330 elements of 16 bytes
    src = rnd((R, C), 24)
    idx = rs.permutation(R).astype(np.int32)
    idx[5] = R + 3                                      # outside x: row 5 of the output keeps its contents
    d, di = dev(h, src), dev(h, idx, False)
    out = dev(h, np.full((R, C), 7.0, f32))
    h.call("hipF_randomize", None, out.ptr, d.ptr, di.ptr, out.dim, d.dim)
    got = out.numpy()
    ok = np.arange(R) != 5
    assert (got[ok] == src[idx[ok]]).all() and (got[5] == 7.0).all()
    # element-wise: y = a*x + b*y, y = x*(1-x)*e, column-vector and row-vector operands
    x, y = rnd((R, C), 25), rnd((R, C), 26)
    dx, dy = dev(h, x), dev(h, y)
    h.call("hipF_add_mat", None, f32(-0.25), dx.ptr, f32(0.5), dy.ptr, dx.dim)
    close(dy.numpy(), f32(-0.25) * x + f32(0.5) * y)
    e, yy = rnd((R, C), 27), np.random.RandomState(28).random_sample((R, C)).astype(f32)
    de, dyy, dd = dev(h, e), dev(h, yy), dev(h, np.zeros((R, C), f32))
    h.call("hipF_diff_sigmoid", None, dd.ptr, de.ptr, dyy.ptr, de.dim)
    close(dd.numpy(), yy * (f32(1.0) - yy) * e)
    col, row = rnd((C,), 29), rnd((R,), 30)
    dm = dev(h, x)
    h.call("hipF_add_vec_to_rows", None, f32(2.0), dev(h, col, False).ptr, f32(1.0), dm.ptr, dm.dim)
    close(dm.numpy(), f32(2.0) * col[None, :] + x)
    dm = dev(h, x)
    h.call("hipF_add_vec_to_cols", None, f32(2.0), dev(h, row, False).ptr, f32(1.0), dm.ptr, dm.dim)
    close(dm.numpy(), f32(2.0) * row[:, None] + x)
    ds = dev(h, np.zeros((R, C), f32))
    h.call("hipF_sigmoid", None, ds.ptr, dx.ptr, dx.dim)
    close(ds.numpy(), O.sigmoid(x))


@pytest.mark.parametrize("rows,cols", [(58, 209), (1, 1), (300, 1), (1500, 255), (20, 1941), (40, 9001)])
def test_row_kernels_write_their_window_only(hip, rows, cols):
    """softmax, xent_eval and the fused output layer on a WINDOW of a larger buffer whose row length is not a multiple
    of 4: the columns behind the window's last one belong to somebody else and must come back untouched (the
    16-byte row stores once ran to the end of the last group of 4) -- found by tests/fuzz/op_fuzz.py."""
    import ctypes
    h = hip
    rs = np.random.RandomState(31)
    FILL = f32(-777.25)
    c0, pad = 4, 7
    S = c0 + cols + pad
    S += (-S) % 4                                       # 16-byte row pitch, so the 16-byte paths are taken
    x = (2 * rs.standard_normal((rows, cols))).astype(f32)

    def window(data=None):
        host = np.full((rows + 2, S), FILL, f32)
        if data is not None:
            host[1:1 + rows, c0:c0 + cols] = data
        d = h.DevMatrix.from_numpy(host, pitched=False)
        return d, ctypes.c_void_p(d.ptr.value + 4 * (S + c0))

    def outside_untouched(d):
        got = d.numpy().copy()
        win = got[1:1 + rows, c0:c0 + cols].copy()
        got[1:1 + rows, c0:c0 + cols] = FILL
        assert (got == FILL).all(), np.argwhere(got != FILL)[:4]
        return win

    dim = h.MatrixDim(rows, cols, S)
    dx, px = window(x)
    dy, py = window()
    h.call("hipF_softmax", None, py, px, dim)
    p = O.softmax(x)
    close(outside_untouched(dy), p, 1e-4, 1e-7)
    tgt = rs.randint(0, cols, rows).astype(np.int32)
    dp, pp = window(p)
    dq, pq = window()
    cnt, loss = h.DevMatrix(1, 2, np.int64, pitched=False), h.DevMatrix(1, 1, np.float64, pitched=False)
    h.call("hipF_xent_eval", None, pp, dev(h, tgt, False).ptr, pq, None, None, cnt.ptr, loss.ptr, dim)
    want = p.copy()
    want[np.arange(rows), tgt] -= 1
    close(outside_untouched(dq), want, 1e-6, 1e-7)
    if cols <= 2100:
        I = 24
        W, b = (0.3 * rs.standard_normal((cols, I))).astype(f32), (0.1 * rs.standard_normal(cols)).astype(f32)
        xin = rs.standard_normal((rows, I)).astype(f32)
        dpo, ppo = window()
        ddi, pdi = window()
        dW, din = dev(h, W), dev(h, xin)
        h.call("hipF_affine_forward_xent", None, 112, rows, cols, I, din.ptr, din.stride, dW.ptr, dW.stride, dev(h, b, False).ptr,
               ppo, S, 1, dev(h, tgt, False).ptr, pdi, S, None, None, cnt.ptr, loss.ptr)
        pr = O.softmax((xin.astype(np.float64) @ W.T.astype(np.float64) + b).astype(f32))
        close(outside_untouched(dpo), pr, 2e-4, 1e-7)
        wd = pr.copy()
        wd[np.arange(rows), tgt] -= 1
        close(outside_untouched(ddi), wd, 2e-4, 2e-6)


def test_hybrid_taus_bit_exact_vs_oracle_parity_unpinned(hip):
    """HybridTaus streams against the oracle's restatement of cu-randkernels.cu:44-94, bit for bit.  Parity unpinned:
    the reference holds no vector for its generator and its kernels need nvcc, so nothing it produced can be compared."""
    h = hip
    rs = np.random.RandomState(21)
    shape = (64, 200)
    z = [rs.randint(128, 2 ** 31 - 1, shape).astype(np.uint32) for _ in range(4)]
    dz = [dev(h, a) for a in z]
    out = dev(h, np.zeros(shape, f32))
    zz = z
    for _ in range(3):  # state carries over between calls
        h.call("hipF_rand", None, out.ptr, dz[0].ptr, dz[1].ptr, dz[2].ptr, dz[3].ptr, out.dim)
        ref, zz = O.hybrid_taus(*zz)
        assert (out.numpy() == ref).all()
    for a, b in zip(dz, zz):
        assert (a.numpy() == b).all()
    # gaussian: same uniforms, Box-Muller in float -> a few ulp
    h.call("hipF_gauss_rand", None, out.ptr, dz[0].ptr, dz[1].ptr, dz[2].ptr, dz[3].ptr, out.dim)
    ref, zz2 = O.box_muller(*zz)
    close(out.numpy(), ref, 1e-4, 1e-5)
    for a, b in zip(dz, zz2):
        assert (a.numpy() == b).all()
    # fused binarize == rand + compare
    probs = rs.rand(*shape).astype(f32)
    st = dev(h, np.zeros(shape, f32))
    h.call("hipF_binarize_probs_rand", None, st.ptr, dev(h, probs).ptr, dz[0].ptr, dz[1].ptr, dz[2].ptr, dz[3].ptr,
           st.dim)
    u, zz3 = O.hybrid_taus(*zz2)
    assert (st.numpy() == O.binarize_probs(probs, u)).all()
    # unfused binarize
    du = dev(h, u)
    h.call("hipF_binarize_probs", None, st.ptr, dev(h, probs).ptr, du.ptr, st.dim)
    assert (st.numpy() == O.binarize_probs(probs, u)).all()
    # statistics of a long run: mean 0.5, var 1/12
    big = (512, 2048)
    zb = [dev(h, rs.randint(128, 2 ** 31 - 1, big).astype(np.uint32)) for _ in range(4)]
    ob = dev(h, np.zeros(big, f32))
    h.call("hipF_rand", None, ob.ptr, zb[0].ptr, zb[1].ptr, zb[2].ptr, zb[3].ptr, ob.dim)
    v = ob.numpy().astype(np.float64)
    assert abs(v.mean() - 0.5) < 2e-3 and abs(v.var() - 1 / 12) < 2e-3 and v.min() > 0 and v.max() < 1
    m0 = rnd(big, 22)
    om = dev(h, m0)
    h.call("hipF_add_gauss_noise", None, om.ptr, f32(1.0), zb[0].ptr, zb[1].ptr, zb[2].ptr, zb[3].ptr, om.dim)
    g = om.numpy().astype(np.float64) - m0
    assert abs(g.mean()) < 5e-3 and abs(g.var() - 1.0) < 1e-2


GEMM_SHAPES = [
    # (M, N, K)
    (200, 200, 100),          # the reference's own test shape
    (256, 1024, 440), (256, 1483, 1024), (256, 1024, 1483), (1024, 440, 256), (1483, 1024, 256),
    (256, 2048, 2048), (2048, 2048, 256), (256, 9000, 2048),
    (1, 1, 1), (33, 65, 7), (64, 64, 32), (130, 70, 33),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm(hip, M, N, K, ta, tb):
    h = hip
    if M * N * K > 3e8 and (ta, tb) == (1, 1):
        pytest.skip("TT does not occur on the path; small shapes cover it")
    A = rnd((K, M) if ta else (M, K), 30)
    B = rnd((N, K) if tb else (K, N), 31)
    C0 = rnd((M, N), 32)
    dA, dB, dC = dev(h, A), dev(h, B), dev(h, C0)
    scale = np.sqrt(K)
    for alpha, beta in [(1.0, 0.0), (0.5, 1.0), (-0.25, 0.9)]:
        dC.set(C0 if beta != 0 else np.full((M, N), np.nan, f32))
        h.call("hipF_gemm", None, h.TRANS if ta else h.NO_TRANS, h.TRANS if tb else h.NO_TRANS, M, N, K, f32(alpha),
               dA.ptr, dA.stride, dB.ptr, dB.stride, f32(beta), dC.ptr, dC.stride)
        ref = O.gemm(alpha, A, ta, B, tb, beta, C0)
        np.testing.assert_allclose(dC.numpy(), ref, rtol=1e-5, atol=6e-6 * scale)


def test_gemm_unaligned_and_exact(hip):
    """Tight (unpitched) operands take the scalar-load path; small integers make the sum exact,
    and an asymmetric B catches a transposed output map."""
    h = hip
    rs = np.random.RandomState(33)
    M, N, K = 70, 45, 37
    A = rs.randint(-3, 4, (M, K)).astype(f32)
    B = rs.randint(-3, 4, (K, N)).astype(f32)
    dA, dB, dC = dev(h, A, False), dev(h, B, False), dev(h, np.zeros((M, N), f32), False)
    h.call("hipF_gemm", None, h.NO_TRANS, h.NO_TRANS, M, N, K, f32(1), dA.ptr, K, dB.ptr, N, f32(0), dC.ptr, N)
    assert (dC.numpy() == A @ B).all()
    I = np.eye(64, dtype=f32)
    Bm = (np.arange(64 * 64).reshape(64, 64) % 251).astype(f32)
    dI, dB2, dC2 = dev(h, I), dev(h, Bm), dev(h, np.zeros((64, 64), f32))
    h.call("hipF_gemm", None, h.NO_TRANS, h.NO_TRANS, 64, 64, 64, f32(1), dI.ptr, dI.stride, dB2.ptr, dB2.stride,
           f32(0), dC2.ptr, dC2.stride)
    assert (dC2.numpy() == Bm).all()
    # K = 0: C = beta*C
    h.call("hipF_gemm", None, h.NO_TRANS, h.NO_TRANS, 64, 64, 0, f32(1), None, 64, None, 64, f32(2), dC2.ptr,
           dC2.stride)
    assert (dC2.numpy() == 2 * Bm).all()
    # run-to-run reproducibility (fixed summation order, no atomics)
    A2, B2 = rnd((256, 2048), 34), rnd((2048, 2048), 35)
    dA2, dB3 = dev(h, A2), dev(h, B2)
    outs = []
    for _ in range(3):
        dC3 = dev(h, np.zeros((256, 2048), f32))
        h.call("hipF_gemm", None, h.NO_TRANS, h.TRANS, 256, 2048, 2048, f32(1), dA2.ptr, dA2.stride, dB3.ptr,
               dB3.stride, f32(0), dC3.ptr, dC3.stride)
        outs.append(dC3.numpy())
    assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all()


@pytest.mark.parametrize("bm,bn,s", [(64, 64, 1), (64, 64, 4), (128, 64, 2), (64, 128, 1), (128, 128, 2),
                                     (32, 64, 1), (32, 64, 2), (32, 32, 1), (32, 32, 4)])
def test_gemm_every_tile(hip, bm, bn, s):
    """Every tile form of the kernel (the last four pairs split K over the waves of a workgroup), with and
    without a split over workgroups, on ragged shapes in the three operand layouts of the path."""
    h = hip
    try:
        h.call("hipK_gemm_force_plan", bm, bn, s)
        for (M, N, K, ta, tb) in [(256, 2048, 1320, 0, 1), (256, 520, 1000, 0, 0), (130, 70, 200, 1, 0),
                                  (33, 65, 64, 0, 1), (64, 96, 40, 0, 0)]:
            A = rnd((K, M) if ta else (M, K), 36)
            B = rnd((N, K) if tb else (K, N), 37)
            C0 = rnd((M, N), 38)
            dA, dB, dC = dev(h, A), dev(h, B), dev(h, C0)
            h.call("hipF_gemm", None, h.TRANS if ta else h.NO_TRANS, h.TRANS if tb else h.NO_TRANS, M, N, K, f32(0.5),
                   dA.ptr, dA.stride, dB.ptr, dB.stride, f32(1.0), dC.ptr, dC.stride)
            ref = O.gemm(0.5, A, ta, B, tb, 1.0, C0)
            np.testing.assert_allclose(dC.numpy(), ref, rtol=1e-5, atol=6e-6 * np.sqrt(K))
    finally:
        h.call("hipK_gemm_force_plan", 0, 0, 0)


AFFINE_SHAPES = [(256, 1024, 440), (256, 1483, 1024), (256, 2048, 1320), (256, 2048, 2048), (256, 9000, 2048),
                 (16, 17, 24), (100, 33, 50)]


@pytest.mark.parametrize("N,O_,I", AFFINE_SHAPES)
def test_affine_forward_backward_update(hip, N, O_, I):
    h = hip
    x = rnd((N, I), 40)
    W = rnd((O_, I), 41, 0.1)
    b = rnd((O_,), 42, 0.5)
    layer = O.BiasedLinearity(W, b)
    dx, dW, db = dev(h, x), dev(h, W), dev(h, b, False)
    out = dev(h, np.full((N, O_), np.nan, f32))
    pre = layer.propagate(x)
    for act, ref in [(h.ACT_NONE, pre), (h.ACT_SIGMOID, O.sigmoid(pre)), (h.ACT_SOFTMAX, O.softmax(pre))]:
        h.call("hipF_affine_forward", None, act, h.TRANS, N, O_, I, dx.ptr, dx.stride, dW.ptr, dW.stride, db.ptr,
               out.ptr, out.stride)
        got = out.numpy()
        if act == h.ACT_SOFTMAX:
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-12)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=3e-6 * np.sqrt(I))
    # backward with and without the sigmoid derivative of the layer below
    e = rnd((N, O_), 43, 0.01)
    y = O.sigmoid(rnd((N, I), 44))
    de, dy = dev(h, e), dev(h, y)
    eout = dev(h, np.full((N, I), np.nan, f32))
    ref_lin = layer.backpropagate(e)
    h.call("hipF_affine_backward", None, N, O_, I, de.ptr, de.stride, dW.ptr, dW.stride, None, 0, eout.ptr,
           eout.stride)
    np.testing.assert_allclose(eout.numpy(), ref_lin, rtol=1e-5, atol=3e-8 * np.sqrt(O_))
    h.call("hipF_affine_backward", None, N, O_, I, de.ptr, de.stride, dW.ptr, dW.stride, dy.ptr, dy.stride, eout.ptr,
           eout.stride)
    np.testing.assert_allclose(eout.numpy(), O.diff_sigmoid(ref_lin, y), rtol=1e-5, atol=1e-8 * np.sqrt(O_))
    # two SGD steps with momentum, L2 and averaged gradient; then plain (momentum 0, no corr buffers).
    # The L2 term is parity unpinned: the reference's CPU path aborts on --l2-penalty (self-AddMat of a non-square
    # matrix), so the oracle restates the formula of its GPU kernel and no reference output exists to compare with.
    for (mom, l2, avg) in [(0.9, 1e-4, True), (0.0, 0.0, False)]:
        layer = O.BiasedLinearity(W, b)
        layer.learn_rate, layer.momentum, layer.l2_penalty, layer.average_grad = f32(0.008), f32(mom), f32(l2), avg
        dW.set(W)
        db.set(b)
        corr, bcorr = dev(h, np.zeros_like(W)), dev(h, np.zeros_like(b), False)
        for step in range(2):
            layer.update(x, e)
            g = f32(1.0 / N) if avg else f32(1.0)
            l2s = f32(f32(0.008) * f32(l2) * N) if l2 else f32(0)
            use_corr = mom != 0
            h.call("hipF_affine_update", None, N, O_, I, de.ptr, de.stride, dx.ptr, dx.stride, g, f32(mom),
                   f32(0.008), l2s, dW.ptr, dW.stride, corr.ptr if use_corr else None, corr.stride,
                   db.ptr, bcorr.ptr if use_corr else None)
            # north_star: gradients / updated weights within 1e-3 relative
            np.testing.assert_allclose(dW.numpy(), layer.W, rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(db.numpy()[0], layer.b, rtol=1e-5, atol=1e-7)
            if use_corr:
                np.testing.assert_allclose(corr.numpy(), layer.W_corr, rtol=1e-4, atol=1e-7)
                np.testing.assert_allclose(bcorr.numpy()[0], layer.b_corr, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("N,O_,I", [(256, 2048, 1320), (256, 1483, 1024), (37, 50, 24), (300, 130, 72), (1, 65, 8)])
@pytest.mark.parametrize("mom", [0.0, 0.9])
def test_affine_update_fused_equals_halves(hip, N, O_, I, mom):
    """hipF_affine_update_fused (weight pass + bias pass in one launch, the bias sums taken by extra workgroups at
    the start of the gradient GEMM's grid) against hipF_affine_update_weights + hipF_affine_update_bias: same bits,
    into a second weight buffer and in place, outputs that are no multiple of the tile (ragged last bias tile)."""
    h = hip
    x, e = rnd((N, I), 70), rnd((N, O_), 71, 0.01)
    W, b = rnd((O_, I), 72, 0.1), rnd((O_,), 73, 0.5)
    c0, bc0 = rnd((O_, I), 74, 0.01), rnd((O_,), 75, 0.01)
    dx, de = dev(h, x), dev(h, e)
    lr, l2, g = f32(0.008), f32(1e-4), f32(1.0 / N)
    # reference: the two halves, weights into a second buffer
    Wa, Wa2, ba = dev(h, W), dev(h, np.zeros_like(W)), dev(h, b, False)
    ca, bca = dev(h, c0), dev(h, bc0, False)
    h.call("hipF_affine_update_weights", None, N, O_, I, de.ptr, de.stride, dx.ptr, dx.stride, g, f32(mom), lr, l2,
           Wa.ptr, Wa.stride, Wa2.ptr, Wa2.stride, ca.ptr if mom else None, ca.stride)
    h.call("hipF_affine_update_bias", None, N, O_, de.ptr, de.stride, g, f32(mom), lr, ba.ptr, bca.ptr if mom else None)
    for in_place in (False, True):
        Wb, Wb2, bb = dev(h, W), dev(h, np.zeros_like(W)), dev(h, b, False)
        cb, bcb = dev(h, c0), dev(h, bc0, False)
        out = Wb if in_place else Wb2
        h.call("hipF_affine_update_fused", None, N, O_, I, de.ptr, de.stride, dx.ptr, dx.stride, g, f32(mom), lr, l2,
               Wb.ptr, Wb.stride, out.ptr, out.stride, cb.ptr if mom else None, cb.stride, bb.ptr,
               bcb.ptr if mom else None)
        assert np.array_equal(out.numpy(), Wa2.numpy())
        assert np.array_equal(bb.numpy(), ba.numpy())
        if mom:
            assert np.array_equal(cb.numpy(), ca.numpy()) and np.array_equal(bcb.numpy(), bca.numpy())
        if not in_place:
            assert np.array_equal(Wb.numpy(), W)   # the old weights are only read


@pytest.mark.parametrize("mom", [0.0, 0.9])
def test_affine_update_group_equals_single_launches(hip, mom):
    """hipF_affine_update_group (the updates of several layers in one launch) against one hipF_affine_update_fused
    call per item: same bits in weights, bias and momentum buffers.  Eleven items, so the group is cut into two
    launches; shapes with ragged edge tiles, a layer smaller than one tile, one whose rows are not 16-byte
    aligned (launched on its own inside the call), and a second round on the updated weights.  count 0 is a no-op."""
    import ctypes
    h = hip
    shapes = [(256, 2048, 1320), (256, 1483, 1024), (37, 50, 24), (300, 130, 72), (1, 65, 8), (256, 512, 2048),
              (64, 200, 64), (256, 64, 64), (16, 17, 33), (256, 1024, 1024), (128, 700, 260)]
    lr, l2 = 0.008, 1e-4
    sets = []
    for k, (N, O_, I) in enumerate(shapes):
        x, e = rnd((N, I), 170 + k), rnd((N, O_), 190 + k, 0.01)
        W, b = rnd((O_, I), 210 + k, 0.1), rnd((O_,), 230 + k, 0.5)
        c0, bc0 = rnd((O_, I), 250 + k, 0.01), rnd((O_,), 270 + k, 0.01)
        pitched = (N, O_, I) != (16, 17, 33)  # the odd one: dense rows of 33 floats
        one = dict(N=N, O=O_, I=I, g=1.0 / N, x=h.DevMatrix.from_numpy(x, pitched), e=h.DevMatrix.from_numpy(e, pitched))
        for tag in ("a", "b"):
            one["W" + tag], one["b" + tag] = h.DevMatrix.from_numpy(W, pitched), dev(h, b, False)
            one["c" + tag], one["bc" + tag] = h.DevMatrix.from_numpy(c0, pitched), dev(h, bc0, False)
        sets.append(one)
    h.call("hipF_affine_update_group", None, 0, None)
    items = (h.AffineUpdate * len(sets))()
    for k, s in enumerate(sets):
        items[k] = h.AffineUpdate(s["N"], s["O"], s["I"], s["e"]._p.value, s["e"].stride, s["x"]._p.value, s["x"].stride, s["g"], mom,
                                  lr, l2, s["Wb"]._p.value, s["Wb"].stride, s["Wb"]._p.value, s["Wb"].stride,
                                  s["cb"]._p.value if mom else None, s["cb"].stride, s["bb"]._p.value,
                                  s["bcb"]._p.value if mom else None)
    for rnd_no in range(2):
        for s in sets:
            h.call("hipF_affine_update_fused", None, s["N"], s["O"], s["I"], s["e"].ptr, s["e"].stride, s["x"].ptr, s["x"].stride,
                   f32(s["g"]), f32(mom), f32(lr), f32(l2), s["Wa"].ptr, s["Wa"].stride, s["Wa"].ptr, s["Wa"].stride,
                   s["ca"].ptr if mom else None, s["ca"].stride, s["ba"].ptr, s["bca"].ptr if mom else None)
        h.call("hipF_affine_update_group", None, len(sets), ctypes.cast(items, ctypes.c_void_p))
        for k, s in enumerate(sets):
            assert np.array_equal(s["Wa"].numpy(), s["Wb"].numpy()), (rnd_no, shapes[k])
            assert np.array_equal(s["ba"].numpy(), s["bb"].numpy()), (rnd_no, shapes[k])
            if mom:
                assert np.array_equal(s["ca"].numpy(), s["cb"].numpy()) and np.array_equal(s["bca"].numpy(), s["bcb"].numpy())
    # a bad argument in the LAST item: the call is refused and no item before it has run
    before = [s["Wb"].numpy() for s in sets]
    items[len(sets) - 1].lde = 1
    with pytest.raises(h.HipError):
        h.call("hipF_affine_update_group", None, len(sets), ctypes.cast(items, ctypes.c_void_p))
    for s, w in zip(sets, before):
        assert np.array_equal(s["Wb"].numpy(), w)


@pytest.mark.parametrize("N,O_,I", [(256, 2048, 440), (256, 2048, 2048), (37, 50, 24), (64, 130, 72), (300, 700, 260)])
def test_affine_forward_sample_equals_two_calls(hip, N, O_, I):
    """hipF_affine_forward_sample (sigmoid(in * W^T + b) and the Bernoulli sample of it in ONE launch: the positive
    phase of an RBM CD-1 step) against hipF_affine_forward followed by hipF_binarize_probs_rand on copies of the same
    generator state: the same probabilities, the same states and the same generator state afterwards, bit for bit,
    over two consecutive calls; the padding of the state arrays is left alone."""
    h = hip
    rs = np.random.RandomState(N + O_)
    x = h.DevMatrix.from_numpy(rs.randn(N, I).astype(f32))
    W = h.DevMatrix.from_numpy((0.3 * rs.randn(O_, I)).astype(f32))
    b = dev(h, (0.2 * rs.randn(O_)).astype(f32), False)
    pa, sa, pb, sb = (h.DevMatrix(N, O_) for _ in range(4))
    seeds = [rs.randint(128, 2 ** 31 - 1, (N, O_)).astype(np.uint32) for _ in range(4)]
    za = [h.DevMatrix.from_numpy(z) for z in seeds]   # pitched like the probabilities: the same index for both
    zb = [h.DevMatrix.from_numpy(z) for z in seeds]
    assert za[0].stride == pa.stride
    for rep in range(2):
        h.call("hipF_affine_forward", None, h.ACT_SIGMOID, h.TRANS, N, O_, I, x.ptr, x.stride, W.ptr, W.stride, b.ptr, pa.ptr, pa.stride)
        h.call("hipF_binarize_probs_rand", None, sa.ptr, pa.ptr, za[0].ptr, za[1].ptr, za[2].ptr, za[3].ptr, pa.dim)
        h.call("hipF_affine_forward_sample", None, N, O_, I, x.ptr, x.stride, W.ptr, W.stride, b.ptr, pb.ptr, pb.stride, sb.ptr, sb.stride,
               zb[0].ptr, zb[1].ptr, zb[2].ptr, zb[3].ptr)
        assert np.array_equal(pa.numpy(), pb.numpy()), rep
        assert np.array_equal(sa.numpy(), sb.numpy()), rep
        for k in range(4):
            assert np.array_equal(za[k].numpy(), zb[k].numpy()), (rep, k)
    states = sb.numpy()
    assert set(np.unique(states).tolist()) <= {0.0, 1.0}
    assert abs(states.mean() - pb.numpy().mean()) < 0.02
    with pytest.raises(h.HipError):
        h.call("hipF_affine_forward_sample", None, N, O_, I, x.ptr, x.stride, W.ptr, W.stride, b.ptr, pb.ptr, pb.stride, sb.ptr, sb.stride + 4,
               zb[0].ptr, zb[1].ptr, zb[2].ptr, zb[3].ptr)


@pytest.mark.parametrize("width,nl", [(2048, 6), (1024, 3)])
def test_affine_forward_chain_equals_separate_launches(hip, width, nl):
    """hipF_affine_forward_chain (dependent hidden layers in one launch, activations handed from workgroup to workgroup
    inside it) against one hipF_affine_forward launch per layer: the same bits in every layer's output, over several
    launches (the hand-off counters only grow) with fresh inputs; a chain the one-launch form does not take (a first
    layer whose K ends inside a tile) runs as separate launches inside the call and gives the same bits too."""
    import ctypes
    h = hip
    N = 256
    rs = np.random.RandomState(width)
    dims = [width] * (nl + 1)
    for dims in ([width] * (nl + 1), [1320] + [width] * 2):
        n = len(dims) - 1
        x = h.DevMatrix.from_numpy(rs.randn(N, dims[0]).astype(f32))
        Ws = [h.DevMatrix.from_numpy((0.05 * rs.randn(dims[i + 1], dims[i])).astype(f32)) for i in range(n)]
        bs = [dev(h, (0.1 * rs.randn(dims[i + 1])).astype(f32), False) for i in range(n)]
        act_a = [x] + [h.DevMatrix(N, dims[i + 1]) for i in range(n)]
        act_b = [x] + [h.DevMatrix(N, dims[i + 1]) for i in range(n)]
        items = (h.AffineForward * n)()
        for i in range(n):
            items[i] = h.AffineForward(dims[i + 1], dims[i], act_b[i]._p.value, act_b[i].stride, Ws[i]._p.value, Ws[i].stride,
                                       bs[i]._p.value, act_b[i + 1]._p.value, act_b[i + 1].stride)
        for rep in range(4):
            x.set(rs.randn(N, dims[0]).astype(f32))
            for i in range(n):
                h.call("hipF_affine_forward", None, h.ACT_SIGMOID, h.TRANS, N, dims[i + 1], dims[i], act_a[i].ptr, act_a[i].stride,
                       Ws[i].ptr, Ws[i].stride, bs[i].ptr, act_a[i + 1].ptr, act_a[i + 1].stride)
            h.call("hipF_affine_forward_chain", None, h.ACT_SIGMOID, N, n, ctypes.cast(items, ctypes.c_void_p))
            for i in range(1, n + 1):
                assert np.array_equal(act_a[i].numpy(), act_b[i].numpy()), (dims, rep, i)
    # layers that do not follow one another are refused
    items[1].in_ = x._p.value
    with pytest.raises(h.HipError):
        h.call("hipF_affine_forward_chain", None, h.ACT_SIGMOID, N, n, ctypes.cast(items, ctypes.c_void_p))


def test_affine_gradient_group_and_sgd_apply_segments(hip):
    """The two halves of the sharded data-parallel step.
    hipF_affine_gradient_group: gW = e^T * in with the weight matrix's row pitch and gb = colsum(e) for several layers in
    one launch -- against float64 numpy, bit for bit against the same items launched one at a time (another tile size
    above the big-tile threshold, K summed in one piece either way), the row padding left alone; an item with dense
    rows of 33 floats takes the stand-alone kernels inside the call.
    hipF_sgd_apply_segments: the step on a range of an arena that holds those layers (weights, bias, padding) -- bit
    for bit hipF_affine_apply layer by layer, elements outside the range or outside every segment untouched."""
    import ctypes
    h = hip
    shapes = [(256, 2048, 1320), (256, 2048, 2048), (256, 1483, 1024), (37, 50, 24), (300, 130, 72), (16, 17, 33), (256, 9000, 2048)]
    sets, items = [], (h.AffineGradient * len(shapes))()
    for k, (N, O_, I) in enumerate(shapes):
        x, e = rnd((N, I), 570 + k), rnd((N, O_), 590 + k, 0.01)
        pitched = (N, O_, I) != (16, 17, 33)
        dx, de = h.DevMatrix.from_numpy(x, pitched), h.DevMatrix.from_numpy(e, pitched)
        ga, gb_ = h.DevMatrix.from_numpy(np.full((O_, I), 7.0, f32), pitched), h.DevMatrix.from_numpy(np.full((O_, I), 7.0, f32), pitched)
        if pitched and ga.stride > I:   # mark the padding: it must come back untouched
            h.call("hipF_set_const", None, ga.ptr, f32(-3.0), h.MatrixDim(O_, ga.stride, ga.stride))
            h.call("hipF_set_const", None, gb_.ptr, f32(-3.0), h.MatrixDim(O_, gb_.stride, gb_.stride))
        ba, bb = dev(h, np.zeros(O_, f32), False), dev(h, np.zeros(O_, f32), False)
        sets.append(dict(N=N, O=O_, I=I, x=x, e=e, dx=dx, de=de, ga=ga, gb=gb_, ba=ba, bb=bb))
        items[k] = h.AffineGradient(N, O_, I, de._p.value, de.stride, dx._p.value, dx.stride, ga._p.value, ga.stride, ba._p.value)
    h.call("hipF_affine_gradient_group", None, 0, None)
    h.call("hipF_affine_gradient_group", None, len(sets), ctypes.cast(items, ctypes.c_void_p))
    for k, s_ in enumerate(sets):
        one = (h.AffineGradient * 1)(h.AffineGradient(s_["N"], s_["O"], s_["I"], s_["de"]._p.value, s_["de"].stride, s_["dx"]._p.value,
                                                      s_["dx"].stride, s_["gb"]._p.value, s_["gb"].stride, s_["bb"]._p.value))
        h.call("hipF_affine_gradient_group", None, 1, ctypes.cast(one, ctypes.c_void_p))
        G, G1 = s_["ga"].numpy(), s_["gb"].numpy()
        ref = s_["e"].astype(np.float64).T @ s_["x"].astype(np.float64)
        scale = np.abs(s_["e"]).max() * np.abs(s_["x"]).max()
        np.testing.assert_allclose(G, ref, rtol=0, atol=6e-6 * np.sqrt(s_["N"]) * scale * 4)
        assert np.array_equal(G, G1), shapes[k]
        np.testing.assert_allclose(s_["ba"].numpy()[0], s_["e"].astype(np.float64).sum(0), rtol=1e-5, atol=1e-6)
        assert np.array_equal(s_["ba"].numpy(), s_["bb"].numpy()), shapes[k]
        if s_["ga"].stride > s_["I"] and shapes[k] != (16, 17, 33):
            raw = np.zeros((s_["O"], s_["ga"].stride), f32)
            h.call("hipK_memcpy", None, raw.ctypes.data, s_["ga"].ptr, raw.nbytes, 1)
            assert np.all(raw[:, s_["I"]:] == -3.0), shapes[k]
    items[len(sets) - 1].ldg = 1
    with pytest.raises(h.HipError):
        h.call("hipF_affine_gradient_group", None, len(sets), ctypes.cast(items, ctypes.c_void_p))

    # ---- the step on an arena of three layers ---------------------------------------------------------------------
    layers = [(96, 40), (64, 96), (33, 64)]
    hyper = [(1.0 / 64, 0.7, 0.05, 1e-3), (1.0 / 64, 0.7, 0.025, 0.0), (1.0, 0.0, 0.01, 2e-3)]   # gscale, momentum, lr, l2
    spans, off = [], 0
    for O_, I in layers:
        stride = (I * 4 + 255) // 256 * 64
        spans.append((off, stride))
        off += O_ * stride + (O_ + 63) // 64 * 64
    total = (off + 511) // 512 * 512
    rs = np.random.RandomState(99)
    Wa, Ga, Ca = np.zeros(total, f32), np.zeros(total, f32), np.zeros(total, f32)
    for (O_, I), (o, stride) in zip(layers, spans):
        for arr, sc in ((Wa, 0.1), (Ga, 1.0), (Ca, 0.01)):
            m = arr[o:o + O_ * stride].reshape(O_, stride)
            m[:, :I] = sc * rs.standard_normal((O_, I))
            arr[o + O_ * stride:o + O_ * stride + O_] = sc * rs.standard_normal(O_)
    segs = (h.SgdSegment * 6)()
    for k, ((O_, I), (o, stride), (g, mom, lr, l2)) in enumerate(zip(layers, spans, hyper)):
        segs[2 * k] = h.SgdSegment(o, o + O_ * stride, g, mom, lr, l2)
        segs[2 * k + 1] = h.SgdSegment(o + O_ * stride, o + O_ * stride + (O_ + 63) // 64 * 64, g, mom, lr, 0.0)
    for first, count in ((0, total), (total // 4, total // 2), (total - 512, 512), (128, 0)):
        dW, dG, dC = dev(h, Wa, False), dev(h, Ga, False), dev(h, Ca, False)
        h.call("hipF_sgd_apply_segments", None, 6, ctypes.cast(segs, ctypes.c_void_p), first, count, dG.ptr, dW.ptr, dC.ptr)
        gotW, gotC = dW.numpy()[0], dC.numpy()[0]
        expW, expC = Wa.copy(), Ca.copy()
        for (O_, I), (o, stride), (g, mom, lr, l2) in zip(layers, spans, hyper):
            # the same step by hipF_affine_apply on copies of this layer (whole rows, padding included, as one matrix)
            lw, lg, lc = (h.DevMatrix.from_numpy(a[o:o + O_ * stride].reshape(O_, stride).copy(), False) for a in (Wa, Ga, Ca))
            lb, lgb, lcb = (dev(h, a[o + O_ * stride:o + O_ * stride + O_].copy(), False) for a in (Wa, Ga, Ca))
            h.call("hipF_affine_apply", None, O_, stride, lg.ptr, stride, lgb.ptr, f32(g), f32(mom), f32(lr), f32(l2), lw.ptr, stride,
                   lc.ptr, stride, lb.ptr, lcb.ptr)
            expW[o:o + O_ * stride], expC[o:o + O_ * stride] = lw.numpy().ravel(), lc.numpy().ravel()
            expW[o + O_ * stride:o + O_ * stride + O_], expC[o + O_ * stride:o + O_ * stride + O_] = lb.numpy()[0], lcb.numpy()[0]
        inside = np.zeros(total, bool)
        inside[first:first + count] = True
        assert np.array_equal(gotW[inside], expW[inside]) and np.array_equal(gotW[~inside], Wa[~inside]), (first, count)
        assert np.array_equal(gotC[inside], expC[inside]), (first, count)
        assert np.array_equal(gotC[~inside], Ca[~inside])
    with pytest.raises(h.HipError):
        h.call("hipF_sgd_apply_segments", None, 6, ctypes.cast(segs, ctypes.c_void_p), 2, 8, dG.ptr, dW.ptr, dC.ptr)


@pytest.mark.parametrize("N,O_,I", [(256, 1483, 1024), (256, 9000, 2048), (37, 50, 24), (64, 5000, 64), (64, 12000, 128)])
def test_affine_forward_xent_equals_two_calls(hip, N, O_, I):
    """The fused output layer (GEMM + softmax + Xent::EvalVec) against hipF_affine_forward(softmax) followed by
    hipF_xent_eval: identical diff, arg-max ids, counters and loss; the last shape is too wide for the
    register-resident kernel and takes the two-kernel form inside the library."""
    h = hip
    rs = np.random.RandomState(60)
    x, W, b = rnd((N, I), 61), rnd((O_, I), 62, 0.1), rnd((O_,), 63, 0.5)
    tgt = rs.randint(0, O_, N).astype(np.int32)
    dx, dW, db, dt = dev(h, x), dev(h, W), dev(h, b, False), dev(h, tgt, False)
    post_a, diff_a = dev(h, np.zeros((N, O_), f32)), dev(h, np.zeros((N, O_), f32))
    cnt_a, loss_a = h.DevMatrix(1, 2, np.int64, pitched=False), h.DevMatrix(1, 1, np.float64, pitched=False)
    mid_a, lp_a = h.DevMatrix(1, N, np.int32, pitched=False), h.DevMatrix(1, N, f32, pitched=False)
    h.call("hipF_affine_forward", None, h.ACT_SOFTMAX, h.TRANS, N, O_, I, dx.ptr, dx.stride, dW.ptr, dW.stride, db.ptr,
           post_a.ptr, post_a.stride)
    h.call("hipF_xent_eval", None, post_a.ptr, dt.ptr, diff_a.ptr, lp_a.ptr, mid_a.ptr, cnt_a.ptr, loss_a.ptr, post_a.dim)
    for write_post, with_rows in [(1, True), (0, False)]:
        post_b, diff_b = dev(h, np.full((N, O_), np.nan, f32)), dev(h, np.full((N, O_), np.nan, f32))
        cnt_b, loss_b = h.DevMatrix(1, 2, np.int64, pitched=False), h.DevMatrix(1, 1, np.float64, pitched=False)
        mid_b, lp_b = h.DevMatrix(1, N, np.int32, pitched=False), h.DevMatrix(1, N, f32, pitched=False)
        for _ in range(2):   # two calls accumulate
            h.call("hipF_affine_forward_xent", None, h.TRANS, N, O_, I, dx.ptr, dx.stride, dW.ptr, dW.stride, db.ptr,
                   post_b.ptr, post_b.stride, write_post, dt.ptr, diff_b.ptr, diff_b.stride,
                   lp_b.ptr if with_rows else None, mid_b.ptr if with_rows else None, cnt_b.ptr, loss_b.ptr)
        assert (diff_b.numpy() == diff_a.numpy()).all()
        assert cnt_b.numpy()[0].tolist() == (2 * cnt_a.numpy()[0]).tolist()
        assert abs(loss_b.numpy()[0, 0] - 2 * loss_a.numpy()[0, 0]) <= 1e-12 * abs(loss_a.numpy()[0, 0])
        if with_rows:
            assert (mid_b.numpy() == mid_a.numpy()).all() and (lp_b.numpy() == lp_a.numpy()).all()
        if write_post:
            assert (post_b.numpy() == post_a.numpy()).all()
    # and against the oracle: exact frame-accuracy count, loss within 1e-5
    xo = O.Xent()
    xo.eval_vec(O.softmax(O.BiasedLinearity(W, b).propagate(x)), tgt)
    assert int(cnt_a.numpy()[0, 1]) == N and abs(loss_a.numpy()[0, 0] - xo.loss) <= 1e-5 * abs(xo.loss)


@pytest.mark.parametrize("V,H,vt,ht,N", [(440, 2048, "gauss", "bern", 256), (2048, 2048, "bern", "bern", 256),
                                         (24, 32, "bern", "gauss", 256), (24, 32, "bern", "bern", 48), (22, 30, "gauss", "bern", 64)])
def test_rbm_cd1(hip, V, H, vt, ht, N):
    """Two CD-1 steps against the oracle (nnet-rbm.h:136-223).  Bunches of 256 take the one-launch update (both
    weight statistics, the bias statistics and the reconstruction error in it); 48 rows (not whole K tiles) and
    widths that are no multiples of 4 take the separate launches: same results either way."""
    h = hip
    rs = np.random.RandomState(50)
    W = rnd((H, V), 51, 0.1)
    vb, hb = rnd((V,), 52, 0.1), rnd((H,), 53, 0.1)
    pos_vis = rnd((N, V), 54) if vt == "gauss" else rs.rand(N, V).astype(f32)
    u = rs.rand(N, H).astype(f32)
    g = rnd((N, H), 55)
    rbm = O.Rbm(W, vb, hb, vt, ht)
    rbm.learn_rate, rbm.momentum, rbm.l2_penalty = f32(0.01), f32(0.5), f32(2e-4)
    dW, dvb, dhb = dev(h, W), dev(h, vb, False), dev(h, hb, False)
    ddW, ddvb, ddhb = dev(h, np.zeros_like(W)), dev(h, np.zeros_like(vb), False), dev(h, np.zeros_like(hb), False)
    dpv = dev(h, pos_vis)
    dph, dnh, dnv = dev(h, np.zeros((N, H), f32)), dev(h, np.zeros((N, H), f32)), dev(h, np.zeros((N, V), f32))
    hact = h.ACT_SIGMOID if ht == "bern" else h.ACT_NONE
    vact = h.ACT_SIGMOID if vt == "bern" else h.ACT_NONE
    for step in range(2):
        ref = O.rbm_cd1_step(rbm, pos_vis, uniform=u, gauss=g)
        h.call("hipF_affine_forward", None, hact, h.TRANS, N, H, V, dpv.ptr, dpv.stride, dW.ptr, dW.stride, dhb.ptr,
               dph.ptr, dph.stride)
        np.testing.assert_allclose(dph.numpy(), ref["pos_hid"], rtol=1e-5, atol=3e-6)
        # the sample is injected from the oracle so that a last-ulp flip cannot change the comparison
        dsample = dev(h, ref["neg_hid_sample"])
        h.call("hipF_affine_forward", None, vact, h.NO_TRANS, N, V, H, dsample.ptr, dsample.stride, dW.ptr, dW.stride,
               dvb.ptr, dnv.ptr, dnv.stride)
        np.testing.assert_allclose(dnv.numpy(), ref["neg_vis"], rtol=2e-5, atol=2e-5)
        h.call("hipF_affine_forward", None, hact, h.TRANS, N, H, V, dnv.ptr, dnv.stride, dW.ptr, dW.stride, dhb.ptr,
               dnh.ptr, dnh.stride)
        np.testing.assert_allclose(dnh.numpy(), ref["neg_hid"], rtol=2e-5, atol=2e-5)
        # (copies for the two-call form below: weights and bias statistics as launches of their own)
        W_before, dW_before = dW.numpy(), ddW.numpy()
        vb_before, dvb_before, hb_before, dhb_before = dvb.numpy(), ddvb.numpy(), dhb.numpy(), ddhb.numpy()
        cW, cdW = dev(h, dW.numpy()), dev(h, ddW.numpy())
        cvb, cdvb = dev(h, dvb.numpy()[0], False), dev(h, ddvb.numpy()[0], False)
        chb, cdhb = dev(h, dhb.numpy()[0], False), dev(h, ddhb.numpy()[0], False)
        h.call("hipF_rbm_update", None, N, H, V, dpv.ptr, dpv.stride, dph.ptr, dph.stride, dnv.ptr, dnv.stride,
               dnh.ptr, dnh.stride, f32(0.01), f32(0.5), f32(2e-4), dW.ptr, dW.stride, ddW.ptr, ddW.stride, dvb.ptr,
               ddvb.ptr, dhb.ptr, ddhb.ptr)
        # hipF_rbm_update carries the bias statistics in the weight launch where it can: same bits as the two calls
        h.call("hipF_rbm_update_weights", None, N, H, V, dpv.ptr, dpv.stride, dph.ptr, dph.stride, dnv.ptr, dnv.stride,
               dnh.ptr, dnh.stride, f32(0.01), f32(0.5), f32(2e-4), cW.ptr, cW.stride, cdW.ptr, cdW.stride)
        h.call("hipF_rbm_update_bias", None, N, H, V, dpv.ptr, dpv.stride, dph.ptr, dph.stride, dnv.ptr, dnv.stride,
               dnh.ptr, dnh.stride, f32(0.01), f32(0.5), cvb.ptr, cdvb.ptr, chb.ptr, cdhb.ptr)
        for a, b in ((dW, cW), (ddW, cdW), (dvb, cvb), (ddvb, cdvb), (dhb, chb), (ddhb, cdhb)):
            assert np.array_equal(a.numpy(), b.numpy())
        # hipF_rbm_update_mse: the same step plus the reconstruction error, which rides in the weight launch too:
        # every result as above, the error as hipF_mse_eval adds it up (same bits) and as the oracle does
        eW, edW = dev(h, W_before), dev(h, dW_before)
        evb, edvb = dev(h, vb_before[0], False), dev(h, dvb_before[0], False)
        ehb, edhb = dev(h, hb_before[0], False), dev(h, dhb_before[0], False)
        loss_a, loss_b = h.DevMatrix(1, 1, np.float64, pitched=False), h.DevMatrix(1, 1, np.float64, pitched=False)
        for _ in range(2):   # (twice: the arrival counter must be back at zero for the second call)
            h.call("hipF_mse_eval", None, dnv.ptr, dpv.ptr, None, None, loss_a.ptr, dnv.dim)
        h.call("hipF_mse_eval", None, dnv.ptr, dpv.ptr, None, None, loss_b.ptr, dnv.dim)
        h.call("hipF_rbm_update_mse", None, N, H, V, dpv.ptr, dpv.stride, dph.ptr, dph.stride, dnv.ptr, dnv.stride,
               dnh.ptr, dnh.stride, f32(0.01), f32(0.5), f32(2e-4), eW.ptr, eW.stride, edW.ptr, edW.stride, evb.ptr,
               edvb.ptr, ehb.ptr, edhb.ptr, loss_b.ptr)
        for a, b in ((dW, eW), (ddW, edW), (dvb, evb), (ddvb, edvb), (dhb, ehb), (ddhb, edhb)):
            assert np.array_equal(a.numpy(), b.numpy())
        assert loss_b.numpy()[0, 0] == loss_a.numpy()[0, 0] and loss_a.numpy()[0, 0] > 0
        want = 0.5 * float(((dnv.numpy().astype(np.float64) - pos_vis) ** 2).sum())
        assert abs(loss_a.numpy()[0, 0] - 2 * want) <= 1e-5 * 2 * want
        np.testing.assert_allclose(dW.numpy(), rbm.W, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ddW.numpy(), rbm.W_corr, rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(dvb.numpy()[0], rbm.vis_bias, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dhb.numpy()[0], rbm.hid_bias, rtol=1e-4, atol=1e-6)


def test_argument_checks(hip):
    h = hip
    bad = h.MatrixDim(4, 8, 4)  # stride < cols
    assert h.lib.hipF_sigmoid(None, P(16), P(16), bad) == 100001
    assert b"requirement" in h.lib.hipK_error_string(100001)
    assert h.lib.hipF_gemm(None, 5, h.TRANS, 4, 4, 4, f32(1), P(16), 4, P(16), 4, f32(0), P(16), 4) == 100001
    assert h.lib.hipF_gemm(None, h.NO_TRANS, h.TRANS, 4, 4, 4, f32(1), P(16), 2, P(16), 4, f32(0), P(16), 4) == 100001
    # empty matrices are a no-op
    assert h.lib.hipF_sigmoid(None, None, None, h.MatrixDim(0, 0, 0)) == 0
    assert h.lib.hipF_softmax(None, None, None, h.MatrixDim(0, 5, 5)) == 0


@pytest.mark.parametrize("rows,cols", [(1, 1), (7, 50), (300, 9000), (4100, 33), (64, 2048)])
def test_count_nonfinite(hip, rows, cols):
    """hipF_count_nonfinite = the two host scans of nnet-forward (NaN, +inf; -inf is not counted there either,
    nnetbin/nnet-forward.cc:147-156), on pitched and on dense rows, and on a clean matrix."""
    h = hip
    rs = np.random.RandomState(rows * 7 + cols)
    x = rs.standard_normal((rows, cols)).astype(f32)
    cnt = h.DevMatrix(1, 2, np.int32, pitched=False)
    dx = dev(h, x)
    h.call("hipF_count_nonfinite", None, dx.ptr, dx.dim, cnt.ptr)
    assert cnt.numpy().ravel().tolist() == [0, 0]
    n = rows * cols
    flat = x.reshape(-1)
    k_nan, k_inf, k_ninf = min(n, 5), min(max(n - 5, 0), 3), min(max(n - 8, 0), 2)
    pos = rs.permutation(n)
    flat[pos[:k_nan]] = np.nan
    flat[pos[k_nan:k_nan + k_inf]] = np.inf
    flat[pos[k_nan + k_inf:k_nan + k_inf + k_ninf]] = -np.inf
    for pitched in (True, False):
        dx = dev(h, x, pitched)
        h.call("hipF_count_nonfinite", None, dx.ptr, dx.dim, cnt.ptr)
        assert cnt.numpy().ravel().tolist() == [k_nan, k_inf]
