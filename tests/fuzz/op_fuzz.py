#!/usr/bin/env python3
"""Random-shape check of the C-ABI kernels on WINDOWS of larger device buffers: an operand is a window (first row,
first column, rows, columns) of a buffer with its own row pitch, so pointers are often not 16-byte aligned, row
lengths are odd, and there is memory all around the result.  Each call is compared with numpy on the window AND
the rest of the buffer must come back untouched (a write outside the window is an error even where the values are
right).  Shapes include one row, one column, cache-sized matrices (the flat streaming kernels) and widths around
the kernels' internal limits.     python tests/fuzz/op_fuzz.py [--cases 300] [--seed 1]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kaldi_amd import hip as h  # noqa: E402
from oracle import nnet1_oracle as O  # noqa: E402

f32 = np.float32
FILL = f32(-777.25)


class Window:
    """rows x cols at (r0, c0) of a (R x S) buffer filled with a marker value"""

    def __init__(self, rs, rows, cols, data=None, aligned=False):
        self.rows, self.cols = rows, cols
        self.r0 = 0 if aligned else int(rs.randint(0, 3))
        self.c0 = 0 if aligned else int(rs.randint(0, 6))
        pad_c = 0 if aligned else int(rs.randint(0, 9))
        self.S = self.c0 + cols + pad_c
        if aligned:
            self.S = (self.S + 63) // 64 * 64
        self.R = self.r0 + rows + (0 if aligned else int(rs.randint(0, 3)))
        self.host = np.full((self.R, self.S), FILL, f32)
        if data is not None:
            self.host[self.r0:self.r0 + rows, self.c0:self.c0 + cols] = data
        self.dev = h.DevMatrix.from_numpy(self.host, pitched=False)

    @property
    def ptr(self):
        return ctypes.c_void_p(self.dev.ptr.value + 4 * (self.r0 * self.S + self.c0))

    @property
    def dim(self):
        return h.MatrixDim(self.rows, self.cols, self.S)

    def like(self, rs, data=None):
        """another window with the same geometry (the element-wise entry points take ONE MatrixDim for all operands)"""
        w = Window.__new__(Window)
        w.rows, w.cols, w.r0, w.c0, w.S, w.R = self.rows, self.cols, self.r0, self.c0, self.S, self.R
        w.host = np.full((w.R, w.S), FILL, f32)
        if data is not None:
            w.host[w.r0:w.r0 + w.rows, w.c0:w.c0 + w.cols] = data
        w.dev = h.DevMatrix.from_numpy(w.host, pitched=False)
        return w

    def check(self, want, rtol, atol, what):
        got = self.dev.numpy()
        win = got[self.r0:self.r0 + self.rows, self.c0:self.c0 + self.cols]
        outside = got.copy()
        outside[self.r0:self.r0 + self.rows, self.c0:self.c0 + self.cols] = FILL
        if not (outside == FILL).all():
            bad = np.argwhere(outside != FILL)[0]
            return "%s wrote outside its window, e.g. buffer element %s (window %dx%d at %d,%d of %dx%d)" % (
                what, tuple(bad), self.rows, self.cols, self.r0, self.c0, self.R, self.S)
        if want is not None and not np.allclose(win, want, rtol=rtol, atol=atol):
            k = np.unravel_index(np.argmax(np.abs(win - want)), win.shape)
            return "%s: window %dx%d at %d,%d pitch %d: max abs %.3g at %s (got %.6g want %.6g)" % (
                what, self.rows, self.cols, self.r0, self.c0, self.S, np.abs(win - want).max(), k, win[k], want[k])
        return None


def shape(rs):
    kind = rs.randint(0, 10)
    if kind == 0:
        return 1, int(rs.choice([1, 3, 64, 1000, 9000]))
    if kind == 1:
        return int(rs.choice([1, 7, 300, 5000])), 1
    if kind == 2:                      # cache-sized: the flat streaming kernels (>= 2^20 16-byte elements when aligned)
        return int(rs.choice([3301, 4100])), int(rs.choice([1320, 1316, 1321]))
    if kind == 3:                      # around the wave-per-row limits
        return int(rs.choice([1024, 1500, 2100])), int(rs.choice([255, 256, 257, 2047, 2048, 2049]))
    return int(rs.randint(1, 400)), int(rs.randint(1, 2600))


def one_case(rs):
    rows, cols = shape(rs)
    aligned = rs.rand() < 0.4
    op = rs.choice(["sigmoid", "diff_sigmoid", "add_mat", "scale", "mul_elements", "add_vec_to_rows", "softmax", "find_row_max_id",
                    "add_row_sum_mat", "add_col_sum_mat", "randomize", "gemm", "xent_eval", "affine_forward", "affine_backward",
                    "affine_update", "affine_forward_xent", "unary", "vec_scale", "part_ops", "l2bound", "l1", "diff_xent",
                    "expand", "copy", "mse_eval", "nonfinite", "relu_family", "softmax_part", "rbm_update", "rbm_update", "rng"])
    x = rs.standard_normal((rows, cols)).astype(f32)
    tag = "%s %dx%d%s" % (op, rows, cols, " aligned" if aligned else "")
    if op == "sigmoid":
        a = Window(rs, rows, cols, x, aligned)
        y = a.like(rs)
        h.call("hipF_sigmoid", None, y.ptr, a.ptr, a.dim)
        return tag, y.check(O.sigmoid(x), 1e-5, 1e-6, op) or a.check(x, 0, 0, op + " (input)")
    if op == "diff_sigmoid":
        yv = rs.random_sample((rows, cols)).astype(f32)
        e, y = Window(rs, rows, cols, x, aligned), None
        y = e.like(rs, yv)
        out = e.like(rs)
        h.call("hipF_diff_sigmoid", None, out.ptr, e.ptr, y.ptr, e.dim)
        return tag, out.check(O.diff_sigmoid(x, yv), 1e-5, 1e-6, op)
    if op == "add_mat":
        y0 = rs.standard_normal((rows, cols)).astype(f32)
        a = Window(rs, rows, cols, x, aligned)
        y = a.like(rs, y0)
        al, be = f32(rs.uniform(-2, 2)), f32(rs.choice([0.0, 1.0, 0.3]))
        h.call("hipF_add_mat", None, al, a.ptr, be, y.ptr, a.dim)
        return tag, y.check(al * x + be * y0, 1e-5, 1e-5, op)
    if op == "scale":
        a = Window(rs, rows, cols, x, aligned)
        h.call("hipF_scale", None, a.ptr, f32(1.5), a.dim)
        return tag, a.check(f32(1.5) * x, 1e-6, 1e-6, op)
    if op == "mul_elements":
        y0 = rs.standard_normal((rows, cols)).astype(f32)
        a = Window(rs, rows, cols, x, aligned)
        y = a.like(rs, y0)
        h.call("hipF_mul_elements", None, y.ptr, a.ptr, a.dim)
        return tag, y.check(y0 * x, 1e-6, 1e-6, op)
    if op == "add_vec_to_rows":
        v = rs.standard_normal(cols).astype(f32)
        a = Window(rs, rows, cols, x, aligned)
        dv = h.DevMatrix.from_numpy(v, pitched=False)
        h.call("hipF_add_vec_to_rows", None, f32(0.5), dv.ptr, f32(1.0), a.ptr, a.dim)
        return tag, a.check(f32(0.5) * v[None, :] + x, 1e-5, 1e-5, op)
    if op == "softmax":
        a = Window(rs, rows, cols, 3 * x, aligned)
        y = a.like(rs)
        h.call("hipF_softmax", None, y.ptr, a.ptr, a.dim)
        return tag, y.check(O.softmax(3 * x), 1e-4, 1e-7, op)
    if op == "find_row_max_id":
        if rs.rand() < 0.3 and cols > 2:     # ties: the first maximum wins
            x[:, cols // 2] = x.max(axis=1) + 1
            x[:, cols - 1] = x[:, cols // 2]
        a = Window(rs, rows, cols, x, aligned)
        ids = h.DevMatrix(1, rows, np.int32, pitched=False)
        vals = h.DevMatrix(1, rows, f32, pitched=False)
        h.call("hipF_find_row_max_id", None, a.ptr, vals.ptr, ids.ptr, a.dim)
        want = x.argmax(axis=1)
        if not (ids.numpy()[0] == want).all():
            return tag, "arg-max differs in %d rows" % int((ids.numpy()[0] != want).sum())
        if not (vals.numpy()[0] == x.max(axis=1)).all():
            return tag, "row maxima differ"
        return tag, a.check(x, 0, 0, op + " (input)")
    if op in ("add_row_sum_mat", "add_col_sum_mat"):
        a = Window(rs, rows, cols, x, aligned)
        n = cols if op == "add_row_sum_mat" else rows
        v0 = rs.standard_normal(n).astype(f32)
        dv = h.DevMatrix.from_numpy(v0, pitched=False)
        be = f32(rs.choice([0.0, 1.0]))
        h.call("hipF_" + op, None, f32(1.0), a.ptr, a.dim, be, dv.ptr)
        s = x.astype(np.float64).sum(axis=0 if op == "add_row_sum_mat" else 1)
        want = (s + be * v0).astype(f32)
        scale = np.abs(x).astype(np.float64).sum(axis=0 if op == "add_row_sum_mat" else 1).max()
        if not np.allclose(dv.numpy()[0], want, rtol=1e-4, atol=2e-6 * max(scale, 1.0)):
            return tag, "sums differ: max abs %.3g (scale %.3g)" % (np.abs(dv.numpy()[0] - want).max(), scale)
        return tag, a.check(x, 0, 0, op + " (input)")
    if op == "randomize":
        a = Window(rs, rows, cols, x, aligned)
        n_out = int(rs.randint(1, rows + 1))
        idx = rs.randint(0, rows, n_out).astype(np.int32)
        y = Window(rs, n_out, cols, None, aligned)
        di = h.DevMatrix.from_numpy(idx, pitched=False)
        h.call("hipF_randomize", None, y.ptr, a.ptr, di.ptr, y.dim, a.dim)
        return tag, y.check(x[idx], 0, 0, op)
    if op == "gemm":
        M, N, K = int(rs.randint(1, 300)), int(rs.randint(1, 700)), int(rs.randint(1, 900))
        ta, tb = bool(rs.rand() < 0.5), bool(rs.rand() < 0.5)
        A = rs.standard_normal((K, M) if ta else (M, K)).astype(f32)
        B = rs.standard_normal((N, K) if tb else (K, N)).astype(f32)
        C0 = rs.standard_normal((M, N)).astype(f32)
        al, be = f32(rs.uniform(-1, 1)), f32(rs.choice([0.0, 1.0, -0.5]))
        wa, wb, wc = Window(rs, A.shape[0], A.shape[1], A, aligned), Window(rs, B.shape[0], B.shape[1], B, aligned), Window(rs, M, N, C0, aligned)
        h.call("hipF_gemm", None, 112 if ta else 111, 112 if tb else 111, M, N, K, al, wa.ptr, wa.S, wb.ptr, wb.S, be, wc.ptr, wc.S)
        want = al * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)) + be * C0
        tag = "gemm %s%s %dx%dx%d%s" % ("T" if ta else "N", "T" if tb else "N", M, N, K, " aligned" if aligned else "")
        return tag, wc.check(want.astype(f32), 1e-4, 1e-4 * np.sqrt(K), "gemm")
    if op == "unary":
        which = rs.choice(["set_const", "binarize", "add_const", "apply_floor", "apply_truncate", "apply_exp", "apply_log", "invert_elements", "round"])
        data = np.abs(x) + f32(0.5) if which in ("apply_log", "invert_elements") else x
        a = Window(rs, rows, cols, data, aligned)
        args = {"set_const": (f32(2.5),), "binarize": (f32(0.1),), "add_const": (f32(-1.5),), "apply_floor": (f32(0.2),),
                "apply_truncate": (f32(-0.3), f32(0.4)), "apply_exp": (), "apply_log": (), "invert_elements": (), "round": ()}[which]
        h.call("hipF_" + which, None, a.ptr, *args, a.dim)
        want = {"set_const": lambda: np.full_like(x, 2.5), "binarize": lambda: (x > f32(0.1)).astype(f32), "add_const": lambda: x - f32(1.5),
                "apply_floor": lambda: np.maximum(x, f32(0.2)), "apply_truncate": lambda: np.clip(x, f32(-0.3), f32(0.4)),
                "apply_exp": lambda: np.exp(x), "apply_log": lambda: np.log(data), "invert_elements": lambda: f32(1) / data,
                "round": lambda: np.where(x >= 0, np.floor(x + f32(0.5)), np.ceil(x - f32(0.5)))}[which]()
        return tag + " " + which, a.check(want.astype(f32), 2e-6, 2e-6, which)
    if op == "vec_scale":
        which = rs.choice(["mul_cols_vec", "mul_rows_vec", "div_cols_vec", "div_rows_vec", "add_vec_to_cols"])
        a = Window(rs, rows, cols, x, aligned)
        per_col = which in ("mul_cols_vec", "div_cols_vec")
        v = (np.abs(rs.standard_normal(cols if per_col else rows)) + 0.5).astype(f32)
        dv = h.DevMatrix.from_numpy(v, pitched=False)
        vb = v[None, :] if per_col else v[:, None]
        if which == "add_vec_to_cols":
            h.call("hipF_add_vec_to_cols", None, f32(0.5), dv.ptr, f32(1.0), a.ptr, a.dim)
            want = f32(0.5) * vb + x
        else:
            h.call("hipF_" + which, None, a.ptr, dv.ptr, a.dim)
            want = x * vb if which.startswith("mul") else x * (f32(1) / vb)
        return tag + " " + which, a.check(want.astype(f32), 2e-6, 2e-6, which)
    if op == "part_ops":
        a = Window(rs, rows, cols, x, aligned)
        r, c = int(rs.randint(1, rows + 1)), int(rs.randint(1, cols + 1))
        ro, co = int(rs.randint(0, rows - r + 1)), int(rs.randint(0, cols - c + 1))
        want = x.copy()
        if rs.rand() < 0.5:
            h.call("hipF_part_set_const", None, a.ptr, f32(3.0), a.dim, ro, r, co, c)
            want[ro:ro + r, co:co + c] = 3.0
            return tag + " part_set_const", a.check(want, 0, 0, "part_set_const")
        row = rs.standard_normal(c).astype(f32)
        dr = h.DevMatrix.from_numpy(row, pitched=False)
        h.call("hipF_add_vec_to_partial_rows", None, f32(2.0), co, dr.ptr, c, f32(1.0), a.ptr, a.dim)
        want[:, co:co + c] += f32(2.0) * row[None, :]
        return tag + " add_vec_to_partial_rows", a.check(want, 2e-6, 2e-6, "add_vec_to_partial_rows")
    if op == "l2bound":
        a = Window(rs, rows, cols, x, aligned)
        bound = f32(0.7 * np.sqrt(cols))
        h.call("hipF_apply_row_l2upperbound", None, a.ptr, bound, a.dim)
        return tag, a.check(O.apply_row_l2_upper_bound(x.copy(), bound), 2e-5, 2e-6, op)
    if op == "l1":
        g0 = rs.standard_normal((rows, cols)).astype(f32)
        w0 = (0.01 * x).astype(f32)
        w, g = Window(rs, rows, cols, w0, aligned), None
        g = w.like(rs, g0)
        l1, lr = f32(0.02), f32(0.1)
        h.call("hipF_regularize_l1", None, w.ptr, g.ptr, l1, lr, w.dim)
        ww, gg = O.regularize_l1(w0.copy(), g0.copy(), l1, lr)
        return tag, w.check(ww, 2e-6, 1e-7, op + " (weights)") or g.check(gg, 2e-6, 1e-7, op + " (gradient)")
    if op == "diff_xent":
        p = O.softmax(2 * x)
        tgt = rs.randint(0, cols, rows).astype(np.int32)
        a = Window(rs, rows, cols, p, aligned)
        lp = h.DevMatrix(1, rows, f32, pitched=False)
        h.call("hipF_diff_xent", None, h.DevMatrix.from_numpy(tgt, pitched=False).ptr, a.ptr, lp.ptr, a.dim)
        want, wlp = O.diff_xent(tgt, p.copy())
        msg = a.check(want, 1e-6, 1e-7, op)
        if msg is None and not np.allclose(lp.numpy()[0], wlp, rtol=1e-5, atol=1e-6):
            msg = "log posteriors differ"
        return tag, msg
    if op == "expand":
        cols_in = max(1, min(cols, 60))
        xin = rs.standard_normal((rows, cols_in)).astype(f32)
        offs = rs.randint(-7, 8, int(rs.randint(1, 12))).astype(np.int32)
        a = Window(rs, rows, cols_in, xin, aligned)
        y = Window(rs, rows, cols_in * len(offs), None, aligned)
        h.call("hipF_expand", None, y.ptr, a.ptr, h.DevMatrix.from_numpy(offs, pitched=False).ptr, y.dim, a.dim)
        return tag + " x%d" % len(offs), y.check(O.expand(xin, offs), 0, 0, op)
    if op == "copy":
        n_out = int(rs.randint(1, 2 * cols + 2))
        idx = rs.randint(0, cols, n_out).astype(np.int32)
        a = Window(rs, rows, cols, x, aligned)
        y = Window(rs, rows, n_out, None, aligned)
        h.call("hipF_copy", None, y.ptr, a.ptr, h.DevMatrix.from_numpy(idx, pitched=False).ptr, y.dim, a.dim)
        return tag, y.check(x[:, idx], 0, 0, op)
    if op == "mse_eval":
        b0 = rs.standard_normal((rows, cols)).astype(f32)
        a = Window(rs, rows, cols, x, aligned)
        b = a.like(rs, b0)
        dd = a.like(rs)
        rowsq, loss = h.DevMatrix(1, rows, f32, pitched=False), h.DevMatrix(1, 1, np.float64, pitched=False)
        h.call("hipF_mse_eval", None, a.ptr, b.ptr, dd.ptr, rowsq.ptr, loss.ptr, a.dim)
        rdiff, rrow, rloss = O.mse_eval(x, b0)
        msg = dd.check(rdiff, 1e-6, 1e-6, op)
        if msg is None and abs(loss.numpy()[0, 0] - rloss) > 1e-5 * max(rloss, 1e-3):
            msg = "loss %.8g want %.8g" % (loss.numpy()[0, 0], rloss)
        return tag, msg
    if op == "nonfinite":
        n_nan, n_inf = int(rs.randint(0, 4)), int(rs.randint(0, 4))
        flat = x.reshape(-1)
        pos = rs.permutation(flat.size)[:n_nan + n_inf]
        flat[pos[:n_nan]] = np.nan
        flat[pos[n_nan:]] = np.inf
        a = Window(rs, rows, cols, x, aligned)
        cnt = h.DevMatrix(1, 2, np.int32, pitched=False)
        h.call("hipF_count_nonfinite", None, a.ptr, a.dim, cnt.ptr)
        want = [int(np.isnan(x).sum()), int(np.isposinf(x).sum())]
        return tag, None if cnt.numpy()[0].tolist() == want else "counts %s want %s" % (cnt.numpy()[0].tolist(), want)
    if op == "relu_family":
        which = rs.choice(["relu", "softrelu", "diff_relu", "diff_softrelu"])
        a = Window(rs, rows, cols, 3 * x, aligned)
        y = a.like(rs)
        if which == "relu":
            h.call("hipF_relu", None, y.ptr, a.ptr, a.dim)
            want = O.Relu(cols).propagate(3 * x)
        elif which == "softrelu":
            h.call("hipF_softrelu", None, y.ptr, a.ptr, a.dim)
            want = O.SoftRelu(cols).propagate(3 * x)
        else:
            e0 = rs.standard_normal((rows, cols)).astype(f32)
            e = a.like(rs, e0)
            h.call("hipF_" + which, None, y.ptr, e.ptr, a.ptr, a.dim)
            want = (O.Relu(cols).backpropagate(e0, y_out=np.maximum(3 * x, 0)) if which == "diff_relu"
                    else O.SoftRelu(cols).backpropagate(e0, y_in=3 * x))
            if which == "diff_relu":   # the entry point takes the layer's output y; a holds 3x here, same sign pattern
                want = (e0 * ((3 * x) > 0)).astype(f32)
        return tag + " " + which, y.check(want, 2e-5, 2e-6, which)
    if op == "softmax_part":
        ids = x.argmax(axis=1).astype(np.int32)
        a = Window(rs, rows, cols, x, aligned)
        y = a.like(rs)
        h.call("hipF_softmax_part", None, a.ptr, h.DevMatrix.from_numpy(ids, pitched=False).ptr, y.ptr, a.dim)
        return tag, y.check(np.exp(x - x[np.arange(rows), ids][:, None]), 2e-6, 1e-7, op)
    if op == "rng":
        rows, cols = min(rows, 300), min(cols, 700)     # (the oracle's generator is numpy code over whole arrays)
        which = rs.choice(["rand", "binarize_probs_rand", "add_gauss_noise"])
        probs = rs.random_sample((rows, cols)).astype(f32)
        a = Window(rs, rows, cols, probs, aligned)
        out = a.like(rs, probs)
        # generator states laid out like the window's buffer: one state per buffer element
        zs = [rs.randint(128, 2 ** 31 - 1, (a.R, a.S)).astype(np.uint32) for _ in range(4)]
        dz = [h.DevMatrix.from_numpy(z, pitched=False) for z in zs]
        zp = [ctypes.c_void_p(d.ptr.value + 4 * (a.r0 * a.S + a.c0)) for d in dz]
        zwin = [z[a.r0:a.r0 + rows, a.c0:a.c0 + cols] for z in zs]
        tag = "%s %dx%d%s" % (which, rows, cols, " aligned" if aligned else "")
        if which == "rand":
            h.call("hipF_rand", None, out.ptr, *zp, a.dim)
            u, znew = O.hybrid_taus(*zwin)
            msg = out.check(u, 0, 0, which)
        elif which == "binarize_probs_rand":
            h.call("hipF_binarize_probs_rand", None, out.ptr, a.ptr, *zp, a.dim)
            u, znew = O.hybrid_taus(*zwin)
            msg = out.check(O.binarize_probs(probs, u), 0, 0, which)
        else:
            h.call("hipF_add_gauss_noise", None, out.ptr, f32(0.5), *zp, a.dim)
            g, znew = O.box_muller(*zwin)
            msg = out.check((probs + f32(0.5) * g).astype(f32), 1e-4, 1e-5, which)
        for d, z0, zn in zip(dz, zs, znew):
            if msg is None:
                want = z0.copy()
                want[a.r0:a.r0 + rows, a.c0:a.c0 + cols] = zn
                if not (d.numpy() == want).all():
                    msg = "generator states differ (or were advanced outside the window)"
        return tag, msg
    if op == "rbm_update":
        N, H, V = int(rs.choice([1, 7, 64, 256])), int(rs.randint(1, 500)), int(rs.randint(1, 500))
        if rs.rand() < 0.3:
            H, V = int(rs.choice([1024, 2048])), int(rs.choice([440, 2048]))
        tag = "rbm_update N=%d H=%d V=%d%s" % (N, H, V, " aligned" if aligned else "")
        rbm = O.Rbm((0.1 * rs.standard_normal((H, V))).astype(f32), (0.1 * rs.standard_normal(V)).astype(f32), (0.1 * rs.standard_normal(H)).astype(f32))
        rbm.W_corr = (0.01 * rs.standard_normal((H, V))).astype(f32)
        rbm.vis_bias_corr = (0.01 * rs.standard_normal(V)).astype(f32)
        rbm.hid_bias_corr = (0.01 * rs.standard_normal(H)).astype(f32)
        rbm.learn_rate, rbm.momentum, rbm.l2_penalty = f32(0.05), f32(rs.choice([0.0, 0.5])), f32(rs.choice([0.0, 2e-3]))
        pv, nv = rs.random_sample((N, V)).astype(f32), rs.random_sample((N, V)).astype(f32)
        ph, nh = rs.random_sample((N, H)).astype(f32), rs.random_sample((N, H)).astype(f32)
        wpv, wnv = Window(rs, N, V, pv, aligned), None
        wnv = wpv.like(rs, nv)                      # (the fused error needs one pitch for the two visible matrices)
        wph, wnh = Window(rs, N, H, ph, aligned), Window(rs, N, H, nh, aligned)
        wW, wdW = Window(rs, H, V, rbm.W, aligned), Window(rs, H, V, rbm.W_corr, aligned)
        dvb, ddvb = h.DevMatrix.from_numpy(rbm.vis_bias, pitched=False), h.DevMatrix.from_numpy(rbm.vis_bias_corr, pitched=False)
        dhb, ddhb = h.DevMatrix.from_numpy(rbm.hid_bias, pitched=False), h.DevMatrix.from_numpy(rbm.hid_bias_corr, pitched=False)
        args = [None, N, H, V, wpv.ptr, wpv.S, wph.ptr, wph.S, wnv.ptr, wnv.S, wnh.ptr, wnh.S, rbm.learn_rate, rbm.momentum, rbm.l2_penalty,
                wW.ptr, wW.S, wdW.ptr, wdW.S, dvb.ptr, ddvb.ptr, dhb.ptr, ddhb.ptr]
        with_mse = rs.rand() < 0.5
        if with_mse:
            loss = h.DevMatrix(1, 1, np.float64, pitched=False)
            h.call("hipF_rbm_update_mse", *args, loss.ptr)
        else:
            h.call("hipF_rbm_update", *args)
        rbm.rbm_update(pv, ph, nv, nh)
        msg = (wW.check(rbm.W, 2e-4, 2e-6, "rbm (W)") or wdW.check(rbm.W_corr, 2e-4, 2e-6, "rbm (dW)") or
               wpv.check(pv, 0, 0, "rbm (pos_vis)") or wnh.check(nh, 0, 0, "rbm (neg_hid)"))
        for name, d, want in (("vis_bias", dvb, rbm.vis_bias), ("dvis_bias", ddvb, rbm.vis_bias_corr), ("hid_bias", dhb, rbm.hid_bias),
                              ("dhid_bias", ddhb, rbm.hid_bias_corr)):
            if msg is None and not np.allclose(d.numpy()[0], want, rtol=2e-4, atol=2e-6):
                msg = "%s differs: max abs %.3g" % (name, np.abs(d.numpy()[0] - want).max())
        if msg is None and with_mse:
            want = 0.5 * ((nv.astype(np.float64) - pv) ** 2).sum()
            if abs(loss.numpy()[0, 0] - want) > 1e-5 * max(want, 1e-3):
                msg = "reconstruction error %.8g want %.8g" % (loss.numpy()[0, 0], want)
        return tag + (" +mse" if with_mse else ""), msg
    if op == "xent_eval":
        p = O.softmax(2 * x)
        tgt = rs.randint(0, cols, rows).astype(np.int32)
        a = Window(rs, rows, cols, p, aligned)
        dq = a.like(rs)
        dt = h.DevMatrix.from_numpy(tgt, pitched=False)
        lp, mid = h.DevMatrix(1, rows, f32, pitched=False), h.DevMatrix(1, rows, np.int32, pitched=False)
        cnt, loss = h.DevMatrix(1, 2, np.int64, pitched=False), h.DevMatrix(1, 1, np.float64, pitched=False)
        h.call("hipF_xent_eval", None, a.ptr, dt.ptr, dq.ptr, lp.ptr, mid.ptr, cnt.ptr, loss.ptr, a.dim)
        want = p.copy()
        want[np.arange(rows), tgt] -= 1
        msg = dq.check(want, 1e-6, 1e-7, op) or a.check(p, 0, 0, op + " (input)")
        if msg is None and not (mid.numpy()[0] == p.argmax(axis=1)).all():
            msg = "arg-max differs"
        if msg is None and cnt.numpy()[0].tolist() != [int((p.argmax(axis=1) == tgt).sum()), rows]:
            msg = "counters %s" % cnt.numpy()[0].tolist()
        ref_loss = -np.log(np.maximum(p[np.arange(rows), tgt], 1e-20).astype(np.float64)).sum()
        if msg is None and abs(loss.numpy()[0, 0] - ref_loss) > 1e-5 * max(ref_loss, 1.0):
            msg = "loss %.8g want %.8g" % (loss.numpy()[0, 0], ref_loss)
        return tag, msg
    if op.startswith("affine"):
        N, I, Od = int(rs.choice([1, 7, 64, 256, 300])), int(rs.randint(1, 700)), int(rs.randint(1, 900))
        if rs.rand() < 0.3:
            I, Od = int(rs.choice([440, 1024, 1320])), int(rs.choice([1024, 1483, 2048]))
        W = (rs.standard_normal((Od, I)) / np.sqrt(I)).astype(f32)
        b = (0.1 * rs.standard_normal(Od)).astype(f32)
        xin = rs.standard_normal((N, I)).astype(f32)
        tag = "%s N=%d I=%d O=%d%s" % (op, N, I, Od, " aligned" if aligned else "")
        wW, wx = Window(rs, Od, I, W, aligned), Window(rs, N, I, xin, aligned)
        db = h.DevMatrix.from_numpy(b, pitched=False)
        if op == "affine_forward":
            act = int(rs.choice([0, 1, 2]))
            out = Window(rs, N, Od, None, aligned)
            h.call("hipF_affine_forward", None, act, 112, N, Od, I, wx.ptr, wx.S, wW.ptr, wW.S, db.ptr, out.ptr, out.S)
            z = (xin.astype(np.float64) @ W.T.astype(np.float64) + b).astype(f32)
            want = z if act == 0 else O.sigmoid(z) if act == 1 else O.softmax(z)
            return tag + " act=%d" % act, out.check(want, 2e-4, 2e-5, op) or wW.check(W, 0, 0, op + " (weights)")
        if op == "affine_forward_xent":
            tgt = rs.randint(0, Od, N).astype(np.int32)
            post = Window(rs, N, Od, None, aligned)
            diff = post.like(rs)          # (the entry point wants one row pitch for both)
            dt = h.DevMatrix.from_numpy(tgt, pitched=False)
            cnt, loss = h.DevMatrix(1, 2, np.int64, pitched=False), h.DevMatrix(1, 1, np.float64, pitched=False)
            wp = int(rs.rand() < 0.5)
            h.call("hipF_affine_forward_xent", None, 112, N, Od, I, wx.ptr, wx.S, wW.ptr, wW.S, db.ptr, post.ptr, post.S, wp,
                   dt.ptr, diff.ptr, diff.S, None, None, cnt.ptr, loss.ptr)
            z = (xin.astype(np.float64) @ W.T.astype(np.float64) + b).astype(f32)
            pr = O.softmax(z)
            want = pr.copy()
            want[np.arange(N), tgt] -= 1
            msg = diff.check(want, 2e-4, 2e-6, op + " (diff)") or post.check(pr if wp else None, 2e-4, 2e-7, op + " (posteriors)")
            if msg is None and int(cnt.numpy()[0, 1]) != N:
                msg = "frame counter %s" % cnt.numpy()[0].tolist()
            return tag + " write_post=%d" % wp, msg
        e = rs.standard_normal((N, Od)).astype(f32)
        we = Window(rs, N, Od, e, aligned)
        if op == "affine_backward":
            use_y = rs.rand() < 0.6
            yv = rs.random_sample((N, I)).astype(f32)
            wy = Window(rs, N, I, yv, aligned)
            out = Window(rs, N, I, None, aligned)
            h.call("hipF_affine_backward", None, N, Od, I, we.ptr, we.S, wW.ptr, wW.S, wy.ptr if use_y else None, wy.S, out.ptr, out.S)
            want = (e.astype(np.float64) @ W.astype(np.float64)).astype(f32)
            if use_y:
                want = want * yv * (1 - yv)
            return tag + (" y" if use_y else ""), out.check(want, 2e-4, 2e-4, op)
        # affine_update: in place, momentum buffer and bias
        corr0 = (0.01 * rs.standard_normal((Od, I))).astype(f32)
        bc0 = (0.01 * rs.standard_normal(Od)).astype(f32)
        wc = Window(rs, Od, I, corr0, aligned)
        dbc = h.DevMatrix.from_numpy(bc0, pitched=False)
        g, mom, lr = f32(1.0 / N if rs.rand() < 0.5 else 1.0), f32(rs.choice([0.0, 0.9])), f32(0.01)
        h.call("hipF_affine_update_fused", None, N, Od, I, we.ptr, we.S, wx.ptr, wx.S, g, mom, lr, f32(0.0), wW.ptr, wW.S, wW.ptr, wW.S,
               wc.ptr, wc.S, db.ptr, dbc.ptr)
        corr = g * (e.T.astype(np.float64) @ xin.astype(np.float64)) + mom * corr0
        bc = g * e.astype(np.float64).sum(axis=0) + mom * bc0
        tol = 2e-4 * max(1.0, float(np.abs(corr).max()))
        msg = wc.check(corr.astype(f32), 2e-4, tol, op + " (correction)") or wW.check((W - lr * corr).astype(f32), 2e-4, lr * tol + 1e-6, op + " (weights)")
        if msg is None and not np.allclose(dbc.numpy()[0], bc, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(bc).max())):
            msg = "bias correction differs"
        if msg is None and not np.allclose(db.numpy()[0], b - lr * bc, rtol=2e-4, atol=1e-5):
            msg = "bias differs"
        return tag, msg
    return tag, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    h.init(-1)
    bad = 0
    for i in range(args.cases):
        rs = np.random.RandomState(args.seed * 100000 + i)
        tag, msg = one_case(rs)
        if msg:
            bad += 1
            print("case %d %s: %s" % (i, tag, msg), flush=True)
        elif i % 50 == 0:
            print("case %d %s: ok" % (i, tag), flush=True)
    print("%d of %d cases differ" % (bad, args.cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
