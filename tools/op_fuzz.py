#!/usr/bin/env python3
"""Random-shape check of the C-ABI kernels on WINDOWS of larger device buffers: an operand is a window (first row,
first column, rows, columns) of a buffer with its own row pitch, so pointers are often not 16-byte aligned, row
lengths are odd, and there is memory all around the result.  Each call is compared with numpy on the window AND
the rest of the buffer must come back untouched (a write outside the window is an error even where the values are
right).  Shapes include one row, one column, cache-sized matrices (the flat streaming kernels) and widths around
the kernels' internal limits.     python tools/op_fuzz.py [--cases 300] [--seed 1]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
                    "affine_update", "affine_forward_xent"])
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
