"""CPU restatement (numpy) of the reference's nnet1 hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (kaldi_amd/) never does.  Every function follows the CPU branch of the reference
(the `else` arm taken when CuDevice is disabled, i.e. the path the reference runs without a
GPU) and cites the lines it restates.  Paths are relative to /root/reference/src.

Pinned by: tests/golden/*.npz, produced by oracle/ref_dump.cc linked against the reference's
own objects (oracle/Makefile), see tests/test_oracle_golden.py.

Arithmetic conventions of the reference CPU path kept here:
  * BaseFloat = float32 (base/kaldi-types.h:27); matrices row-major.
  * GEMM = cblas_sgemm (matrix/kaldi-matrix.cc:206-222): float32 in/out; the summation order
    inside BLAS is unspecified, so products are formed in float64 and rounded once -- within
    a few float32 ulp of any BLAS order.
  * AddRowSumMat / AddColSumMat use a double accumulator (matrix/kaldi-vector.cc:566-593).
  * ApplySoftMax sums exp() in float32, in index order (matrix/kaldi-vector.cc:677-685).
"""
import ctypes
import ctypes.util
import struct

import numpy as np

f32 = np.float32


# ------------------------------------------------------------------------------------------------
# cu:: math, CPU branches
# ------------------------------------------------------------------------------------------------
def sigmoid(x):
    """cudamatrix/cu-math-inl.h:167-173: y = 1.0/(1.0+exp(-x)) evaluated in double, stored float."""
    x = np.asarray(x, f32)
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


def diff_sigmoid(ein, y):
    """cudamatrix/cu-math-inl.h:195-202: eout = ein * y*(1.0-y) (double intermediate)."""
    ein = np.asarray(ein, f32).astype(np.float64)
    y = np.asarray(y, f32).astype(np.float64)
    return (ein * y * (1.0 - y)).astype(f32)


def softmax(x):
    """cudamatrix/cu-math-inl.h:244-250 -> VectorBase::ApplySoftMax, matrix/kaldi-vector.cc:677-685:
    max, then sum += (d[i] = exp(d[i]-max)) in float32 index order, then Scale(1.0/sum)."""
    x = np.asarray(x, f32)
    out = np.empty_like(x)
    for r in range(x.shape[0]):
        row = x[r]
        mx = row.max()
        e = np.exp((row - mx).astype(f32)).astype(f32)
        s = f32(0.0)
        # float32 running sum in index order; cumsum keeps the sequential rounding
        s = np.cumsum(e, dtype=f32)[-1] if e.size else f32(0)
        out[r] = e * f32(1.0 / s)
    return out


def find_row_max_id(mat):
    """cudamatrix/cu-math-inl.h:335-349: max=-1e21, id=-1; strict '<' keeps the FIRST maximum."""
    mat = np.asarray(mat, f32)
    ids = np.full(mat.shape[0], -1, np.int32)
    for r in range(mat.shape[0]):
        row = mat[r]
        if row.size == 0:
            continue
        c = int(np.argmax(row))  # first occurrence of the maximum
        if row[c] > f32(-1e21):
            ids[r] = c
    return ids


def diff_xent(tgt, net_out):
    """cudamatrix/cu-math-inl.h:372-378: log_post[r] = log(out[r,t]); out[r,t] -= 1.0.
    (The GPU kernel floors the posterior at 1e-20 before the log, cu-kernels.cu:741; the CPU
    branch does not.  The floor is applied here too: it only matters when a posterior
    underflows, where the CPU branch would return -inf.)"""
    diff = np.array(net_out, f32, copy=True)
    rows = np.arange(diff.shape[0])
    p = diff[rows, tgt]
    log_post = np.log(np.maximum(p, f32(1e-20))).astype(f32)
    diff[rows, tgt] = p - f32(1.0)
    return diff, log_post


def randomize(src, copy_from):
    """cudamatrix/cu-math-inl.h:407-413: tgt.Row(i) = src.Row(copy_from[i])."""
    return np.asarray(src, f32)[np.asarray(copy_from)]


def expand(src, frame_offsets):
    """cudamatrix/cu-math-inl.h:439-451: splice with edge clamping."""
    src = np.asarray(src, f32)
    T = src.shape[0]
    parts = []
    for off in frame_offsets:
        idx = np.clip(np.arange(T) + int(off), 0, T - 1)
        parts.append(src[idx])
    return np.concatenate(parts, axis=1)


def copy_cols(src, copy_from_indices):
    """cudamatrix/cu-math-inl.h:477-486: tgt(r,c) = src(r, idx[c])."""
    return np.asarray(src, f32)[:, np.asarray(copy_from_indices)]


def regularize_l1(wei, grad, l1, lr):
    """cudamatrix/cu-math-inl.h:270-290."""
    wei = np.array(wei, f32, copy=True)
    grad = np.array(grad, f32, copy=True)
    l1 = f32(l1)
    lr = f32(lr)
    l1_signed = np.where(wei < 0, -l1, l1).astype(f32)
    after = (wei - lr * grad - l1_signed).astype(f32)
    flip = ((after > 0) ^ (wei > 0)) & (wei != 0)
    keep = (~flip) & (wei != 0)
    new_w = wei.copy()
    new_w[flip] = 0
    grad[flip] = 0
    new_w[keep] = (wei - l1_signed)[keep]
    return new_w, grad


# ------------------------------------------------------------------------------------------------
# CuMatrix / CuVector ops, CPU branches (matrix/kaldi-matrix.cc, kaldi-vector.cc)
# ------------------------------------------------------------------------------------------------
def gemm(alpha, A, transA, B, transB, beta, C):
    """CuMatrix::AddMatMat -> MatrixBase::AddMatMat -> cblas_sgemm (matrix/kaldi-matrix.cc:206-222)."""
    A = np.asarray(A, f32).astype(np.float64)
    B = np.asarray(B, f32).astype(np.float64)
    if transA:
        A = A.T
    if transB:
        B = B.T
    prod = (A @ B).astype(f32)
    if beta == 0:
        return (f32(alpha) * prod).astype(f32)
    return (f32(alpha) * prod + f32(beta) * np.asarray(C, f32)).astype(f32)


def add_row_sum_mat(alpha, M, beta, vec):
    """VectorBase::AddRowSumMat, matrix/kaldi-vector.cc:566-578 (double accumulator over rows)."""
    s = np.asarray(M, f32).astype(np.float64).sum(axis=0)
    return (f32(alpha) * s + f32(beta) * np.asarray(vec, f32)).astype(f32)


def add_col_sum_mat(alpha, M, beta, vec):
    """VectorBase::AddColSumMat, matrix/kaldi-vector.cc:580-593."""
    s = np.asarray(M, f32).astype(np.float64).sum(axis=1)
    return (f32(alpha) * s + f32(beta) * np.asarray(vec, f32)).astype(f32)


# ------------------------------------------------------------------------------------------------
# nnet components
# ------------------------------------------------------------------------------------------------
class BiasedLinearity:
    """nnet/nnet-biasedlinearity.h:26-157.  linearity_ is (dim_out x dim_in)."""
    marker = "<biasedlinearity>"
    updatable = True

    def __init__(self, W, b):
        self.W = np.array(W, f32)
        self.b = np.array(b, f32)
        self.W_corr = np.zeros_like(self.W)  # zero at construction (CuMatrix::Resize zero-fills)
        self.b_corr = np.zeros_like(self.b)
        self.learn_rate = f32(0)
        self.momentum = f32(0)
        self.l2_penalty = f32(0)
        self.l1_penalty = f32(0)
        self.l2_upper_bound = f32(0)
        self.average_grad = False

    @property
    def dim_out(self):
        return self.W.shape[0]

    @property
    def dim_in(self):
        return self.W.shape[1]

    def propagate(self, x):
        """:58-63  out = bias (rows); out += in * W^T."""
        out = np.broadcast_to(self.b, (x.shape[0], self.b.size)).astype(f32)
        return gemm(1.0, x, False, self.W, True, 1.0, out)

    def backpropagate(self, err, y_in=None, y_out=None):
        """:65-69  out_err = in_err * W."""
        return gemm(1.0, err, False, self.W, False, 0.0, None)

    def update(self, x, err):
        """:71-102."""
        n = x.shape[0]
        g = f32(1.0 / n) if self.average_grad else f32(1.0)
        self.W_corr = gemm(g, err, True, x, False, self.momentum, self.W_corr)
        self.b_corr = add_row_sum_mat(g, err, self.momentum, self.b_corr)
        if self.l2_penalty != 0:
            l2 = f32(self.learn_rate * self.l2_penalty * n)
            self.W = (f32(-l2) * self.W + self.W).astype(f32)
        if self.l1_penalty != 0:
            l1 = f32(self.learn_rate * n * self.l1_penalty)
            self.W, self.W_corr = regularize_l1(self.W, self.W_corr, l1, self.learn_rate)
        self.W = (f32(-self.learn_rate) * self.W_corr + self.W).astype(f32)
        self.b = (f32(-self.learn_rate) * self.b_corr + self.b).astype(f32)
        if self.l2_upper_bound > 0:
            self.W = apply_row_l2_upper_bound(self.W, self.l2_upper_bound)


def apply_row_l2_upper_bound(W, bound):
    """cudamatrix/cu-matrix-inl.h (CPU branch of ApplyRowL2UpperBound) / cu-kernels.cu:431-454."""
    W = np.array(W, f32, copy=True)
    for j in range(W.shape[0]):
        row = W[j].astype(np.float64)
        max_mag = np.abs(row).max()
        l2 = np.sqrt(((row / max_mag) ** 2).sum())
        if max_mag * l2 > bound:
            W[j] = (((row / max_mag) / l2) * bound).astype(f32)
    return W


class Sigmoid:
    """nnet/nnet-activation.h:195-219."""
    marker = "<sigmoid>"
    updatable = False

    def __init__(self, dim):
        self.dim_in = self.dim_out = dim

    def propagate(self, x):
        return sigmoid(x)

    def backpropagate(self, err, y_in=None, y_out=None):
        return diff_sigmoid(err, y_out)  # y = PropagateBuffer()[idx+1]


class Softmax:
    """nnet/nnet-activation.h:221-247.  Backpropagate copies the error."""
    marker = "<softmax>"
    updatable = False

    def __init__(self, dim):
        self.dim_in = self.dim_out = dim

    def propagate(self, x):
        return softmax(x)

    def backpropagate(self, err, y_in=None, y_out=None):
        return np.array(err, f32, copy=True)


class Expand:
    """nnet/nnet-various.h:35-80."""
    marker = "<expand>"
    updatable = False

    def __init__(self, dim_in, dim_out, offsets):
        self.dim_in, self.dim_out = dim_in, dim_out
        self.offsets = [int(round(o)) for o in offsets]

    def propagate(self, x):
        return expand(x, self.offsets)


class Copy:
    """nnet/nnet-various.h:87-135 (indices are 1-based on disk, 0-based here)."""
    marker = "<copy>"
    updatable = False

    def __init__(self, dim_in, dim_out, indices0):
        self.dim_in, self.dim_out = dim_in, dim_out
        self.indices = [int(i) for i in indices0]

    def propagate(self, x):
        return copy_cols(x, self.indices)


class Relu:
    """nnet/nnet-activation.h:123-147 over cu::Relu / cu::DiffRelu, CPU branch (cudamatrix/cu-math-inl.h:39-90):
    y = x > 0 ? x : 0 ; eout = ein * (y > 0), y = this layer's output.  (The reference's GPU wrappers send
    cuda_diff_relu to the diff_sigmoid kernel, cu-kernels.h:143-145; the kernels themselves, cu-kernels.cu:526-545,
    compute what the CPU branch does, and that is what is restated here.)"""
    marker = "<relu>"
    updatable = False

    def __init__(self, dim):
        self.dim_in = self.dim_out = dim

    def propagate(self, x):
        x = np.asarray(x, f32)
        return np.where(x > 0, x, f32(0)).astype(f32)

    def backpropagate(self, err, y_in=None, y_out=None):
        return (np.asarray(err, f32) * (np.asarray(y_out, f32) > 0)).astype(f32)


class SoftRelu:
    """nnet/nnet-activation.h:149-174 over cu::SoftRelu / cu::DiffSoftRelu, CPU branch (cu-math-inl.h:92-147):
    y = x > 4 ? x : log(1 + exp(x)) ; eout = x > 4 ? ein : ein * e^x / (1 + e^x), x = this layer's INPUT.  The
    C++ expressions mix float operands with double literals: computed in double, rounded on assignment."""
    marker = "<softrelu>"
    updatable = False

    def __init__(self, dim):
        self.dim_in = self.dim_out = dim

    def propagate(self, x):
        x = np.asarray(x, f32)
        soft = np.log(1.0 + np.exp(x.astype(np.float32)).astype(np.float64))   # exp(float) -> float, 1 + . in double, log in double
        return np.where(x > 4.0, x, soft.astype(f32)).astype(f32)

    def backpropagate(self, err, y_in=None, y_out=None):
        x, e = np.asarray(y_in, f32), np.asarray(err, f32)
        f = np.exp(x)                                                            # Real f = exp(x)
        soft = (e * f).astype(f32).astype(np.float64) / (1.0 + f.astype(np.float64))   # ein * f / (1.0 + f)
        return np.where(x > 4.0, e, soft.astype(f32)).astype(f32)


class Scale:
    """nnet/nnet-activation.h:27-70: y = s * x both ways; one float of data.  (The marker carries the reference's
    typo, nnet-component.cc:44.)"""
    marker = ",<scale>"
    updatable = False

    def __init__(self, dim, scale=1.0):
        self.dim_in = self.dim_out = dim
        self.scale = f32(scale)

    def propagate(self, x):
        return (np.asarray(x, f32) * self.scale).astype(f32)

    def backpropagate(self, err, y_in=None, y_out=None):
        return (np.asarray(err, f32) * self.scale).astype(f32)


class Dropout:
    """nnet/nnet-activation.h:72-121: y = mask .* x, the same mask for the error; mask = (1 - ratio > uniform).
    The draws are injected (`uniform`): the reference's CPU branch of CuRand::BinarizeProbs never sizes its state
    (cu-rand-inl.h:162-168) and cannot run this component, so nothing pins the stream itself."""
    marker = "<dropout>"
    updatable = False

    def __init__(self, dim, ratio=0.5):
        self.dim_in = self.dim_out = dim
        self.ratio = f32(ratio)
        self.mask = None
        self.uniform = None

    def propagate(self, x):
        x = np.asarray(x, f32)
        self.mask = binarize_probs(np.full(x.shape, f32(1) - self.ratio, f32), self.uniform)
        return (x * self.mask).astype(f32)

    def backpropagate(self, err, y_in=None, y_out=None):
        return (np.asarray(err, f32) * self.mask).astype(f32)


class Nnet:
    """nnet/nnet-nnet.{h,cc}."""

    def __init__(self, layers):
        self.layers = list(layers)
        self.propagate_buf = []
        self.backpropagate_buf = []

    def set_train_opts(self, learn_rate, momentum=0.0, l2_penalty=0.0, l1_penalty=0.0, l2_upper_bound=0.0,
                       average_grad=False, learn_factors=None):
        """nnet-nnet.cc:165-203 and nnet-nnet.h:184-224."""
        k = 0
        for l in self.layers:
            if l.updatable:
                lr = f32(learn_rate)
                if learn_factors:
                    lr = f32(lr * f32(learn_factors[k]))
                    k += 1
                l.learn_rate = lr
                l.momentum = f32(momentum)
                l.l2_penalty = f32(l2_penalty)
                l.l1_penalty = f32(l1_penalty)
                l.l2_upper_bound = f32(l2_upper_bound)
                l.average_grad = bool(average_grad)

    def propagate(self, x):
        """nnet-nnet.cc:26-48: keeps every layer's output for backprop."""
        bufs = [np.array(x, f32, copy=True)]
        for l in self.layers:
            bufs.append(l.propagate(bufs[-1]))
        self.propagate_buf = bufs
        return bufs[-1].copy()

    def feedforward(self, x):
        """nnet-nnet.cc:120-145 (same arithmetic, no buffers kept)."""
        y = np.asarray(x, f32)
        for l in self.layers:
            y = l.propagate(y)
        return y

    def backpropagate(self, in_err, want_out_err=False):
        """nnet-nnet.cc:50-118: per layer, top down: Backpropagate with the OLD weights, then
        Update; the input-gradient of layer 0 is skipped unless asked for."""
        L = len(self.layers)
        buf = self.propagate_buf
        err = np.asarray(in_err, f32)
        bbuf = [None] * max(L - 1, 0)
        out_err = None
        i = L - 1
        if i > 0:
            cur = err
            for i in range(L - 1, 0, -1):
                l = self.layers[i]
                bbuf[i - 1] = l.backpropagate(cur, buf[i], buf[i + 1])
                if l.updatable and l.learn_rate > 0:
                    l.update(buf[i], cur)
                cur = bbuf[i - 1]
            l0 = self.layers[0]
            if want_out_err:
                out_err = l0.backpropagate(bbuf[0], buf[0], buf[1])
            if l0.updatable and l0.learn_rate > 0:
                l0.update(buf[0], bbuf[0])
        else:
            l0 = self.layers[0]
            if want_out_err:
                out_err = l0.backpropagate(err, buf[0], buf[1])
            if l0.updatable and l0.learn_rate > 0:
                l0.update(buf[0], err)
        self.backpropagate_buf = bbuf
        return out_err


class Xent:
    """nnet/nnet-loss.{h,cc}: EvalVec :59-94, Report :97-105."""

    def __init__(self):
        self.frames = 0
        self.correct = 0
        self.loss = 0.0  # double

    def eval_vec(self, net_out, target):
        target = np.asarray(target, np.int64)
        max_id = find_row_max_id(net_out)
        correct = int((max_id == target).sum())
        diff, log_post = diff_xent(target, net_out)
        # Vector<float>::Sum() then `loss_ -=` in double
        self.loss -= float(np.asarray(log_post, f32).astype(np.float64).sum())
        self.frames += net_out.shape[0]
        self.correct += correct
        return diff, log_post, max_id, correct

    def report(self):
        s = "Xent:%s frames:%d err/frm:%s" % (_fmt(self.loss), self.frames, _fmt(self.loss / self.frames))
        if self.correct >= 0:
            s += "\nFRAME_ACCURACY >> %s%% <<" % _fmt(100.0 * self.correct / self.frames)
        return s


def _fmt(v):
    """ostream << double with the default precision (6 significant digits)."""
    return "%g" % v


def mse_eval(net_out, target):
    """nnet/nnet-loss.cc:160-182: diff, per-row sum of squares (AddColSumMat), 0.5*sum."""
    diff = (np.asarray(net_out, f32) - np.asarray(target, f32)).astype(f32)
    sq = (diff * diff).astype(f32)
    row = add_col_sum_mat(1.0, sq, 0.0, np.zeros(sq.shape[0], f32))
    return diff, row, 0.5 * float(row.astype(np.float64).sum())


# ------------------------------------------------------------------------------------------------
# Cache (frame shuffling), nnet/nnet-cache.{h,cc}
# ------------------------------------------------------------------------------------------------
class Cache:
    EMPTY, FILLING, FULL, EMPTYING = range(4)

    def __init__(self, cachesize, bunchsize):
        """:28-47"""
        assert cachesize > 0 and bunchsize > 0 and cachesize >= bunchsize and cachesize % bunchsize == 0
        self.cachesize, self.bunchsize = cachesize, bunchsize
        self.state = Cache.EMPTY
        self.filling_pos = 0
        self.emptying_pos = 0
        self.randomized = False
        self.features = None
        self.targets = None
        self.features_leftover = np.zeros((0, 0), f32)
        self.targets_leftover = np.zeros(0, np.int32)
        self.features_random = None
        self.targets_random = None

    def add_data(self, feats, targets):
        """:51-136"""
        assert self.state != Cache.FULL
        feats = np.asarray(feats, f32)
        targets = np.asarray(targets, np.int32)
        if self.features is None or self.features.shape[0] != self.cachesize:
            self.features = np.zeros((self.cachesize, feats.shape[1]), f32)
            self.targets = np.zeros(self.cachesize, np.int32)
        if self.state == Cache.EMPTY:
            self.state = Cache.FILLING
            self.filling_pos = 0
            leftover = min(self.features_leftover.shape[0], self.cachesize)
            if leftover > 0:
                self.features[:leftover] = self.features_leftover[:leftover]
                self.targets[:leftover] = self.targets_leftover[:leftover]
                self.features_leftover = np.zeros((0, 0), f32)
                self.targets_leftover = np.zeros(0, np.int32)
                self.filling_pos += leftover
        space = self.cachesize - self.filling_pos
        n = feats.shape[0]
        fill = min(space, n)
        self.features[self.filling_pos:self.filling_pos + fill] = feats[:fill]
        self.targets[self.filling_pos:self.filling_pos + fill] = targets[:fill]
        if n - fill > 0:
            self.features_leftover = feats[fill:].copy()
            self.targets_leftover = targets[fill:].copy()
        self.filling_pos += fill
        if self.filling_pos == self.cachesize:
            self.state = Cache.FULL

    def randomize(self, randmask):
        """:140-163 with the permutation injected (the reference draws it with
        std::random_shuffle over rand(), :149-151)."""
        randmask = np.asarray(randmask)
        assert randmask.size == self.filling_pos
        self.features_random = np.zeros_like(self.features)
        self.targets_random = np.zeros_like(self.targets)
        self.features_random[:self.filling_pos] = self.features[randmask]
        self.targets_random[:self.filling_pos] = self.targets[randmask]
        self.randomized = True

    def full(self):
        return self.state == Cache.FULL

    def empty(self):
        """nnet-cache.h:56-58"""
        return self.state == Cache.EMPTY or self.filling_pos < self.bunchsize

    def get_bunch(self):
        """:167-209 -- note the final partial bunch is never handed out."""
        assert self.state != Cache.EMPTY
        if self.state in (Cache.FULL, Cache.FILLING):
            self.state = Cache.EMPTYING
            self.emptying_pos = 0
        a, b = self.emptying_pos, self.emptying_pos + self.bunchsize
        if self.randomized:
            f, t = self.features_random[a:b].copy(), self.targets_random[a:b].copy()
        else:
            f, t = self.features[a:b].copy(), self.targets[a:b].copy()
        self.emptying_pos += self.bunchsize
        if self.emptying_pos > self.filling_pos - self.bunchsize:
            self.state = Cache.EMPTY
        return f, t


# ------------------------------------------------------------------------------------------------
# RBM, nnet/nnet-rbm.h:65-333
# ------------------------------------------------------------------------------------------------
class Rbm:
    marker = "<rbm>"
    updatable = True

    def __init__(self, vis_hid, vis_bias, hid_bias, vis_type="bern", hid_type="bern"):
        self.W = np.array(vis_hid, f32)  # (hid x vis)
        self.vis_bias = np.array(vis_bias, f32)
        self.hid_bias = np.array(hid_bias, f32)
        self.vis_type, self.hid_type = vis_type, hid_type
        self.W_corr = np.zeros_like(self.W)
        self.vis_bias_corr = np.zeros_like(self.vis_bias)
        self.hid_bias_corr = np.zeros_like(self.hid_bias)
        self.learn_rate = f32(0)
        self.momentum = f32(0)
        self.l2_penalty = f32(0)

    @property
    def dim_in(self):
        return self.W.shape[1]

    @property
    def dim_out(self):
        return self.W.shape[0]

    def propagate(self, vis):
        """:136-145"""
        out = np.broadcast_to(self.hid_bias, (vis.shape[0], self.hid_bias.size)).astype(f32)
        out = gemm(1.0, vis, False, self.W, True, 1.0, out)
        return sigmoid(out) if self.hid_type == "bern" else out

    def reconstruct(self, hid):
        """:158-176"""
        out = np.broadcast_to(self.vis_bias, (hid.shape[0], self.vis_bias.size)).astype(f32)
        out = gemm(1.0, hid, False, self.W, False, 1.0, out)
        return sigmoid(out) if self.vis_type == "bern" else out

    def rbm_update(self, pos_vis, pos_hid, neg_vis, neg_hid):
        """:178-223"""
        N = f32(pos_vis.shape[0])
        lr, mom = self.learn_rate, self.momentum
        g = f32(lr / N)
        self.W_corr = gemm(-g, neg_hid, True, neg_vis, False, mom, self.W_corr)
        self.W_corr = gemm(g, pos_hid, True, pos_vis, False, 1.0, self.W_corr)
        self.W_corr = (f32(-lr * self.l2_penalty) * self.W + self.W_corr).astype(f32)
        self.W = (self.W_corr + self.W).astype(f32)
        self.vis_bias_corr = add_row_sum_mat(-g, neg_vis, mom, self.vis_bias_corr)
        self.vis_bias_corr = add_row_sum_mat(g, pos_vis, 1.0, self.vis_bias_corr)
        self.vis_bias = (self.vis_bias_corr + self.vis_bias).astype(f32)
        self.hid_bias_corr = add_row_sum_mat(-g, neg_hid, mom, self.hid_bias_corr)
        self.hid_bias_corr = add_row_sum_mat(g, pos_hid, 1.0, self.hid_bias_corr)
        self.hid_bias = (self.hid_bias_corr + self.hid_bias).astype(f32)


def rbm_as_nnet(rbm):
    """Rbm::WriteAsNnet, nnet/nnet-rbm.h:233-249: the hidden side as <biasedlinearity>, followed by
    <sigmoid> when the hidden units are Bernoulli."""
    layers = [BiasedLinearity(rbm.W.copy(), rbm.hid_bias.copy())]
    if rbm.hid_type == "bern":
        layers.append(Sigmoid(rbm.W.shape[0]))
    return layers


def write_rbm_as_nnet(rbm, binary):
    """rbm-convert-to-nnet's output file: in text mode WriteAsNnet ends with a newline of its own,
    which follows the bias vector's when there is no <sigmoid> line (nnet-rbm.h:247)."""
    out = write_nnet(rbm_as_nnet(rbm), binary)
    if not binary and rbm.hid_type != "bern":
        out += b"\n"
    return out


def binarize_probs(probs, uniform):
    """cudamatrix/cu-randkernels.cu:113-120 / cu-rand-inl.h:163-167 with the uniform draws injected:
    state = probs > rand ? 1 : 0."""
    return (np.asarray(probs, f32) > np.asarray(uniform, f32)).astype(f32)


def rbm_cd1_step(rbm, pos_vis, uniform=None, gauss=None):
    """nnetbin/rbm-train-cd1-frmshuff.cc:140-160 with the random matrix injected."""
    pos_vis = np.asarray(pos_vis, f32)
    pos_hid = rbm.propagate(pos_vis)
    if rbm.hid_type == "bern":
        neg_hid = binarize_probs(pos_hid, uniform)
    else:
        neg_hid = (f32(1.0) * np.asarray(gauss, f32) + pos_hid).astype(f32)  # AddGaussNoise, cu-rand-inl.h:173-177
    neg_vis = rbm.reconstruct(neg_hid)
    neg_hid2 = rbm.propagate(neg_vis)
    rbm.rbm_update(pos_vis, pos_hid, neg_vis, neg_hid2)
    _, _, mse = mse_eval(neg_vis, pos_vis)
    return dict(pos_hid=pos_hid, neg_hid_sample=neg_hid, neg_vis=neg_vis, neg_hid=neg_hid2, mse=mse)


# ------------------------------------------------------------------------------------------------
# HybridTaus generator, cudamatrix/cu-randkernels.cu:44-120 (uint32 arithmetic, bit-exact)
# ------------------------------------------------------------------------------------------------
def _taus(z, s1, s2, s3, m):
    b = (((z << np.uint32(s1)) ^ z) >> np.uint32(s2)).astype(np.uint32)
    return (((z & np.uint32(m)) << np.uint32(s3)) ^ b).astype(np.uint32)


def hybrid_taus(z1, z2, z3, z4):
    """One uniform draw per element; returns (values float32, new states).  Redraws elements whose
    value rounds to 0 or 1 (the do/while of :62-69)."""
    z = [np.array(a, np.uint32, copy=True) for a in (z1, z2, z3, z4)]
    out = np.zeros(z[0].shape, f32)
    todo = np.ones(z[0].shape, bool)
    with np.errstate(over="ignore"):
        while todo.any():
            n1 = _taus(z[0], 13, 19, 12, 4294967294)
            n2 = _taus(z[1], 2, 25, 4, 4294967288)
            n3 = _taus(z[2], 3, 11, 17, 4294967280)
            n4 = (np.uint32(1664525) * z[3] + np.uint32(1013904223)).astype(np.uint32)
            for zi, ni in zip(z, (n1, n2, n3, n4)):
                zi[todo] = ni[todo]
            bits = (z[0] ^ z[1] ^ z[2] ^ z[3]).astype(np.uint32)
            v = (2.3283064365387e-10 * bits.astype(np.float64)).astype(f32)
            out[todo] = v[todo]
            todo = todo & ~((out > 0) & (out < 1))
    return out, z


def box_muller(z1, z2, z3, z4):
    """cu-randkernels.cu:86-94."""
    u0, z = hybrid_taus(z1, z2, z3, z4)
    u1, z = hybrid_taus(*z)
    r = np.sqrt(-2.0 * np.log(u0).astype(f32).astype(np.float64)).astype(f32)
    theta = (f32(6.283185307179586476925286766558) * u1).astype(f32)
    return (r * np.sin(theta).astype(f32)).astype(f32), z


# ------------------------------------------------------------------------------------------------
# libc rand() based helpers of base/kaldi-math.{h,cc}; used for seeds and the cache permutation
# ------------------------------------------------------------------------------------------------
_libc = None


def _rand():
    global _libc
    if _libc is None:
        _libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6")
        _libc.rand.restype = ctypes.c_int
    return _libc.rand()


RAND_MAX = 2147483647


def rand_int(lo, hi):
    """base/kaldi-math.cc:36-61 RandInt(min,max) (non-MSVC branch): min + rand() % (max+1-min)."""
    return lo + (_rand() % (hi + 1 - lo))


def srand(seed):
    global _libc
    _rand()
    _libc.srand(ctypes.c_uint(seed))


# ------------------------------------------------------------------------------------------------
# Model / table I/O: base/io-funcs*.{h,cc}, matrix/kaldi-matrix.cc:1038-1265,
# matrix/kaldi-vector.cc:846-983, nnet/nnet-component.cc:72-167, util/kaldi-holder-inl.h
# ------------------------------------------------------------------------------------------------
def _w_token(tok):
    return tok.encode() + b" "


def _w_i32(v, binary):
    return (b"\x04" + struct.pack("<i", v)) if binary else (b"%d " % v)


def write_matrix(m, binary):
    m = np.asarray(m, f32)
    if binary:
        return b"FM " + _w_i32(m.shape[0], True) + _w_i32(m.shape[1], True) + m.astype("<f4").tobytes()
    if m.shape[1] == 0:
        return b" [ ]\n"
    s = " ["
    for r in m:
        s += "\n  " + "".join(_cxx_float(v) + " " for v in r)
    return (s + "]\n").encode()


def write_vector(v, binary, double=False):
    if binary:
        if double:
            return b"DV " + _w_i32(len(v), True) + np.asarray(v, "<f8").tobytes()
        return b"FV " + _w_i32(len(v), True) + np.asarray(v, "<f4").tobytes()
    return (" [ " + "".join(_cxx_float(x) + " " for x in v) + "]\n").encode()


def _cxx_float(v):
    """ostream << float/double at precision 7 (InitKaldiOutputStream raises it to 7)."""
    return "%.7g" % float(v)


def write_nnet(layers, binary):
    """Nnet::Write -> Component::Write (nnet-component.cc:160-167); file-level "\\0B" from Output."""
    out = b"\0B" if binary else b""
    for l in layers:
        out += _w_token(l.marker) + _w_i32(l.dim_out, binary) + _w_i32(l.dim_in, binary)
        if not binary:
            out += b"\n"
        if isinstance(l, BiasedLinearity):
            out += write_matrix(l.W, binary) + write_vector(l.b, binary)
        elif isinstance(l, Rbm):
            out += _w_token(l.vis_type) + _w_token(l.hid_type)
            out += write_matrix(l.W, binary) + write_vector(l.vis_bias, binary) + write_vector(l.hid_bias, binary)
        elif isinstance(l, Expand):
            out += write_vector(np.asarray(l.offsets, np.float64), binary, double=True)
        elif isinstance(l, Copy):
            out += write_vector(np.asarray(l.indices, np.float64) + 1.0, binary, double=True)
        elif isinstance(l, (Scale, Dropout)):
            v = l.scale if isinstance(l, Scale) else l.ratio
            # WriteBasicType<float> (io-funcs-inl.h:30-48): binary = size byte + 4 bytes, text = value + space
            out += (b"\x04" + np.float32(v).tobytes()) if binary else (_cxx_float(v) + " \n").encode()
    return out


class _Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def peek(self):
        return self.d[self.p] if self.p < len(self.d) else -1

    def skip_ws(self):
        while self.p < len(self.d) and chr(self.d[self.p]).isspace():
            self.p += 1

    def token(self, binary):
        if not binary:
            self.skip_ws()
        a = self.p
        while self.p < len(self.d) and not chr(self.d[self.p]).isspace():
            self.p += 1
        t = self.d[a:self.p].decode()
        self.p += 1  # the separating space
        return t

    def i32(self, binary):
        if binary:
            assert self.d[self.p] == 4, "expected int32 size byte"
            v = struct.unpack_from("<i", self.d, self.p + 1)[0]
            self.p += 5
            return v
        return int(self.token(False))

    def f32(self, binary):
        """ReadBasicType<float> (io-funcs.cc)"""
        if binary:
            assert self.d[self.p] == 4, "expected float size byte"
            v = struct.unpack_from("<f", self.d, self.p + 1)[0]
            self.p += 5
            return v
        return float(self.token(False))

    def matrix(self, binary):
        if binary:
            tok = self.token(True)
            assert tok in ("FM", "DM"), tok
            r, c = self.i32(True), self.i32(True)
            dt = "<f4" if tok == "FM" else "<f8"
            n = r * c * int(dt[-1])
            m = np.frombuffer(self.d, dt, r * c, self.p).reshape(r, c).astype(f32)
            self.p += n
            return m
        self.skip_ws()
        assert self.token(False) == "["
        rows, cur = [], []
        while True:
            self.skip_ws_no_nl()
            ch = chr(self.peek())
            if ch == "]":
                self.p += 1
                if self.peek() == 10:
                    self.p += 1
                if cur:
                    rows.append(cur)
                break
            if ch in "\n;":
                self.p += 1
                if cur:
                    rows.append(cur)
                    cur = []
                continue
            cur.append(float(self.token(True)))
            self.p -= 1  # token() ate the separator; it may be a newline that ends the row
        return np.array(rows, f32).reshape(len(rows), len(rows[0]) if rows else 0)

    def skip_ws_no_nl(self):
        while self.p < len(self.d) and self.d[self.p] in (32, 9, 13):
            self.p += 1

    def vector(self, binary):
        if binary:
            tok = self.token(True)
            assert tok in ("FV", "DV"), tok
            n = self.i32(True)
            dt = "<f4" if tok == "FV" else "<f8"
            v = np.frombuffer(self.d, dt, n, self.p).copy()
            self.p += n * int(dt[-1])
            return v
        self.skip_ws()
        assert self.token(False) == "["
        vals = []
        while True:
            self.skip_ws_no_nl()
            if chr(self.peek()) == "]":
                self.p += 1
                if self.peek() == 10:
                    self.p += 1
                break
            vals.append(float(self.token(True)))
            self.p -= 1
        return np.array(vals, np.float64)


def read_nnet(data):
    """Nnet::Read (nnet-nnet.cc:147-163) + Component::Read (nnet-component.cc:72-158)."""
    rd = _Reader(data)
    binary = data[:2] == b"\0B"
    if binary:
        rd.p = 2
    layers = []
    while True:
        if not binary:
            rd.skip_ws()
        if rd.peek() == -1:
            break
        marker = rd.token(binary)
        dim_out, dim_in = rd.i32(binary), rd.i32(binary)
        if marker == "<biasedlinearity>":
            W = rd.matrix(binary)
            b = rd.vector(binary).astype(f32)
            assert W.shape == (dim_out, dim_in) and b.size == dim_out
            layers.append(BiasedLinearity(W, b))
        elif marker == "<sigmoid>":
            layers.append(Sigmoid(dim_out))
        elif marker == "<softmax>":
            layers.append(Softmax(dim_out))
        elif marker == "<rbm>":
            vt, ht = rd.token(binary), rd.token(binary)
            W = rd.matrix(binary)
            vb = rd.vector(binary).astype(f32)
            hb = rd.vector(binary).astype(f32)
            layers.append(Rbm(W, vb, hb, vt, ht))
        elif marker == "<expand>":
            layers.append(Expand(dim_in, dim_out, rd.vector(binary)))
        elif marker == "<copy>":
            layers.append(Copy(dim_in, dim_out, rd.vector(binary) - 1.0))
        elif marker == "<relu>":
            layers.append(Relu(dim_out))
        elif marker == "<softrelu>":
            layers.append(SoftRelu(dim_out))
        elif marker in (",<scale>", "<dropout>"):
            v = rd.f32(binary)
            layers.append(Scale(dim_out, v) if marker == ",<scale>" else Dropout(dim_out, v))
        else:
            raise ValueError("Unknown marker" + marker)
    return layers


def write_matrix_ark(pairs, binary=True):
    """util/kaldi-holder-inl.h:43-53 (KaldiObjectHolder<Matrix>): key, space, [\\0B] matrix."""
    out = b""
    for key, m in pairs:
        out += key.encode() + b" " + (b"\0B" if binary else b"") + write_matrix(m, binary)
    return out


def write_int32_vector_ark(pairs, binary=True):
    """util/kaldi-holder-inl.h:195-222 (BasicVectorHolder<int32>)."""
    out = b""
    for key, v in pairs:
        out += key.encode() + b" "
        if binary:
            out += b"\0B" + _w_i32(len(v), True) + b"".join(_w_i32(int(x), True) for x in v)
        else:
            out += "".join("%d " % int(x) for x in v).encode() + b"\n"
    return out


def read_matrix_ark(data):
    """Sequential read of a Matrix<float> archive (binary or text entries)."""
    rd = _Reader(data)
    out = []
    while rd.p < len(data):
        key = rd.token(True)
        if not key:
            break
        binary = data[rd.p:rd.p + 2] == b"\0B"
        if binary:
            rd.p += 2
        out.append((key, rd.matrix(binary)))
    return out
