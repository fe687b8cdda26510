"""ctypes binding of libkaldi_nnet.so: the host C++ layer (Nnet, Xent, Cache, Rbm ...) behind the
small C API of kaldi_amd/csrc/capi/nnet1_capi.h.  Glue for tests and the benchmark."""
import ctypes
import os

import numpy as np

from . import hip

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libkaldi_nnet.so")
BIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")
if not os.path.exists(LIB_PATH):
    raise ImportError("kaldi_amd: %s is missing; build it first (__graft_entry__.build())" % LIB_PATH)
lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIGS = {
    "nnet1_last_error": (ctypes.c_char_p, []),
    "nnet1_select_gpu": (I, [I]),
    "nnet1_create": (P, [ctypes.c_char_p]),
    "nnet1_destroy": (None, [P]),
    "nnet1_num_layers": (I, [P]),
    "nnet1_input_dim": (I, [P]),
    "nnet1_output_dim": (I, [P]),
    "nnet1_set_train_opts": (I, [P, F, F, F, F, F, I, I, ctypes.c_char_p]),
    "nnet1_set_data_parallel": (I, [P, I]),
    "nnet1_train_bunch": (I, [P, P, I, I, I, P, I]),
    "nnet1_feedforward": (I, [P, P, I, I, I, P, I]),
    "nnet1_keep_posteriors": (I, [P, I]),
    "nnet1_fusion_enabled": (I, [P]),
    "nnet1_get_stats": (I, [P, P, P, P]),
    "nnet1_reset_stats": (I, [P]),
    "nnet1_write": (I, [P, ctypes.c_char_p, I]),
    "nnet1_get_posteriors": (I, [P, P, I, I]),
    "nnet1_get_layer_params": (I, [P, I, P, P]),
    "rbm1_create": (P, [ctypes.c_char_p]),
    "rbm1_destroy": (None, [P]),
    "rbm1_set_train_opts": (I, [P, F, F, F]),
    "rbm1_cd1_bunch": (I, [P, P, I, I, I]),
    "rbm1_write": (I, [P, ctypes.c_char_p, I]),
    "rbm1_dims": (I, [P, P, P]),
    "nnet1_synchronize": (I, []),
    "nnet1_shard_assign": (I, [I, I, P, P]),
    "nnet1_substitute_job": (I, [ctypes.c_char_p, I, ctypes.c_char_p, I]),
    "nnet1_table_scan_shard": (I, [ctypes.c_char_p, I, I, P, P, P]),
    "nnet1_comm_nonce": (I, [ctypes.c_long, ctypes.c_char_p, I]),
    "nnet1_comm_id_file_write": (I, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, I]),
    "nnet1_comm_id_file_read": (I, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, I]),
}
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


class NnetError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise NnetError(lib.nnet1_last_error().decode())


class Trainer:
    """Nnet + Xent, i.e. the state of nnet-train-xent-hardlab-frmshuff's inner loop."""

    def __init__(self, model_path):
        self.h = lib.nnet1_create(model_path.encode())
        if not self.h:
            raise NnetError(lib.nnet1_last_error().decode())
        self.input_dim = lib.nnet1_input_dim(self.h)
        self.output_dim = lib.nnet1_output_dim(self.h)
        self.num_layers = lib.nnet1_num_layers(self.h)

    def set_train_opts(self, learn_rate, momentum=0.0, l2_penalty=0.0, l1_penalty=0.0, l2_upper_bound=0.0,
                       average_grad=False, fused=True, learn_factors=""):
        _check(lib.nnet1_set_train_opts(self.h, learn_rate, momentum, l2_penalty, l1_penalty, l2_upper_bound,
                                        int(average_grad), int(fused), learn_factors.encode()))

    def set_data_parallel(self, nranks):
        _check(lib.nnet1_set_data_parallel(self.h, nranks))

    def train_bunch(self, feats_ptr, ld, rows, cols, labels_ptr, cross_validate=False):
        _check(lib.nnet1_train_bunch(self.h, feats_ptr, ld, rows, cols, labels_ptr, int(cross_validate)))

    def keep_posteriors(self, keep):
        """False: training steps do not write the posteriors (nothing reads them in a training loop)"""
        _check(lib.nnet1_keep_posteriors(self.h, int(keep)))

    def fusion_enabled(self):
        rc = lib.nnet1_fusion_enabled(self.h)
        if rc < 0:
            raise NnetError(lib.nnet1_last_error().decode())
        return rc == 1

    def feedforward(self, feats, out):
        """feats, out: hip.DevMatrix"""
        _check(lib.nnet1_feedforward(self.h, feats.ptr, feats.stride, feats.rows, feats.cols, out.ptr, out.stride))

    def stats(self):
        fr, co, lo = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_double()
        _check(lib.nnet1_get_stats(self.h, ctypes.byref(fr), ctypes.byref(co), ctypes.byref(lo)))
        return fr.value, co.value, lo.value

    def reset_stats(self):
        _check(lib.nnet1_reset_stats(self.h))

    def posteriors(self, rows):
        out = np.empty((rows, self.output_dim), np.float32)
        _check(lib.nnet1_get_posteriors(self.h, out.ctypes.data_as(P), rows, self.output_dim))
        return out

    def layer_params(self, layer, dim_out, dim_in):
        W = np.empty((dim_out, dim_in), np.float32)
        b = np.empty(dim_out, np.float32)
        _check(lib.nnet1_get_layer_params(self.h, layer, W.ctypes.data_as(P), b.ctypes.data_as(P)))
        return W, b

    def write(self, path, binary=True):
        _check(lib.nnet1_write(self.h, path.encode(), int(binary)))

    def close(self):
        if self.h:
            lib.nnet1_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RbmTrainer:
    """Rbm + CuRand + MseProgress, i.e. the state of rbm-train-cd1-frmshuff's inner loop."""

    def __init__(self, model_path):
        self.h = lib.rbm1_create(model_path.encode())
        if not self.h:
            raise NnetError(lib.nnet1_last_error().decode())
        v, hd = I(), I()
        _check(lib.rbm1_dims(self.h, ctypes.byref(v), ctypes.byref(hd)))
        self.vis, self.hid = v.value, hd.value

    def set_train_opts(self, learn_rate, momentum=0.0, l2_penalty=0.0):
        _check(lib.rbm1_set_train_opts(self.h, learn_rate, momentum, l2_penalty))

    def cd1_bunch(self, feats_ptr, ld, rows, cols):
        _check(lib.rbm1_cd1_bunch(self.h, feats_ptr, ld, rows, cols))

    def write(self, path, binary=True):
        _check(lib.rbm1_write(self.h, path.encode(), int(binary)))

    def close(self):
        if self.h:
            lib.rbm1_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_assign(num_ranks, num_frames):
    """Rank of every utterance under the sharded nnet-forward's rule (host only, no GPU)."""
    nf = np.ascontiguousarray(num_frames, dtype=np.int64)
    out = np.zeros(nf.size, np.int32)
    _check(lib.nnet1_shard_assign(num_ranks, nf.size, nf.ctypes.data, out.ctypes.data))
    return out


def substitute_job(spec, rank):
    buf = ctypes.create_string_buffer(len(spec) + 32)
    _check(lib.nnet1_substitute_job(spec.encode(), rank, buf, len(buf)))
    return buf.value.decode()


def table_scan_shard(rspecifier, num_ranks, rank):
    """(entries kept, frames kept, payload bytes read) of rank `rank` walking a float-matrix table the way the
    sharded nnet-forward does; raises NnetError on a read error (host only)."""
    k, f, b = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
    _check(lib.nnet1_table_scan_shard(rspecifier.encode(), num_ranks, rank, ctypes.byref(k), ctypes.byref(f), ctypes.byref(b)))
    return k.value, f.value, b.value


def comm_nonce(parent_pid):
    """The launch nonce of the RCCL id file as a rank with this parent process computes it (host only)."""
    buf = ctypes.create_string_buffer(512)
    _check(lib.nnet1_comm_nonce(parent_pid, buf, len(buf)))
    return buf.value.decode()


def comm_id_file_write(path, nonce, ident):
    _check(lib.nnet1_comm_id_file_write(path.encode(), nonce.encode(), bytes(ident), len(ident)))


def comm_id_file_read(path, nonce, id_bytes):
    """The id if `path` is a complete id file of the launch `nonce`, else None."""
    buf = ctypes.create_string_buffer(id_bytes)
    rc = lib.nnet1_comm_id_file_read(path.encode(), nonce.encode(), buf, id_bytes)
    if rc < 0:
        raise NnetError("comm_id_file_read: bad arguments")
    return buf.raw if rc == 1 else None


def synchronize():
    _check(lib.nnet1_synchronize())
