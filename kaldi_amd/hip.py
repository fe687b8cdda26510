"""ctypes binding of libkaldi_hip.so (C ABI declared in include/kaldi_hip.h).

Signatures are read from the header itself, so every symbol the header declares is bound and a
missing export is an import-time error rather than a silent fallback.
"""
import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("KALDI_AMD_HIP_LIB") or os.path.join(_HERE, "lib", "libkaldi_hip.so")  # override: diagnostic builds
HEADER_PATH = os.path.join(ROOT, "include", "kaldi_hip.h")

NO_TRANS, TRANS = 111, 112
ACT_NONE, ACT_SIGMOID, ACT_SOFTMAX = 0, 1, 2
COMM_ID_BYTES = 128


class MatrixDim(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("stride", ctypes.c_int32)]


class AffineUpdate(ctypes.Structure):
    """khAffineUpdate: one item of hipF_affine_update_group (the arguments of hipF_affine_update_fused)."""
    _fields_ = [("N", ctypes.c_int32), ("O", ctypes.c_int32), ("I", ctypes.c_int32),
                ("e", ctypes.c_void_p), ("lde", ctypes.c_int32), ("in_", ctypes.c_void_p), ("ldin", ctypes.c_int32),
                ("gscale", ctypes.c_float), ("momentum", ctypes.c_float), ("lr", ctypes.c_float), ("l2", ctypes.c_float),
                ("W_in", ctypes.c_void_p), ("ldw_in", ctypes.c_int32), ("W_out", ctypes.c_void_p), ("ldw_out", ctypes.c_int32),
                ("corr", ctypes.c_void_p), ("ldcorr", ctypes.c_int32), ("bias", ctypes.c_void_p), ("bcorr", ctypes.c_void_p)]


class AffineGradient(ctypes.Structure):
    """khAffineGradient: one item of hipF_affine_gradient_group."""
    _fields_ = [("N", ctypes.c_int32), ("O", ctypes.c_int32), ("I", ctypes.c_int32),
                ("e", ctypes.c_void_p), ("lde", ctypes.c_int32), ("in_", ctypes.c_void_p), ("ldin", ctypes.c_int32),
                ("gW", ctypes.c_void_p), ("ldg", ctypes.c_int32), ("gb", ctypes.c_void_p)]


class AffineForward(ctypes.Structure):
    """khAffineForward: one layer of hipF_affine_forward_chain."""
    _fields_ = [("O", ctypes.c_int32), ("I", ctypes.c_int32), ("in_", ctypes.c_void_p), ("ldin", ctypes.c_int32),
                ("W", ctypes.c_void_p), ("ldw", ctypes.c_int32), ("bias", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("ldout", ctypes.c_int32)]


class SgdSegment(ctypes.Structure):
    """khSgdSegment: one run of elements of hipF_sgd_apply_segments with its hyper-parameters."""
    _fields_ = [("begin", ctypes.c_size_t), ("end", ctypes.c_size_t), ("gscale", ctypes.c_float),
                ("momentum", ctypes.c_float), ("lr", ctypes.c_float), ("l2", ctypes.c_float)]


_CTYPES = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
    "khStream_t": ctypes.c_void_p, "khEvent_t": ctypes.c_void_p, "MatrixDim": MatrixDim,
}


def _arg_type(decl):
    decl = decl.strip()
    if "*" in decl or "[" in decl:
        return ctypes.c_void_p
    if decl.startswith("double "):
        return ctypes.c_double
    toks = [t for t in re.split(r"\s+", decl) if t not in ("const", "unsigned")]
    base = toks[0]
    if base == "long":
        return ctypes.c_longlong
    return _CTYPES[base]


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"\b(int|const char \*)\s*(hip[A-Z0-9]+_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = [] if args in ("", "void") else [_arg_type(a) for a in args.split(",")]
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, argtypes)
    return protos


class HipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "kaldi_amd: %s is missing; the HIP device library must be built first "
        "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
PROTOS = parse_header()
for _name, (_res, _args) in PROTOS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def check(status):
    if status != 0:
        raise HipError("kaldi_hip error %d: %s" % (status, lib.hipK_error_string(status).decode()))


def call(name, *args):
    """Call an entry point by name and raise on a non-zero status."""
    check(getattr(lib, name)(*args))


def init(device=-1):
    call("hipK_init", device)


def synchronize():
    call("hipK_device_synchronize")


class Timer:
    """HIP-event timer on the library's default stream (or a given one)."""

    def __init__(self, stream=None):
        self.stream = stream
        self.e0, self.e1 = ctypes.c_void_p(), ctypes.c_void_p()
        call("hipK_event_create", ctypes.byref(self.e0))
        call("hipK_event_create", ctypes.byref(self.e1))

    def start(self):
        call("hipK_event_record", self.e0, self.stream)

    def stop(self):
        """-> elapsed milliseconds (blocks until the stop event has been reached)."""
        call("hipK_event_record", self.e1, self.stream)
        call("hipK_event_synchronize", self.e1)
        ms = ctypes.c_float()
        call("hipK_event_elapsed_ms", ctypes.byref(ms), self.e0, self.e1)
        return ms.value


class _OwnedPtr(ctypes.c_void_p):
    """A device address that keeps the DevMatrix it came from alive: `DevMatrix(...).ptr` passed straight
    into a call must not have its memory freed before the call is made."""


class DevMatrix:
    """A pitched float32 (or int32 / uint32 / double / int64) device buffer with host<->device copies.
    Owns its memory.  Test/bench plumbing only."""

    def __init__(self, rows, cols, dtype=np.float32, pitched=True, zero=True):
        self.rows, self.cols, self.dtype = int(rows), int(cols), np.dtype(dtype)
        self._p = ctypes.c_void_p()
        es = self.dtype.itemsize
        if pitched and es in (4, 8):
            st = ctypes.c_int32()
            call("hipK_malloc_pitch", ctypes.byref(self._p), ctypes.byref(st), self.rows, self.cols, es)
            self.stride = st.value
        else:
            self.stride = self.cols
            call("hipK_malloc", ctypes.byref(self._p), self.rows * self.cols * es)
        if zero and self.nbytes:
            call("hipK_memset_async", None, self.ptr, 0, self.nbytes)

    @property
    def ptr(self):
        p = _OwnedPtr(self._p.value)
        p._owner = self
        return p

    @property
    def nbytes(self):
        return self.rows * self.stride * self.dtype.itemsize

    @property
    def dim(self):
        return MatrixDim(self.rows, self.cols, self.stride)

    @classmethod
    def from_numpy(cls, a, pitched=True):
        a = np.ascontiguousarray(a)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        m = cls(a.shape[0], a.shape[1], a.dtype, pitched, zero=True)
        m.set(a)
        return m

    def set(self, a):
        a = np.ascontiguousarray(a, self.dtype).reshape(self.rows, self.cols)
        if a.size == 0:
            return
        es = self.dtype.itemsize
        call("hipK_memcpy_2d", None, self.ptr, self.stride * es, a.ctypes.data_as(ctypes.c_void_p), self.cols * es,
             self.cols * es, self.rows, 0)

    def numpy(self):
        out = np.empty((self.rows, self.cols), self.dtype)
        if out.size:
            es = self.dtype.itemsize
            synchronize()
            call("hipK_memcpy_2d", None, out.ctypes.data_as(ctypes.c_void_p), self.cols * es, self.ptr,
                 self.stride * es, self.cols * es, self.rows, 1)
        return out

    def free(self):
        if self._p:
            lib.hipK_free(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
