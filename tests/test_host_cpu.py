"""CPU-side checks of the shipped libraries: they load without a GPU, export every symbol the
public headers declare, and the host random stream is glibc's.  No compute call is made."""
import ctypes
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_library_exports_every_declared_symbol():
    import kaldi_amd.hip as h
    assert len(h.PROTOS) >= 85
    for name in h.PROTOS:
        assert hasattr(h.lib, name), name
    # every cudaF_/cudaI32_ wrapper of the reference surface that the hot path uses has its hipF_/hipI32_ twin
    for op in ["set_const", "add_mat", "add_vec_to_rows", "sigmoid", "diff_sigmoid", "softmax", "softmax_part",
               "find_row_max_id", "diff_xent", "randomize", "expand", "copy", "add_row_sum_mat", "add_col_sum_mat",
               "div_rows_vec", "mul_cols_vec", "mul_elements", "apply_log", "rand", "gauss_rand", "binarize_probs",
               "regularize_l1", "apply_row_l2upperbound", "gemm"]:
        assert "hipF_" + op in h.PROTOS, op
    assert "hipI32_set_const" in h.PROTOS


def test_host_library_exports_capi():
    hdr = open(os.path.join(ROOT, "kaldi_amd", "csrc", "capi", "nnet1_capi.h")).read()
    names = re.findall(r"\b((?:nnet1|rbm1)_\w+)\s*\(", hdr)
    lib = ctypes.CDLL(os.path.join(ROOT, "kaldi_amd", "lib", "libkaldi_nnet.so"))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


def test_host_random_stream_is_glibc_rand():
    lib = ctypes.CDLL(os.path.join(ROOT, "kaldi_amd", "lib", "libkaldi_nnet.so"))
    libc = ctypes.CDLL("libc.so.6")
    libc.rand.restype = ctypes.c_int
    for seed in (1, 7, 123456789):
        out = (ctypes.c_int * 2000)()
        assert lib.nnet1_rand_stream(seed, 2000, out) == 0
        libc.srand(seed)
        assert list(out) == [libc.rand() for _ in range(2000)]


def test_tools_print_usage_without_gpu():
    for tool, n in [("nnet-forward", 3), ("nnet-train-xent-hardlab-frmshuff", 4), ("rbm-train-cd1-frmshuff", 3),
                    ("nnet-train-xent-hardlab-perutt", 4), ("nnet-train-mse-tgtmat-frmshuff", 4), ("nnet-copy", 2),
                    ("rbm-convert-to-nnet", 2)]:
        exe = os.path.join(ROOT, "kaldi_amd", "bin", tool)
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1
        assert ("Usage:  " + tool) in r.stderr.decode()
        r = subprocess.run([exe, "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and "--verbose" in r.stderr.decode() and "--print-args" in r.stderr.decode()
