"""CPU-side checks of the shipped libraries: they load without a GPU, export every symbol the
public headers declare, and the host random stream is glibc's.  No compute call is made."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

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


def test_public_header_is_plain_c_and_struct_layouts_match_the_binding(tmp_path):
    """include/kaldi_hip.h compiles as C (the boundary a foreign-function binding sees), and the structs that cross
    it by value or by array -- MatrixDim, khAffineUpdate, khAffineGradient, khSgdSegment -- have the size and field offsets the ctypes mirror in
    kaldi_amd/hip.py assumes."""
    import kaldi_amd.hip as h
    cname = {"in_": "in"}
    structs = [("khAffineUpdate", h.AffineUpdate), ("khAffineGradient", h.AffineGradient), ("khSgdSegment", h.SgdSegment),
               ("khAffineForward", h.AffineForward)]
    lines = ['  printf("%zu\\n", sizeof(MatrixDim));\n']
    for cn, py in structs:
        lines.append('  printf("%%zu\\n", sizeof(%s));\n' % cn)
        lines += ['  printf("%%zu\\n", offsetof(%s, %s));\n' % (cn, cname.get(f, f)) for f, _ in py._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "kaldi_hip.h"\nint main(void) {\n' + "".join(lines) +
                   "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out.pop(0) == ctypes.sizeof(h.MatrixDim)
    for cn, py in structs:
        assert out.pop(0) == ctypes.sizeof(py), cn
        for f, _ in py._fields_:
            assert out.pop(0) == getattr(py, f).offset, (cn, f)


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


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC + "/nnetbin"), reason="the reference tree is only present in the build container")
def test_reference_sources_compile_against_these_headers(tmp_path):
    """Source compatibility of the class API (CuMatrix / CuVector / CuStlVector / cu:: / Nnet / Component / Xent /
    Cache / Rbm): the reference's own callers are syntax-checked (g++ -fsyntax-only) against the headers of
    kaldi_amd/csrc.  Two groups:
      clients of the nnet classes -- the three tools and the reference's nnet-nnet.cc -- see only this
        repository's headers;
      the reference's Xent / Mse and Cache, whose .cc files implement their own headers' private members, are
        checked together with those two headers against this repository's cudamatrix / base / util layer."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = ["-I" + os.path.join(root, "kaldi_amd", "csrc"), "-I" + os.path.join(root, "include")]
    groups = {
        "clients": ["nnet/nnet-nnet.cc", "nnetbin/nnet-forward.cc", "nnetbin/nnet-train-xent-hardlab-frmshuff.cc",
                    "nnetbin/rbm-train-cd1-frmshuff.cc", "nnetbin/nnet-train-xent-hardlab-perutt.cc",
                    "nnetbin/nnet-train-mse-tgtmat-frmshuff.cc", "nnetbin/nnet-copy.cc", "nnetbin/rbm-convert-to-nnet.cc"],
        "with_own_headers": ["nnet/nnet-loss.cc", "nnet/nnet-loss.h", "nnet/nnet-cache.cc", "nnet/nnet-cache.h"],
    }
    failures = []
    for gname, files in groups.items():
        d = tmp_path / gname
        for f in files:
            (d / os.path.dirname(f)).mkdir(parents=True, exist_ok=True)
            (d / f).write_bytes(open(os.path.join(REF_SRC, f), "rb").read())   # a temporary copy, never committed
        for f in files:
            if not f.endswith(".cc"):
                continue
            r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + str(d)] + inc + [str(d / f)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            if r.returncode != 0:
                failures.append("%s: %s" % (f, r.stderr.decode()[:1500]))
    assert not failures, "\n".join(failures)


@pytest.mark.skipif(not os.path.isdir("/root/reference/egs"), reason="the reference tree is only present in the build container")
def test_scheduler_contract_fixture_matches_the_reference_script():
    """tests/golden/scheduler_contract.json -- the option names and the awk program that
    tests/test_gpu_nnet.py::test_recipe_scheduler_contract drives the trainer with -- is what
    egs/aurora4/a00/steps/nnet_scheduler/train_scheduler.sh:76-120 of the reference holds."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_scheduler_contract", os.path.join(ROOT, "tests", "golden", "make_scheduler_contract.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = mod.extract("/root/reference")
    got = json.load(open(os.path.join(ROOT, "tests", "golden", "scheduler_contract.json")))
    assert got == want
    # every option the script passes is one the trainer registers
    exe = os.path.join(ROOT, "kaldi_amd", "bin", "nnet-train-xent-hardlab-frmshuff")
    usage = subprocess.run([exe, "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).stderr.decode()
    for o in set(got["train_options"]) | set(got["crossval_options"]):
        assert ("--" + o) in usage, o


def test_damaged_tables_are_read_errors_not_memory_errors():
    """The table readers of the tools (float matrices and vectors, integer vectors; sequential, random access, script
    files), built with the address and undefined-behaviour sanitizers, on archives with flipped, cut, inserted and
    overwritten bytes (tests/fuzz/table_mutation.py): every run reads to the end or reports an error."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "table_mutation.py"), "--cases", "250", "--seed", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "250 damaged tables, 0 findings" in r.stdout
