"""The CuMatrix / CuVector / cu:: classes against the reference's own (CPU branch): one random walk over the public
methods, built against each code base (tests/fuzz/cumatrix_walk.cc), must leave the same matrices behind."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))

pytestmark = pytest.mark.gpu


def test_class_api_walk_agrees_with_the_reference_classes(tmp_path):
    import cumatrix_walk as walk

    if not os.path.exists(walk.REF):
        pytest.skip("oracle/_ref/bin/cumatrix_walk not built (needs the reference sources at build time)")
    assert os.path.exists(walk.OURS), "kaldi_amd/bin/cumatrix_walk missing: make -C kaldi_amd/csrc host"
    counts, bad = {}, []
    for seed in (101, 102):
        fr, fo = str(tmp_path / "ref.bin"), str(tmp_path / "ours.bin")
        walk.run(walk.REF, seed, 800, fr)
        walk.run(walk.OURS, seed, 800, fo)
        ref, ours = walk.read_dump(fr), walk.read_dump(fo)
        walk.compare(seed, ref, ours, counts, bad)
    assert not bad, "\n".join(bad[:20])
    assert len(counts) == len(walk.OP_NAMES)  # every method group was reached


def test_partial_row_vector_reaching_the_last_column_is_refused_as_in_the_reference():
    import subprocess

    import cumatrix_walk as walk

    if not os.path.exists(walk.REF):
        pytest.skip("oracle/_ref/bin/cumatrix_walk not built")
    for binary in (walk.REF, walk.OURS):
        p = subprocess.run([binary, "refused"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert p.returncode != 0, binary
        assert b"Incorrect dimensions: Cols:8 VectorDim:3 offset:5" in p.stderr, (binary, p.stderr[-500:])


def test_network_api_walk_agrees_with_the_reference_classes(tmp_path):
    """Nnet / Component / Xent / Mse / MseProgress / Cache: tests/fuzz/nnet_walk.cc on random model files."""
    import numpy as np

    import nnet_walk as walk

    if not os.path.exists(walk.REF):
        pytest.skip("oracle/_ref/bin/nnet_walk not built (needs the reference sources at build time)")
    assert os.path.exists(walk.OURS), "kaldi_amd/bin/nnet_walk missing: make -C kaldi_amd/csrc host"
    rs = np.random.RandomState(7)
    counts, bad = {}, []
    for case in range(16):
        layers, desc = walk.random_model(rs)
        binary = bool(rs.rand() < 0.5)
        model = str(tmp_path / "m.nnet")
        open(model, "wb").write(walk.O.write_nnet(layers, binary))
        fr, fo = str(tmp_path / "ref.bin"), str(tmp_path / "ours.bin")
        rc_r, err_r = walk.run(walk.REF, model, 500 + case, 50, fr)
        rc_o, err_o = walk.run(walk.OURS, model, 500 + case, 50, fo)
        assert (rc_r == 0) == (rc_o == 0), (desc, rc_r, rc_o, err_r[-300:], err_o[-300:])
        if rc_r == 0:
            walk.compare("case %d (%s)" % (case, desc), walk.read_dump(fr), walk.read_dump(fo), bad, counts)
    assert not bad, "\n".join(bad[:20])
    for what in ("train_err", "propagate_buffer", "backpropagate_buffer", "cache_bunch", "write_text", "xent_report"):
        assert counts.get(what, 0) > 0, what


def test_rbm_api_walk_agrees_with_the_reference_classes(tmp_path):
    """Rbm: Propagate / Reconstruct / RbmUpdate / accessors / written forms, tests/fuzz/rbm_walk.cc on random stacks."""
    import numpy as np

    import rbm_walk as walk

    W = walk.W
    if not os.path.exists(walk.REF):
        pytest.skip("oracle/_ref/bin/rbm_walk not built (needs the reference sources at build time)")
    assert os.path.exists(walk.OURS), "kaldi_amd/bin/rbm_walk missing: make -C kaldi_amd/csrc host"
    rs = np.random.RandomState(11)
    counts, bad = {}, []
    for case in range(10):
        layers, desc = walk.random_stack(rs)
        model = str(tmp_path / "m.rbm")
        open(model, "wb").write(W.O.write_nnet(layers, bool(rs.rand() < 0.5)))
        fr, fo = str(tmp_path / "ref.bin"), str(tmp_path / "ours.bin")
        rc_r, err_r = W.run(walk.REF, model, 700 + case, 50, fr)
        rc_o, err_o = W.run(walk.OURS, model, 700 + case, 50, fo)
        assert rc_r == 0 and rc_o == 0, (desc, rc_r, rc_o, err_r[-300:], err_o[-300:])
        W.compare("case %d (%s)" % (case, desc), W.read_dump(fr), W.read_dump(fo), bad, counts, update_tags=walk.UPDATES)
    assert not bad, "\n".join(bad[:20])
    for what in ("cd1_err", "reconstruct", "update_weight", "write_as_nnet", "write_as_decoder", "stack_feedforward"):
        assert counts.get(what, 0) > 0, what
