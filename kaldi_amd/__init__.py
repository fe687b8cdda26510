"""kaldi_amd: MI355X-native nnet1 hot path (device library + host tools).

The Python side is glue for tests and benchmarks only: it loads the in-tree shared libraries
through ctypes.  There is no CPU fallback -- importing `kaldi_amd.hip` fails loudly when
libkaldi_hip.so has not been built (run `python -c "import __graft_entry__ as g; g.build()"`).
"""
from . import hip  # noqa: F401
