"""One rank of a multi-process data-parallel job on a shared GPU (started by tests/test_gpu_dist.py): joins the
host-staged communicator, trains its own rows of every bunch through Nnet::SetDataParallel and writes the
parameters it ends with.
usage: python tests/dp_host_rank.py <model> <data.npz> <rank> <nranks> <exchange-file | rccl:id-file> <out.npz>
data.npz: X [steps, rows, dim] float32, T [steps, rows] int32, split [nranks + 1] row offsets of the ranks,
opts [lr, momentum, l2, l1, max_norm]; KALDI_AMD_DP_EXCHANGE / KALDI_AMD_DP_CHUNKS come from the environment."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import hip as h  # noqa: E402
import kaldi_amd.nnet as nn  # noqa: E402

model, data, rank, nranks, xfile, out = sys.argv[1], np.load(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
if xfile.startswith("rccl:"):
    # one GPU per rank, RCCL: rank 0 hands the unique id to the others through the id file
    import ctypes
    import time
    h.init(rank)
    path, nonce = xfile[5:], "test-%d" % nranks
    if rank == 0:
        uid = (ctypes.c_ubyte * h.COMM_ID_BYTES)()
        h.call("hipK_comm_get_unique_id", uid)
        nn.comm_id_file_write(path, nonce, bytes(uid))
    else:
        for _ in range(600):
            got = nn.comm_id_file_read(path, nonce, h.COMM_ID_BYTES)
            if got is not None:
                break
            time.sleep(0.1)
        uid = (ctypes.c_ubyte * h.COMM_ID_BYTES)(*got)
    h.call("hipK_comm_init", uid, rank, nranks)
else:
    h.init(0)
    h.call("hipK_comm_init_host", xfile.encode(), rank, nranks, 1 << 16)   # small slots: the large layers go in pieces
X, T, split, opts = data["X"], data["T"], data["split"], data["opts"]
dims = [int(d) for d in data["dims"]]
tr = nn.Trainer(model)
tr.set_train_opts(float(opts[0]), float(opts[1]), float(opts[2]), float(opts[3]), float(opts[4]), True, True)
tr.set_data_parallel(nranks)
lo, hi = int(split[rank]), int(split[rank + 1])
for s in range(X.shape[0]):
    x = h.DevMatrix.from_numpy(np.ascontiguousarray(X[s, lo:hi]))
    t = h.DevMatrix.from_numpy(np.ascontiguousarray(T[s, lo:hi]), pitched=False)
    tr.train_bunch(x.ptr, x.stride, hi - lo, dims[0], t.ptr)
params = {}
for k in range(len(dims) - 1):
    W, b = tr.layer_params(2 * k, dims[k + 1], dims[k])
    params["W%d" % k], params["b%d" % k] = W, b
if rank == 0:
    tr.write(out + ".nnet")
tr.close()
h.call("hipK_comm_barrier", None)
h.call("hipK_comm_destroy")
np.savez(out, **params)
print("rank %d of %d done" % (rank, nranks))
