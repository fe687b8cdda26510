"""Diagnostic: the library's RCCL communicator with one rank (init, sum all-reduce, min, barrier, destroy),
alone and inside a process where torch has already loaded its own RCCL and opened a process group.
usage: python tools/rccl_probe.py [torch]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KALDI_HIP_COMM_FORCE_RCCL"] = "1"   # one rank, but through ncclAllReduce / ncclAllGather all the same
with_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if with_torch:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(8, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
from kaldi_amd import hip as h  # noqa: E402

h.init(0)
uid = (ctypes.c_ubyte * h.COMM_ID_BYTES)()
h.call("hipK_comm_get_unique_id", uid)
h.call("hipK_comm_init", uid, 0, 1)
x = np.arange(1 << 20, dtype=np.float32)
d = h.DevMatrix.from_numpy(x, False)
h.call("hipF_allreduce_sum", None, d.ptr, x.size)
assert (d.numpy()[0] == x).all()
g = h.DevMatrix(1, x.size, np.float32, pitched=False)
h.call("hipF_allgather", None, d.ptr, g.ptr, x.size)
assert (g.numpy()[0] == x).all()
# the sharded exchange: grouped reduce-scatter + all-reduce, step range, in-place all-gather
y = np.arange(1 << 16, dtype=np.float32)
e, bsum = h.DevMatrix.from_numpy(y, False), h.DevMatrix.from_numpy(y[:100].copy(), False)
h.call("hipK_comm_group_begin")
h.call("hipF_reduce_scatter_sum", None, e.ptr, y.size)
h.call("hipF_allreduce_sum", None, bsum.ptr, 100)
h.call("hipK_comm_group_end")
first, count = ctypes.c_size_t(), ctypes.c_size_t()
h.call("hipK_comm_shard_range", y.size, ctypes.byref(first), ctypes.byref(count))
assert (first.value, count.value) == (0, y.size)
h.call("hipF_allgather_shards", None, e.ptr, y.size)
assert (e.numpy()[0] == y).all() and (bsum.numpy()[0] == y[:100]).all()
v = h.DevMatrix.from_numpy(np.array([7], np.int64), False)
h.call("hipK_allreduce_i64", None, v.ptr, 1, 1)
assert int(v.numpy()[0, 0]) == 7
h.call("hipK_comm_barrier", None)
h.call("hipK_comm_destroy")
print("rccl probe ok", "(torch process group active)" if with_torch else "")
if with_torch:
    dist.destroy_process_group()
