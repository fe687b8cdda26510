// Multi-GPU collectives: one process per GPU, RCCL over xGMI.  The reference has no
// counterpart (single GPU, src/cudamatrix/cu-device.cc); synchronous data-parallel training
// sums per-layer gradients with these calls, ordered on the same stream as the kernels.
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {
struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  bool loopback = false;  // stand-in for nranks replicas that all hold this process's data
  bool one_rank = false;  // ... that does the device work of rank 0 only (timing runs)
};
template <typename T>
__global__ void scale_kernel(T *buf, size_t n, T f) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = buf[i] * f;
}
Comm &comm() {
  static Comm c;
  return c;
}
// KALDI_HIP_COMM_FORCE_RCCL: a one-rank communicator still goes through RCCL (to exercise the calls on one GPU)
bool force_rccl() {
  static const bool f = getenv("KALDI_HIP_COMM_FORCE_RCCL") != nullptr;
  return f;
}
int nccl_fail(ncclResult_t r, const char *what) {
  snprintf(kh::ctx().last_msg, sizeof(kh::ctx().last_msg), "%s: %s", what, ncclGetErrorString(r));
  return KH_ERR_COMM;
}
}  // namespace

#define KH_NCCL(call)                                  \
  do {                                                 \
    ncclResult_t r__ = (call);                         \
    if (r__ != ncclSuccess) return nccl_fail(r__, #call); \
  } while (0)

static_assert(sizeof(ncclUniqueId) == KH_COMM_ID_BYTES, "unique id size");

KH_EXPORT int hipK_comm_get_unique_id(unsigned char id[KH_COMM_ID_BYTES]) {
  KH_REQUIRE(id != nullptr);
  ncclUniqueId uid;
  KH_NCCL(ncclGetUniqueId(&uid));
  memcpy(id, &uid, KH_COMM_ID_BYTES);
  return KH_OK;
}

KH_EXPORT int hipK_comm_init(const unsigned char id[KH_COMM_ID_BYTES], int rank, int nranks) {
  KH_CHECK_INIT();
  KH_REQUIRE(id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
  Comm &c = comm();
  if (c.comm) {
    KH_NCCL(ncclCommDestroy(c.comm));
    c.comm = nullptr;
  }
  ncclUniqueId uid;
  memcpy(&uid, id, KH_COMM_ID_BYTES);
  KH_NCCL(ncclCommInitRank(&c.comm, nranks, uid, rank));
  c.rank = rank;
  c.nranks = nranks;
  c.loopback = false;
  c.one_rank = false;
  return KH_OK;
}

KH_EXPORT int hipK_comm_init_loopback(int nranks) {
  KH_CHECK_INIT();
  KH_REQUIRE(nranks >= 1);
  int rc = hipK_comm_destroy();
  if (rc != KH_OK) return rc;
  Comm &c = comm();
  c.nranks = nranks;
  c.loopback = nranks > 1;
  c.one_rank = false;
  return KH_OK;
}

KH_EXPORT int hipK_comm_init_loopback_one_rank(int nranks) {
  int rc = hipK_comm_init_loopback(nranks);
  if (rc != KH_OK) return rc;
  comm().one_rank = comm().loopback;
  return KH_OK;
}

KH_EXPORT int hipK_comm_destroy(void) {
  Comm &c = comm();
  if (c.comm) {
    KH_NCCL(ncclCommDestroy(c.comm));
    c.comm = nullptr;
  }
  c.rank = 0;
  c.nranks = 1;
  c.loopback = false;
  c.one_rank = false;
  return KH_OK;
}

KH_EXPORT int hipK_comm_rank(int *rank, int *nranks) {
  if (rank) *rank = comm().rank;
  if (nranks) *nranks = comm().nranks;
  return KH_OK;
}

template <typename T>
static int allreduce(khStream_t s, T *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op = ncclSum) {
  KH_CHECK_INIT();
  if (count == 0) return KH_OK;
  KH_REQUIRE(buf != nullptr);
  Comm &c = comm();
  if (c.nranks == 1 && !(force_rccl() && c.comm)) return KH_OK;  // a single replica already holds the sum
  if (c.loopback) {
    // every replica holds the same values: the sum is nranks times them, min and max are them
    if (op == ncclSum)
      hipLaunchKernelGGL(scale_kernel<T>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, kh::pick(s), buf, count,
                         (T)c.nranks);
    return kh::launch_status();
  }
  if (!c.comm) return KH_ERR_NOT_INITIALISED;
  KH_NCCL(ncclAllReduce(buf, buf, count, dt, op, c.comm, kh::pick(s)));
  return KH_OK;
}

KH_EXPORT int hipF_allreduce_sum(khStream_t s, float *buf, size_t count) {
  return allreduce<float>(s, buf, count, ncclFloat32);
}
KH_EXPORT int hipF_allgather(khStream_t s, const float *send, float *recv, size_t count) {
  KH_CHECK_INIT();
  if (count == 0) return KH_OK;
  KH_REQUIRE(send != nullptr && recv != nullptr);
  Comm &c = comm();
  hipStream_t st = kh::pick(s);
  if (c.loopback || (c.nranks <= 1 && !(force_rccl() && c.comm))) {
    // every replica holds this process's block
    const int n = c.loopback ? c.nranks : 1;
    for (int r = 0; r < n; ++r)
      if (recv + (size_t)r * count != send)
        KH_HIP(hipMemcpyAsync(recv + (size_t)r * count, send, count * sizeof(float), hipMemcpyDeviceToDevice, st));
    return KH_OK;
  }
  if (!c.comm) return KH_ERR_NOT_INITIALISED;
  KH_NCCL(ncclAllGather(send, recv, count, ncclFloat32, c.comm, st));
  return KH_OK;
}

// ---- sharded exchange: reduce-scatter, step on the own shard, all-gather ------------------------
// A buffer of nranks * shard_count elements is cut into one shard per rank, shard r at r * shard_count.
KH_EXPORT int hipK_comm_group_begin(void) {
  Comm &c = comm();
  if (c.loopback || !c.comm) return KH_OK;
  KH_NCCL(ncclGroupStart());
  return KH_OK;
}
KH_EXPORT int hipK_comm_group_end(void) {
  Comm &c = comm();
  if (c.loopback || !c.comm) return KH_OK;
  KH_NCCL(ncclGroupEnd());
  return KH_OK;
}

KH_EXPORT int hipK_comm_shard_range(size_t shard_count, size_t *first, size_t *count) {
  KH_REQUIRE(first != nullptr && count != nullptr);
  Comm &c = comm();
  // the loopback stand-in plays every rank in turn, so it owns every shard (as one rank: rank 0's)
  *first = c.loopback ? 0 : (size_t)c.rank * shard_count;
  *count = (c.loopback && !c.one_rank) ? (size_t)c.nranks * shard_count : shard_count;
  return KH_OK;
}

KH_EXPORT int hipF_reduce_scatter_sum(khStream_t s, float *buf, size_t shard_count) {
  KH_CHECK_INIT();
  if (shard_count == 0) return KH_OK;
  KH_REQUIRE(buf != nullptr);
  Comm &c = comm();
  if (c.nranks == 1 && !(force_rccl() && c.comm)) return KH_OK;
  if (c.loopback) {
    // every replica holds the same values: each shard's sum is nranks times it
    const size_t count = c.one_rank ? shard_count : shard_count * c.nranks;
    hipLaunchKernelGGL(scale_kernel<float>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, kh::pick(s), buf, count,
                       (float)c.nranks);
    return kh::launch_status();
  }
  if (!c.comm) return KH_ERR_NOT_INITIALISED;
  // in place: the receive buffer is this rank's shard of the send buffer
  KH_NCCL(ncclReduceScatter(buf, buf + (size_t)c.rank * shard_count, shard_count, ncclFloat32, ncclSum, c.comm, kh::pick(s)));
  return KH_OK;
}

KH_EXPORT int hipF_allgather_shards(khStream_t s, float *buf, size_t shard_count) {
  KH_CHECK_INIT();
  if (shard_count == 0) return KH_OK;
  KH_REQUIRE(buf != nullptr);
  Comm &c = comm();
  if (c.loopback || (c.nranks == 1 && !(force_rccl() && c.comm))) return KH_OK;  // every shard is already here
  if (!c.comm) return KH_ERR_NOT_INITIALISED;
  KH_NCCL(ncclAllGather(buf + (size_t)c.rank * shard_count, buf, shard_count, ncclFloat32, c.comm, kh::pick(s)));
  return KH_OK;
}

KH_EXPORT int hipK_allreduce_sum_f64(khStream_t s, double *buf, size_t count) {
  return allreduce<double>(s, buf, count, ncclFloat64);
}
KH_EXPORT int hipK_allreduce_sum_i64(khStream_t s, long long *buf, size_t count) {
  return allreduce<long long>(s, buf, count, ncclInt64);
}

KH_EXPORT int hipK_allreduce_i64(khStream_t s, long long *buf, size_t count, int op) {
  KH_REQUIRE(op >= 0 && op <= 2);
  return allreduce<long long>(s, buf, count, ncclInt64, op == 0 ? ncclSum : op == 1 ? ncclMin : ncclMax);
}

KH_EXPORT int hipK_comm_barrier(khStream_t s) {
  KH_CHECK_INIT();
  Comm &c = comm();
  if (c.nranks == 1 || c.loopback) return hipK_stream_synchronize(s);
  if (!c.comm) return KH_ERR_NOT_INITIALISED;
  void *ws;
  int rc = kh::ensure_workspace(kh::pick(s), 256, &ws);
  if (rc != KH_OK) return rc;
  KH_NCCL(ncclAllReduce(ws, ws, 1, ncclFloat32, ncclSum, c.comm, kh::pick(s)));
  return hipK_stream_synchronize(s);
}
