// Multi-GPU collectives: one process per GPU, RCCL over xGMI.  The reference has no
// counterpart (single GPU, src/cudamatrix/cu-device.cc); synchronous data-parallel training
// sums per-layer gradients with these calls, ordered on the same stream as the kernels.
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

#include "common.h"

namespace {
// ---- host-staged communicator ----------------------------------------------------------------
// Real ranks without RCCL: processes (which may share one GPU) exchange through a file that all of them map -- a
// control block with a counting barrier, then one slot per rank.  A collective copies the rank's data to its slot
// (device -> host, behind everything queued on the stream), meets the others at the barrier, combines the slots on
// the host IN RANK ORDER (so every rank computes the same bits, and a reduce-scatter gives the bits of an
// all-reduce), meets them again and copies the result back.  Blocking and slow by design: it exists so that the
// multi-rank code paths -- rank-dependent shard offsets, in-place reduce-scatter and all-gather, uneven data -- run
// with different data on every rank on a box with one GPU.
constexpr int kHostMaxRanks = 64;
struct HostPost {  // what a rank is about to do: compared across the ranks at the start of every collective
  unsigned long long seq, kind, count;
};
struct HostCtl {
  std::atomic<unsigned> count, generation, failed;
  HostPost post[kHostMaxRanks];
};
struct HostComm {
  int rank = 0, nranks = 1;
  char *base = nullptr;
  size_t map_bytes = 0, slot_bytes = 0;
  unsigned long long seq = 0;  // collectives entered so far
  int timeout_s = 120;
  bool trace = false;
  // Device copies go through this process's own pinned buffer (slot_bytes), and the CPU moves the bytes between it
  // and the shared slots: the GPU runtime never sees the mapping the ranks share.
  char *pinned = nullptr;
  double t0 = 0.0;
  double now() const {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec - t0;
  }
  void say(const char *what) const {
    if (trace) fprintf(stderr, "[host comm %9.3f ms] rank %d #%llu %s\n", 1e3 * now(), rank, seq, what);
  }
  static constexpr size_t kCtlBytes = 4096;
  HostCtl *ctl() { return reinterpret_cast<HostCtl *>(base); }
  char *slot(int r) { return base + kCtlBytes + (size_t)r * slot_bytes; }
  // false after timeout_s without the others (a rank died): the caller reports an error instead of hanging
  bool barrier() {
    HostCtl *c = ctl();
    if (c->failed.load()) return false;
    const unsigned g = c->generation.load();
    if (c->count.fetch_add(1) + 1 == (unsigned)nranks) {
      c->count.store(0);
      c->generation.store(g + 1);
      return true;
    }
    const time_t t0 = time(nullptr);
    for (unsigned spins = 0; c->generation.load() == g; ++spins) {
      // a short spin, then sleep: a waiting rank must not keep a core from the threads its peers' GPU runtime needs
      if (spins < 2000) sched_yield(); else usleep(50);
      if (c->failed.load() || time(nullptr) - t0 > timeout_s) {
        c->failed.store(1);
        return false;
      }
    }
    return true;
  }
  void unmap() {
    if (base) munmap(base, map_bytes);
    base = nullptr;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
  }
};

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  bool loopback = false;  // stand-in for nranks replicas that all hold this process's data
  bool one_rank = false;  // ... that does the device work of rank 0 only (timing runs)
  HostComm *host = nullptr;  // ranks that exchange through a mapped file instead of RCCL
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
int host_fail(const char *what) {
  snprintf(kh::ctx().last_msg, sizeof(kh::ctx().last_msg), "host communicator: %s", what);
  return KH_ERR_COMM;
}

#define KH_HOST_MEET(h)                                                                  \
  do {                                                                                   \
    if (!(h)->barrier()) return host_fail("a rank did not reach the collective (timed out or failed)"); \
  } while (0)

// Start of a collective: every rank posts what it is about to do, all meet, and each checks that all posted the same
// (the n-th collective of every rank must be the same call with the same count: a mismatch is a bug in the caller
// that a real transport would answer with a hang or with wrong sums).
int host_begin(HostComm *h, unsigned long long kind, unsigned long long count, const char *name) {
  HostPost &mine = h->ctl()->post[h->rank];
  mine.seq = ++h->seq;
  mine.kind = kind;
  mine.count = count;
  if (h->trace) fprintf(stderr, "[host comm %9.3f ms] rank %d #%llu %s count %llu\n", 1e3 * h->now(), h->rank, h->seq, name, count);
  if (!h->barrier()) {
    snprintf(kh::ctx().last_msg, sizeof(kh::ctx().last_msg),
             "host communicator: rank %d waited %d s for the others in collective #%llu (%s, count %llu)", h->rank,
             h->timeout_s, h->seq, name, count);
    return KH_ERR_COMM;
  }
  for (int r = 0; r < h->nranks; ++r) {
    const HostPost &o = h->ctl()->post[r];
    if (o.seq != mine.seq || o.kind != kind || o.count != count) {
      h->ctl()->failed.store(1);
      snprintf(kh::ctx().last_msg, sizeof(kh::ctx().last_msg),
               "host communicator: collective mismatch: rank %d is in #%llu (%s, count %llu), rank %d in #%llu (kind %llu, count %llu)",
               h->rank, mine.seq, name, count, r, o.seq, o.kind, o.count);
      return KH_ERR_COMM;
    }
  }
  KH_HOST_MEET(h);  // all have compared before anyone posts its next collective
  return KH_OK;
}

struct OpSum {
  static constexpr int kId = 0;
  template <typename T>
  static T apply(T a, T b) { return a + b; }
};
struct OpMin {
  static constexpr int kId = 1;
  template <typename T>
  static T apply(T a, T b) { return b < a ? b : a; }
};
struct OpMax {
  static constexpr int kId = 2;
  template <typename T>
  static T apply(T a, T b) { return b > a ? b : a; }
};

// pieces[k] of n elements each, device -> this rank's slot (piece k at k * n), all behind the stream's queued work
template <typename T>
int host_stage_out(HostComm *h, hipStream_t st, const T *const *pieces, int npieces, size_t n) {
  T *mine = reinterpret_cast<T *>(h->pinned);
  h->say("waits for its stream");
  KH_HIP(hipStreamSynchronize(st));
  h->say("copies out");
  for (int k = 0; k < npieces; ++k)
    KH_HIP(hipMemcpyAsync(mine + (size_t)k * n, pieces[k], n * sizeof(T), hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  memcpy(h->slot(h->rank), h->pinned, (size_t)npieces * n * sizeof(T));
  h->say("staged");
  return KH_OK;
}

// n bytes of host memory (a slot, a vector) -> device, through the pinned buffer
int host_stage_in(HostComm *h, hipStream_t st, void *dst, const void *src, size_t bytes) {
  memcpy(h->pinned, src, bytes);
  KH_HIP(hipMemcpyAsync(dst, h->pinned, bytes, hipMemcpyHostToDevice, st));
  KH_HIP(hipStreamSynchronize(st));
  h->say("copied in");
  return KH_OK;
}

// all-reduce of buf[0..count) in slot-sized chunks
template <typename T, typename Op>
int host_allreduce(HostComm *h, hipStream_t st, T *buf, size_t count) {
  int rc0 = host_begin(h, 100 + sizeof(T) + 10 * Op::kId, count, "all-reduce");
  if (rc0 != KH_OK) return rc0;
  const size_t cap = h->slot_bytes / sizeof(T);
  std::vector<T> acc;
  for (size_t off = 0; off < count; off += cap) {
    const size_t n = count - off < cap ? count - off : cap;
    const T *piece = buf + off;
    int rc = host_stage_out(h, st, &piece, 1, n);
    if (rc != KH_OK) return rc;
    KH_HOST_MEET(h);
    acc.assign(reinterpret_cast<const T *>(h->slot(0)), reinterpret_cast<const T *>(h->slot(0)) + n);
    for (int r = 1; r < h->nranks; ++r) {
      const T *o = reinterpret_cast<const T *>(h->slot(r));
      for (size_t i = 0; i < n; ++i) acc[i] = Op::apply(acc[i], o[i]);
    }
    KH_HOST_MEET(h);  // every rank has read the slots before anyone refills its own
    rc = host_stage_in(h, st, buf + off, acc.data(), n * sizeof(T));
    if (rc != KH_OK) return rc;
  }
  return KH_OK;
}

// in-place reduce-scatter: this rank's shard of buf becomes the rank-ordered sum of every rank's copy of that shard;
// the other shards keep this rank's own values, as with RCCL's in-place form
int host_reduce_scatter(HostComm *h, hipStream_t st, float *buf, size_t shard_count) {
  int rc0 = host_begin(h, 200, shard_count, "reduce-scatter");
  if (rc0 != KH_OK) return rc0;
  const size_t cap = h->slot_bytes / sizeof(float) / h->nranks;
  if (cap == 0) return host_fail("slot too small for a reduce-scatter over this many ranks");
  std::vector<float> acc;
  std::vector<const float *> pieces(h->nranks);
  for (size_t off = 0; off < shard_count; off += cap) {
    const size_t n = shard_count - off < cap ? shard_count - off : cap;
    for (int r = 0; r < h->nranks; ++r) pieces[r] = buf + (size_t)r * shard_count + off;
    int rc = host_stage_out(h, st, pieces.data(), h->nranks, n);
    if (rc != KH_OK) return rc;
    KH_HOST_MEET(h);
    const size_t mine = (size_t)h->rank * n;
    acc.assign(reinterpret_cast<const float *>(h->slot(0)) + mine, reinterpret_cast<const float *>(h->slot(0)) + mine + n);
    for (int r = 1; r < h->nranks; ++r) {
      const float *o = reinterpret_cast<const float *>(h->slot(r)) + mine;
      for (size_t i = 0; i < n; ++i) acc[i] += o[i];
    }
    KH_HOST_MEET(h);
    rc = host_stage_in(h, st, buf + (size_t)h->rank * shard_count + off, acc.data(), n * sizeof(float));
    if (rc != KH_OK) return rc;
  }
  return KH_OK;
}

// all-gather: rank r's `count` elements at send land at recv + r * count on every rank
int host_allgather(HostComm *h, hipStream_t st, const float *send, float *recv, size_t count) {
  int rc0 = host_begin(h, 300, count, "all-gather");
  if (rc0 != KH_OK) return rc0;
  const size_t cap = h->slot_bytes / sizeof(float);
  for (size_t off = 0; off < count; off += cap) {
    const size_t n = count - off < cap ? count - off : cap;
    const float *piece = send + off;
    int rc = host_stage_out(h, st, &piece, 1, n);
    if (rc != KH_OK) return rc;
    KH_HOST_MEET(h);
    for (int r = 0; r < h->nranks; ++r) {
      float *dst = recv + (size_t)r * count + off;
      if (r == h->rank && dst == send + off) continue;
      rc = host_stage_in(h, st, dst, h->slot(r), n * sizeof(float));
      if (rc != KH_OK) return rc;
    }
    KH_HOST_MEET(h);
  }
  return KH_OK;
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
  int rc = hipK_comm_destroy();
  if (rc != KH_OK) return rc;
  Comm &c = comm();
  ncclUniqueId uid;
  memcpy(&uid, id, KH_COMM_ID_BYTES);
  KH_NCCL(ncclCommInitRank(&c.comm, nranks, uid, rank));
  c.rank = rank;
  c.nranks = nranks;
  c.loopback = false;
  c.one_rank = false;
  return KH_OK;
}

KH_EXPORT int hipK_comm_init_host(const char *path, int rank, int nranks, size_t slot_bytes) {
  KH_CHECK_INIT();
  KH_REQUIRE(path != nullptr && nranks >= 1 && nranks <= kHostMaxRanks && rank >= 0 && rank < nranks && slot_bytes >= 4096 &&
             slot_bytes % 256 == 0);
  static_assert(sizeof(HostCtl) <= HostComm::kCtlBytes, "control block");
  int rc = hipK_comm_destroy();
  if (rc != KH_OK) return rc;
  const int fd = open(path, O_RDWR | O_CREAT, 0600);
  if (fd < 0) return host_fail("cannot open the exchange file");
  HostComm *h = new HostComm;
  h->rank = rank;
  h->nranks = nranks;
  h->slot_bytes = slot_bytes;
  h->map_bytes = HostComm::kCtlBytes + (size_t)nranks * slot_bytes;
  if (const char *e = getenv("KALDI_HIP_COMM_TIMEOUT")) h->timeout_s = atoi(e) > 0 ? atoi(e) : h->timeout_s;
  h->trace = getenv("KALDI_HIP_COMM_TRACE") != nullptr;
  // every rank sizes the file to the same length; bytes the file gains read as zero, which is the barrier's start state
  void *m = ftruncate(fd, (off_t)h->map_bytes) == 0 ? mmap(nullptr, h->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0)
                                                    : MAP_FAILED;
  close(fd);
  if (m == MAP_FAILED) {
    delete h;
    return host_fail("cannot size or map the exchange file");
  }
  h->base = static_cast<char *>(m);
  h->t0 = h->now();
  if (hipHostMalloc(reinterpret_cast<void **>(&h->pinned), slot_bytes, hipHostMallocDefault) != hipSuccess) {
    h->unmap();
    delete h;
    return host_fail("cannot allocate the pinned staging buffer");
  }
  Comm &c = comm();
  c.host = h;
  c.rank = rank;
  c.nranks = nranks;
  KH_HOST_MEET(h);  // all ranks have joined
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
  if (c.host) {
    c.host->unmap();
    delete c.host;
    c.host = nullptr;
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
  if (c.host)
    return op == ncclSum   ? host_allreduce<T, OpSum>(c.host, kh::pick(s), buf, count)
           : op == ncclMin ? host_allreduce<T, OpMin>(c.host, kh::pick(s), buf, count)
                           : host_allreduce<T, OpMax>(c.host, kh::pick(s), buf, count);
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
  if (c.host) return host_allgather(c.host, st, send, recv, count);
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
  if (c.host) return host_reduce_scatter(c.host, kh::pick(s), buf, shard_count);
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
  if (c.host) return host_allgather(c.host, kh::pick(s), buf + (size_t)c.rank * shard_count, buf, shard_count);
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
  if (c.host) {
    int rc = hipK_stream_synchronize(s);
    if (rc != KH_OK) return rc;
    return host_begin(c.host, 400, 0, "barrier");
  }
  if (!c.comm) return KH_ERR_NOT_INITIALISED;
  void *ws;
  int rc = kh::ensure_workspace(kh::pick(s), 256, &ws);
  if (rc != KH_OK) return rc;
  KH_NCCL(ncclAllReduce(ws, ws, 1, ncclFloat32, ncclSum, c.comm, kh::pick(s)));
  return hipK_stream_synchronize(s);
}
