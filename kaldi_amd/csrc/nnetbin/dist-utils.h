// Multi-GPU plumbing shared by the nnetbin tools: one process per GPU, rank / world size from
// the command line or the launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK as set by
// torch.distributed.run or mpirun-style wrappers), RCCL unique id exchanged through a file.
// The reference tools are single-GPU; with one rank everything here is a no-op.
#ifndef KALDI_AMD_NNETBIN_DIST_UTILS_H_
#define KALDI_AMD_NNETBIN_DIST_UTILS_H_
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <fstream>
#include <string>
#include <vector>

#include "cudamatrix/cu-device.h"
#include "util/parse-options.h"

namespace kaldi {

struct DistOptions {
  int32 num_ranks, rank, gpu_id;
  std::string comm_id_file;
  DistOptions() : num_ranks(0), rank(-1), gpu_id(-1) {}
  void Register(ParseOptions *po) {
    po->Register("num-ranks", &num_ranks, "Number of cooperating processes, one GPU each (0: take WORLD_SIZE, else 1)");
    po->Register("rank", &rank, "Rank of this process (-1: take RANK, else 0)");
    po->Register("gpu-id", &gpu_id, "GPU ordinal (-1: LOCAL_RANK when multi-rank, else the GPU with most free memory)");
    po->Register("comm-id-file", &comm_id_file, "File through which rank 0 hands the RCCL unique id to the other ranks");
  }
  /// fills the unset fields from the environment
  void Resolve() {
    const char *e;
    if (num_ranks <= 0) num_ranks = (e = getenv("WORLD_SIZE")) ? atoi(e) : 1;
    if (rank < 0) rank = (e = getenv("RANK")) ? atoi(e) : 0;
    if (num_ranks < 1 || rank < 0 || rank >= num_ranks) KALDI_ERR << "Invalid rank " << rank << " of " << num_ranks;
    if (gpu_id < 0 && num_ranks > 1) gpu_id = (e = getenv("LOCAL_RANK")) ? atoi(e) : rank;
    if (comm_id_file.empty() && num_ranks > 1) {
      const char *port = getenv("MASTER_PORT");
      comm_id_file = std::string("/tmp/kaldi_amd_comm_") + (port ? port : "default");
    }
  }
};

/// What tells this launch's id file from a stale one left by a launch that died.  Where the launcher gives the
/// launch a name -- KALDI_AMD_RUN_ID, else TORCHELASTIC_RUN_ID (exported by torch.distributed.run) -- the nonce is
/// that name and the rendezvous port: the same on every node and in every shell.  Without a name the parent
/// process id stands in for one: the ranks of one launch on one node are children of one launcher and a
/// relaunch has another.  Ranks started by hand from several shells, or on several nodes without such a
/// launcher, must therefore export a common KALDI_AMD_RUN_ID.
inline std::string CommNonce(long parent_pid) {
  const char *run = getenv("KALDI_AMD_RUN_ID");
  if (!run || !*run) run = getenv("TORCHELASTIC_RUN_ID");
  const char *port = getenv("MASTER_PORT");
  if (run && *run) return std::string(run) + ":" + (port ? port : "-");
  return std::string("-:") + (port ? port : "-") + ":" + std::to_string(parent_pid);
}
inline std::string CommNonce() { return CommNonce(static_cast<long>(getppid())); }

/// The id file: [8-byte magic | uint32 nonce length | nonce | id].  Written to a fresh file (O_EXCL, mode 0600)
/// beside the final name and renamed into place, so a reader never sees half of it.
inline bool WriteCommIdFile(const std::string &file, const std::string &nonce, const unsigned char *id, size_t id_bytes) {
  static const char magic[8] = {'K', 'A', 'M', 'D', 'I', 'D', '0', '1'};
  const std::string tmp = file + ".tmp." + std::to_string(static_cast<long>(getpid()));
  unlink(file.c_str());  // whatever an earlier launch left
  unlink(tmp.c_str());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600);
  if (fd < 0) return false;
  const uint32 n = nonce.size();
  bool ok = write(fd, magic, sizeof(magic)) == static_cast<ssize_t>(sizeof(magic)) &&
            write(fd, &n, sizeof(n)) == static_cast<ssize_t>(sizeof(n)) &&
            write(fd, nonce.data(), n) == static_cast<ssize_t>(n) &&
            write(fd, id, id_bytes) == static_cast<ssize_t>(id_bytes);
  ok = (close(fd) == 0) && ok;
  return ok && rename(tmp.c_str(), file.c_str()) == 0;
}
/// True when `file` is a complete id file that carries `nonce`; anything else (absent, short, another launch's)
/// is "not yet".
inline bool ReadCommIdFile(const std::string &file, const std::string &nonce, unsigned char *id, size_t id_bytes) {
  static const char magic[8] = {'K', 'A', 'M', 'D', 'I', 'D', '0', '1'};
  std::ifstream is(file.c_str(), std::ios::binary);
  if (!is.good()) return false;
  char m[8];
  uint32 n = 0;
  is.read(m, sizeof(m));
  is.read(reinterpret_cast<char *>(&n), sizeof(n));
  if (!is.good() || memcmp(m, magic, sizeof(m)) != 0 || n != nonce.size()) return false;
  std::string got(n, '\0');
  is.read(&got[0], n);
  is.read(reinterpret_cast<char *>(id), id_bytes);
  return is.gcount() == static_cast<std::streamsize>(id_bytes) && got == nonce;
}

/// selects the GPU and, with more than one rank and need_comm, builds the RCCL communicator
/// (a tool whose ranks never exchange data, like nnet-forward, passes need_comm = false).
/// Rank 0 hands the RCCL id to the others through --comm-id-file: [magic | nonce | id], written to a fresh
/// file (O_EXCL, mode 0600) and renamed into place; readers wait for a file that carries THEIR launch's nonce,
/// so an id left behind by a killed launch is never joined.  For a multi-node run the file must be on storage
/// every node sees and the ranks need a common run id (see CommNonce); ranks started by hand should also be
/// given a --comm-id-file of their own per launch.
inline void DistInit(const DistOptions &o, bool need_comm = true) {
  CuDevice::Instantiate().SelectGpuId(o.gpu_id);
  if (o.num_ranks == 1 || !need_comm) return;
  if (getenv("KALDI_AMD_COMM_LOOPBACK") != NULL) {
    // rehearsal on one GPU: this process runs as its rank of num_ranks with the library's stand-in communicator
    // (a sum all-reduce multiplies by num_ranks, min / max / barrier are local); exercises the multi-rank control flow
    CU_SAFE_CALL(hipK_comm_init_loopback(o.num_ranks));
    return;
  }
  if (const char *file = getenv("KALDI_AMD_COMM_HOST_FILE")) {
    // validation on a box whose ranks share one GPU: real ranks with their own data that exchange through a
    // mapped file instead of RCCL (blocking, host-staged).  The file must be new for every job.
    CU_SAFE_CALL(hipK_comm_init_host(file, o.rank, o.num_ranks, size_t(16) << 20));
    return;
  }
  unsigned char id[KH_COMM_ID_BYTES];
  const std::string nonce = CommNonce();
  if (o.rank == 0) {
    CU_SAFE_CALL(hipK_comm_get_unique_id(id));
    if (!WriteCommIdFile(o.comm_id_file, nonce, id, sizeof(id)))
      KALDI_ERR << "Cannot write the communicator id to " << o.comm_id_file;
  } else {
    bool ok = false;
    for (int tries = 0; tries < 6000 && !ok; ++tries) {  // up to 10 minutes
      ok = ReadCommIdFile(o.comm_id_file, nonce, id, sizeof(id));
      if (!ok) usleep(100000);
    }
    if (!ok)
      KALDI_ERR << "Timed out waiting for this launch's communicator id in " << o.comm_id_file
                << " (ranks started from different shells or nodes must export a common KALDI_AMD_RUN_ID)";
  }
  CU_SAFE_CALL(hipK_comm_init(id, o.rank, o.num_ranks));
  CU_SAFE_CALL(hipK_comm_barrier(CuDevice::Instantiate().Stream()));
  if (o.rank == 0) unlink(o.comm_id_file.c_str());  // every rank has joined
}

inline void DistFinalize(const DistOptions &o, bool need_comm = true) {
  if (o.num_ranks > 1 && need_comm) CU_SAFE_CALL(hipK_comm_destroy());
}

/// all-reduce of one host integer (op 0 sum, 1 min, 2 max); blocking
inline int64 DistAllReduce(const DistOptions &o, int64 value, int op) {
  if (o.num_ranks == 1) return value;
  static long long *dev = NULL;
  if (!dev) {
    void *p = NULL;
    CU_SAFE_CALL(hipK_malloc(&p, sizeof(long long)));
    dev = static_cast<long long *>(p);
  }
  khStream_t s = CuDevice::Instantiate().Stream();
  long long v = value;
  CU_SAFE_CALL(hipK_memcpy(s, dev, &v, sizeof(v), 0));
  CU_SAFE_CALL(hipK_allreduce_i64(s, dev, 1, op));
  CU_SAFE_CALL(hipK_memcpy(s, &v, dev, sizeof(v), 1));
  return v;
}

/// sum all-reduce of one host double; blocking
inline double DistAllReduceSum(const DistOptions &o, double value) {
  if (o.num_ranks == 1) return value;
  static double *dev = NULL;
  if (!dev) {
    void *p = NULL;
    CU_SAFE_CALL(hipK_malloc(&p, sizeof(double)));
    dev = static_cast<double *>(p);
  }
  khStream_t s = CuDevice::Instantiate().Stream();
  double v = value;
  CU_SAFE_CALL(hipK_memcpy(s, dev, &v, sizeof(v), 0));
  CU_SAFE_CALL(hipK_allreduce_sum_f64(s, dev, 1));
  CU_SAFE_CALL(hipK_memcpy(s, &v, dev, sizeof(v), 1));
  return v;
}

/// Utterance -> rank assignment, evaluated identically by every rank while it streams through the
/// same table: each utterance goes to the rank that has received the fewest frames so far (lowest
/// rank on ties), which balances shards by frames rather than by utterance count.
class ShardAssigner {
 public:
  explicit ShardAssigner(int32 num_ranks) : frames_(num_ranks, 0) {}
  int32 Assign(int64 num_frames) {
    int32 best = 0;
    for (size_t r = 1; r < frames_.size(); ++r)
      if (frames_[r] < frames_[best]) best = r;
    frames_[best] += num_frames;
    return best;
  }

 private:
  std::vector<int64> frames_;
};

/// "ark:out.JOB.ark" -> "ark:out.3.ark" for rank 2, the JOB=1:N convention of the recipes
inline std::string SubstituteJob(const std::string &s, int32 rank) {
  std::string out = s;
  const std::string job = std::to_string(rank + 1);
  size_t pos = 0;
  while ((pos = out.find("JOB", pos)) != std::string::npos) {
    out.replace(pos, 3, job);
    pos += job.size();
  }
  return out;
}

}  // namespace kaldi
#endif
