// The GPU singleton: device selection, per-function profile.  Public interface of the
// reference's CuDevice (cudamatrix/cu-device.h:35-89).  There is no CPU fallback in this build:
// a process that cannot get a gfx950 GPU fails with an error instead of silently computing on
// the host.
#ifndef KALDI_AMD_CUDAMATRIX_CU_DEVICE_H_
#define KALDI_AMD_CUDAMATRIX_CU_DEVICE_H_
#include <stdlib.h>

#include <map>
#include <string>

#include "cudamatrix/cu-common.h"

namespace kaldi {

class CuDevice {
 public:
  static CuDevice &Instantiate() {
    static CuDevice dev;
    return dev;
  }
  /// true once a GPU has been selected (lazily on first use)
  bool Enabled() {
    EnsureInit();
    return true;
  }
  /// Pick a GPU by ordinal before first use (-1: the one with the most free memory)
  void SelectGpuId(int32 gpu_id);
  int32 ActiveGpuId() {
    EnsureInit();
    return active_gpu_id_;
  }
  void Verbose(bool v) { verbose_ = v; }
  /// Accumulates wall time per key; only meaningful with Profile(true), which makes every
  /// wrapper synchronise (the reference always did: cu-common.h:35-42).
  void AccuProfile(const std::string &key, double time) { profile_map_[key] += time; }
  void PrintProfile();
  void ResetProfile() { profile_map_.clear(); }
  void Profile(bool on) { profiling_ = on; }
  bool Profiling() const { return profiling_; }
  std::string GetFreeMemory();
  /// The stream every CuMatrix / CuVector operation is queued on (the main stream, unless a
  /// CuStreamScope has redirected the calling code to the side stream).
  khStream_t Stream() {
    EnsureInit();
    return stream_;
  }
  khStream_t MainStream() {
    EnsureInit();
    return main_stream_;
  }
  /// A second stream for work that may run beside the main stream's (weight updates beside the
  /// backward pass).  Whoever queues work there orders it against the main stream with events
  /// and joins it back before returning; everything else only ever sees the main stream.
  khStream_t SideStream() {
    EnsureInit();
    if (side_stream_ == NULL) CU_SAFE_CALL(hipK_stream_create(&side_stream_));
    return side_stream_;
  }
  /// A third stream for small independent kernels (bias updates) that must not sit in the way of
  /// either chain of GEMMs; same rules as the side stream.
  khStream_t AuxStream() {
    EnsureInit();
    if (aux_stream_ == NULL) CU_SAFE_CALL(hipK_stream_create(&aux_stream_));
    return aux_stream_;
  }
  /// A stream for host-to-device uploads that should run beside the training kernels (the tools'
  /// feature uploads); same rules as the side stream.  Nothing on the device ever waits for it: the host does.
  /// (Streams may be created in any order.  A hang that once depended on this stream being created before
  /// the update streams came from a null-stream hipMemset in the device library's launch path, not from here.)
  khStream_t CopyStream() {
    EnsureInit();
    if (copy_stream_ == NULL) CU_SAFE_CALL(hipK_stream_create(&copy_stream_));
    return copy_stream_;
  }
  /// `waiter` continues only after everything queued on `done` so far has finished (no host wait).
  void Order(khStream_t done, khStream_t waiter, khEvent_t *ev) {
    if (*ev == NULL) CU_SAFE_CALL(hipK_event_create_ordering(ev));
    CU_SAFE_CALL(hipK_event_record(*ev, done));
    CU_SAFE_CALL(hipK_stream_wait_event(waiter, *ev));
  }
  /// marks "everything queued on `done` so far" in *ev (created on first use) for a later hipK_stream_wait_event
  void Record(khStream_t done, khEvent_t *ev) {
    if (*ev == NULL) CU_SAFE_CALL(hipK_event_create_ordering(ev));
    CU_SAFE_CALL(hipK_event_record(*ev, done));
  }
  /// Blocks until everything queued on the library's streams has finished (the main stream and,
  /// where they exist, the update streams: updates put off to the next forward pass may be the last
  /// work of a training step).
  void Synchronize() {
    EnsureInit();
    CU_SAFE_CALL(hipK_stream_synchronize(main_stream_));
    if (side_stream_ != NULL) CU_SAFE_CALL(hipK_stream_synchronize(side_stream_));
    if (aux_stream_ != NULL) CU_SAFE_CALL(hipK_stream_synchronize(aux_stream_));
  }
  void EnsureInit() {
    if (!initialised_) SelectGpuId(requested_gpu_id_);
  }

 private:
  friend class CuStreamScope;
  CuDevice()
      : initialised_(false), requested_gpu_id_(-1), active_gpu_id_(-1), verbose_(true), profiling_(false),
        stream_(NULL), main_stream_(NULL), side_stream_(NULL), aux_stream_(NULL), copy_stream_(NULL) {}
  ~CuDevice() {}
  bool initialised_;
  int32 requested_gpu_id_, active_gpu_id_;
  bool verbose_, profiling_;
  khStream_t stream_, main_stream_, side_stream_, aux_stream_, copy_stream_;
  std::map<std::string, double> profile_map_;
  KALDI_DISALLOW_COPY_AND_ASSIGN(CuDevice);
};

/// Collectives queued while one of these is alive may go out as one launch (hipK_comm_group_begin / _end); the
/// group is closed when the scope is left, also by an exception.
class CommGroupScope {
 public:
  CommGroupScope() { CU_SAFE_CALL(hipK_comm_group_begin()); }
  ~CommGroupScope() { hipK_comm_group_end(); }

 private:
  KALDI_DISALLOW_COPY_AND_ASSIGN(CommGroupScope);
};

/// While alive, operations of the calling code are queued on `s` instead of the main stream.
class CuStreamScope {
 public:
  explicit CuStreamScope(khStream_t s) : saved_(CuDevice::Instantiate().stream_) { CuDevice::Instantiate().stream_ = s; }
  ~CuStreamScope() { CuDevice::Instantiate().stream_ = saved_; }

 private:
  khStream_t saved_;
  KALDI_DISALLOW_COPY_AND_ASSIGN(CuStreamScope);
};

}  // namespace kaldi
#endif
