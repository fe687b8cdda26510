// The GPU singleton: device selection, per-function profile.  Public interface of the
// reference's CuDevice (cudamatrix/cu-device.h:35-89).  There is no CPU fallback in this build:
// a process that cannot get a gfx950 GPU fails with an error instead of silently computing on
// the host.
#ifndef KALDI_AMD_CUDAMATRIX_CU_DEVICE_H_
#define KALDI_AMD_CUDAMATRIX_CU_DEVICE_H_
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
  khStream_t Stream() {
    EnsureInit();
    return stream_;
  }
  void Synchronize() {
    EnsureInit();
    CU_SAFE_CALL(hipK_stream_synchronize(stream_));
  }
  void EnsureInit() {
    if (!initialised_) SelectGpuId(requested_gpu_id_);
  }

 private:
  CuDevice() : initialised_(false), requested_gpu_id_(-1), active_gpu_id_(-1), verbose_(true), profiling_(false), stream_(NULL) {}
  ~CuDevice() {}
  bool initialised_;
  int32 requested_gpu_id_, active_gpu_id_;
  bool verbose_, profiling_;
  khStream_t stream_;
  std::map<std::string, double> profile_map_;
  KALDI_DISALLOW_COPY_AND_ASSIGN(CuDevice);
};

}  // namespace kaldi
#endif
