// Objective functions.  Public interface of the reference's Xent / Mse / MseProgress
// (nnet/nnet-loss.h:28-120) with the same report strings (nnet-loss.cc:97-105, 149-207).
// The per-minibatch statistics (frames, correct frames, summed log-posterior / squared error)
// are accumulated ON THE DEVICE and fetched only when a report or an accuracy is asked for, so
// evaluating the loss does not stall the stream every minibatch.
#ifndef KALDI_AMD_NNET_NNET_LOSS_H_
#define KALDI_AMD_NNET_NNET_LOSS_H_
#include <string>
#include <vector>

#include "cudamatrix/cu-math.h"

namespace kaldi {

/// device-resident accumulators: int64 {correct, frames} and double {loss}
class DeviceStats {
 public:
  DeviceStats();
  ~DeviceStats();
  long long *Counters() { return counters_; }
  double *Loss() { return loss_; }
  void Reset();
  /// blocking read-back
  void Fetch(long long *correct, long long *frames, double *loss) const;

 private:
  long long *counters_;
  double *loss_;
  KALDI_DISALLOW_COPY_AND_ASSIGN(DeviceStats);
};

class Xent {
 public:
  Xent() : frames_(0), correct_(0), loss_(0.0), soft_label_only_(false) {}
  ~Xent() {}

  /// cross entropy against soft labels (one target row per frame)
  void Eval(const CuMatrix<BaseFloat> &net_out, const CuMatrix<BaseFloat> &target, CuMatrix<BaseFloat> *diff);
  /// cross entropy against hard labels: diff = net_out - onehot(target); counts frames whose
  /// arg-max equals the label.  diff may be &net_out.
  void EvalVec(const CuMatrix<BaseFloat> &net_out, const std::vector<int32> &target, CuMatrix<BaseFloat> *diff);
  /// same with the labels already on the device (no host<->device traffic at all)
  void EvalVec(const CuMatrix<BaseFloat> &net_out, const CuStlVector<int32> &target, CuMatrix<BaseFloat> *diff);

  /// For a kernel that evaluates the loss itself (Nnet::PropagateXent): the per-frame vectors sized
  /// for `rows` frames and the running device counters, exactly what EvalVec hands to its kernel.
  void FusedBuffers(int32 rows, BaseFloat **log_post, int32 **max_id, long long **counters, double **loss) {
    max_id_.Resize(rows);
    log_post_tgt_.Resize(rows);
    *log_post = log_post_tgt_.Data();
    *max_id = max_id_.Data();
    *counters = stats_.Counters();
    *loss = stats_.Loss();
  }

std::string Report();
  /// The report string for totals gathered elsewhere (all ranks of a data-parallel run), in the format of
  /// Report(): nnet-loss.cc:97-105.  The recipes' scheduler takes field 3 of the FRAME_ACCURACY line.
  static std::string FormatReport(double loss, int64 frames, int64 correct);
  
  void Reset();
  BaseFloat GetFrameAccuracy();
  /// per-frame log-posteriors and arg-max ids of the last EvalVec call (device)
  const CuVector<BaseFloat> &LogPostTgt() const { return log_post_tgt_; }
  const CuStlVector<int32> &MaxId() const { return max_id_; }
  /// totals so far (blocking): frames, correct frames, summed -log posterior
  void GetStats(int64 *frames, int64 *correct, double *loss);

 private:
  void Sync();
  int32 frames_;
  int32 correct_;
  double loss_;
  bool soft_label_only_;
  DeviceStats stats_;
  CuStlVector<int32> max_id_;
  CuStlVector<int32> target_device_;
  CuVector<BaseFloat> log_post_tgt_;
};

class Mse {
 public:
  Mse() : frames_(0), loss_(0.0) {}
  ~Mse() {}
  /// diff = net_out - target; loss += 0.5 * sum(diff^2)
  void Eval(const CuMatrix<BaseFloat> &net_out, const CuMatrix<BaseFloat> &target, CuMatrix<BaseFloat> *diff);
  std::string Report();

 private:
  int32 frames_;
  double loss_;
  DeviceStats stats_;
};

class MseProgress {
 public:
  explicit MseProgress(int32 progress_step = 1e6)
      : progress_step_(progress_step), progress_ctr_(0), frames_(0), frames_progress_(0), loss_(0.0), loss_progress_(0) {}
  ~MseProgress() {}
  /// like Mse::Eval, and logs the running MSE every progress_step frames
  void Eval(const CuMatrix<BaseFloat> &net_out, const CuMatrix<BaseFloat> &target, CuMatrix<BaseFloat> *diff);
  /// Eval in two parts, for a caller whose own launch adds 1/2 * sum((net_out - target)^2) of `rows` frames to the
  /// device counter (Rbm::RbmUpdate with a loss): the counter to add to, and the bookkeeping of Eval (frame
  /// counts, progress log) once that launch is queued on the stream.
  double *DeviceLoss() { return stats_.Loss(); }
  void Evaluated(int32 rows);
  std::string Report();

 private:
  int32 progress_step_;
  int32 progress_ctr_;
  int32 frames_;
  int32 frames_progress_;
  double loss_;
  double loss_progress_;
  std::vector<float> loss_vec_;
  DeviceStats stats_;
};

}  // namespace kaldi
#endif
