#include "nnet/nnet-loss.h"

#include <math.h>

#include <iterator>
#include <sstream>

namespace kaldi {

static khStream_t S() { return CuDevice::Instantiate().Stream(); }

DeviceStats::DeviceStats() : counters_(NULL), loss_(NULL) {
  void *p = NULL;
  CU_SAFE_CALL(hipK_malloc(&p, 2 * sizeof(long long) + sizeof(double)));
  counters_ = static_cast<long long *>(p);
  loss_ = reinterpret_cast<double *>(counters_ + 2);
  Reset();
}
DeviceStats::~DeviceStats() {
  if (counters_) {
    hipK_device_synchronize();
    hipK_free(counters_);
  }
}
void DeviceStats::Reset() { CU_SAFE_CALL(hipK_memset_async(S(), counters_, 0, 2 * sizeof(long long) + sizeof(double))); }
void DeviceStats::Fetch(long long *correct, long long *frames, double *loss) const {
  struct {
    long long c[2];
    double l;
  } host;
  CU_SAFE_CALL(hipK_memcpy(S(), &host, counters_, sizeof(host), 1));
  if (correct) *correct = host.c[0];
  if (frames) *frames = host.c[1];
  if (loss) *loss = host.l;
}

void Xent::Eval(const CuMatrix<BaseFloat> &net_out, const CuMatrix<BaseFloat> &target, CuMatrix<BaseFloat> *diff) {
  KALDI_ASSERT(net_out.NumCols() == target.NumCols());
  KALDI_ASSERT(net_out.NumRows() == target.NumRows());
  diff->Resize(net_out.NumRows(), net_out.NumCols());
  diff->CopyFromMat(net_out);
  diff->AddMat(-1.0, target);
  // frame accuracy is not defined for soft labels
  soft_label_only_ = true;
  // -sum target*log(out), evaluated on the host like the reference does (nnet-loss.cc:41-53)
  Matrix<BaseFloat> t, o;
  target.CopyToMat(&t);
  net_out.CopyToMat(&o);
  Sync();
  for (int32 r = 0; r < o.NumRows(); ++r)
    for (int32 c = 0; c < o.NumCols(); ++c) {
      BaseFloat val = -t(r, c) * log(o(r, c));
      if (KALDI_ISINF(val)) val = 1e10;
      loss_ += val;
    }
  frames_ += net_out.NumRows();
}

void Xent::EvalVec(const CuMatrix<BaseFloat> &net_out, const std::vector<int32> &target, CuMatrix<BaseFloat> *diff) {
  KALDI_ASSERT(static_cast<int32>(target.size()) == net_out.NumRows());
  target_device_.CopyFromVec(target);
  EvalVec(net_out, target_device_, diff);
}

void Xent::EvalVec(const CuMatrix<BaseFloat> &net_out, const CuStlVector<int32> &target, CuMatrix<BaseFloat> *diff) {
  CuProfileScope p("Xent::EvalVec");
  KALDI_ASSERT(static_cast<int32>(target.Dim()) == net_out.NumRows());
  if (&net_out != diff) diff->Resize(net_out.NumRows(), net_out.NumCols());
  max_id_.Resize(net_out.NumRows());
  log_post_tgt_.Resize(net_out.NumRows());
  ::MatrixDim d = net_out.Dim();
  KALDI_ASSERT(diff->Stride() == net_out.Stride());
  CU_SAFE_CALL(hipF_xent_eval(S(), net_out.Data(), target.Data(), diff->Data(), log_post_tgt_.Data(), max_id_.Data(),
                              stats_.Counters(), stats_.Loss(), d));
}

// pulls the device totals into the host members (the device side restarts from zero)
void Xent::Sync() {
  long long correct = 0, frames = 0;
  double loss = 0.0;
  stats_.Fetch(&correct, &frames, &loss);
  stats_.Reset();
  frames_ += static_cast<int32>(frames);
  if (!soft_label_only_) correct_ += static_cast<int32>(correct);
  loss_ += loss;
}

void Xent::GetStats(int64 *frames, int64 *correct, double *loss) {
  Sync();
  if (frames) *frames = frames_;
  if (correct) *correct = correct_;
  if (loss) *loss = loss_;
}

std::string Xent::Report() {
  Sync();
  if (!soft_label_only_) return FormatReport(loss_, frames_, correct_);
  std::ostringstream oss;
  oss << "Xent:" << loss_ << " frames:" << frames_ << " err/frm:" << loss_ / frames_;
  return oss.str();
}

std::string Xent::FormatReport(double loss, int64 frames, int64 correct) {
  std::ostringstream oss;
  oss << "Xent:" << loss << " frames:" << frames << " err/frm:" << loss / frames;
  oss << "\nFRAME_ACCURACY >> " << 100.0 * correct / frames << "% <<";
  return oss.str();
}

void Xent::Reset() {
  stats_.Reset();
  frames_ = 0;
  correct_ = 0;
  loss_ = 0.0;
  soft_label_only_ = false;
}

BaseFloat Xent::GetFrameAccuracy() {
  Sync();
  if (frames_ <= 0) KALDI_ERR << "Incorrect number of frames have been seen: " << frames_;
  return 100.0 * correct_ / frames_;
}

void Mse::Eval(const CuMatrix<BaseFloat> &net_out, const CuMatrix<BaseFloat> &target, CuMatrix<BaseFloat> *diff) {
  CuProfileScope p("Mse::Eval");
  KALDI_ASSERT(net_out.NumCols() == target.NumCols());
  KALDI_ASSERT(net_out.NumRows() == target.NumRows());
  diff->Resize(net_out.NumRows(), net_out.NumCols());
  KALDI_ASSERT(net_out.Stride() == target.Stride() && diff->Stride() == net_out.Stride());
  CU_SAFE_CALL(hipF_mse_eval(S(), net_out.Data(), target.Data(), diff->Data(), NULL, stats_.Loss(), net_out.Dim()));
  frames_ += net_out.NumRows();
}

std::string Mse::Report() {
  double loss = 0.0;
  stats_.Fetch(NULL, NULL, &loss);
  stats_.Reset();
  loss_ += loss;
  std::ostringstream oss;
  oss << "Mse:" << loss_ << " frames:" << frames_ << " err/frm:" << loss_ / frames_ << std::endl;
  return oss.str();
}

void MseProgress::Eval(const CuMatrix<BaseFloat> &net_out, const CuMatrix<BaseFloat> &target, CuMatrix<BaseFloat> *diff) {
  CuProfileScope p("MseProgress::Eval");
  KALDI_ASSERT(net_out.NumCols() == target.NumCols());
  KALDI_ASSERT(net_out.NumRows() == target.NumRows());
  diff->Resize(net_out.NumRows(), net_out.NumCols());
  KALDI_ASSERT(net_out.Stride() == target.Stride() && diff->Stride() == net_out.Stride());
  CU_SAFE_CALL(hipF_mse_eval(S(), net_out.Data(), target.Data(), diff->Data(), NULL, stats_.Loss(), net_out.Dim()));
  Evaluated(net_out.NumRows());
}

void MseProgress::Evaluated(int32 rows) {
  frames_progress_ += rows;
  if (frames_progress_ > progress_step_) {
    double loss = 0.0;
    stats_.Fetch(NULL, NULL, &loss);
    stats_.Reset();
    loss_progress_ += loss;
    const float loss_of_step = loss_progress_ / frames_progress_;
    loss_vec_.push_back(loss_of_step);
    frames_ += frames_progress_;
    loss_ += loss_progress_;
    KALDI_LOG << "Progress chunk #" << ++progress_ctr_ << " mse:" << loss_of_step << " [last "
              << frames_progress_ / 100 / 3600 << "h/" << frames_ / 100 / 3600 << "h]";
    frames_progress_ = 0;
    loss_progress_ = 0;
  }
}

std::string MseProgress::Report() {
  std::ostringstream oss;
  oss << "Mse:" << loss_ << " frames:" << frames_ << " err/frm:" << loss_ / frames_ << std::endl;
  oss << "progress: [";
  std::copy(loss_vec_.begin(), loss_vec_.end(), std::ostream_iterator<float>(oss, " "));
  oss << "]" << std::endl;
  return oss.str();
}

}  // namespace kaldi
