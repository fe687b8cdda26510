// Activation components.  Public interface of the reference's nnet/nnet-activation.h:27-247.
// <sigmoid> and <softmax> are the hot-path ones; the rest are thin wrappers kept so that models
// containing them load and run.
#ifndef KALDI_AMD_NNET_NNET_ACTIVATION_H_
#define KALDI_AMD_NNET_NNET_ACTIVATION_H_
#include "cudamatrix/cu-math.h"
#include "cudamatrix/cu-rand.h"
#include "nnet/nnet-component.h"

namespace kaldi {

/// dst <- src where dst already has src's shape (Component::Propagate / Backpropagate have sized it).  Written in
/// place also when dst is a window onto memory owned elsewhere -- the last component of a forward pass writes
/// straight into the caller's output slot -- where CopyFromMat would let go of the window and allocate.
inline void CopyInPlace(const CuMatrix<BaseFloat> &src, CuMatrix<BaseFloat> *dst) {
  KALDI_ASSERT(dst->NumRows() == src.NumRows() && dst->NumCols() == src.NumCols());
  dst->AssignInPlace(src);
}

/// y = s*x
class Scale : public Component {
 public:
  Scale(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet), scale_(1.0) {}
  ComponentType GetType() const { return kScale; }
  void SetScale(BaseFloat scale) { scale_ = scale; }
  void ReadData(std::istream &is, bool binary) { ReadBasicType(is, binary, &scale_); }
  void WriteData(std::ostream &os, bool binary) const {
    WriteBasicType(os, binary, scale_);
    if (!binary) os << "\n";
  }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) {
    CopyInPlace(in, out);
    out->Scale(scale_);
  }
  void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
    CopyInPlace(in_err, out_err);
    out_err->Scale(scale_);
  }

 private:
  BaseFloat scale_;
};

/// y = mask .* x with mask ~ Bernoulli(1 - drop_ratio), the same mask for the error
class Dropout : public Component {
 public:
  Dropout(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet), drop_ratio_(0.5) {}
  ComponentType GetType() const { return kDropout; }
  void SetDropRatio(BaseFloat ratio) {
    drop_ratio_ = ratio;
    if (prob_.NumRows() > 0) prob_.Set(1 - drop_ratio_);
  }
  BaseFloat GetDropRatio() const { return drop_ratio_; }
  void ReadData(std::istream &is, bool binary) {
    BaseFloat ratio;
    ReadBasicType(is, binary, &ratio);
    if (ratio > 0.0 && ratio < 1.0) drop_ratio_ = ratio;
    else KALDI_WARN << "Invalid drop ratio for the <dropout> layer, using default 0.5.";
  }
  void WriteData(std::ostream &os, bool binary) const {
    WriteBasicType(os, binary, drop_ratio_);
    if (!binary) os << "\n";
  }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) {
    if (prob_.NumRows() != in.NumRows() || prob_.NumCols() != in.NumCols()) {
      prob_.Resize(in.NumRows(), in.NumCols());
      mask_.Resize(in.NumRows(), in.NumCols());
      prob_.Set(1 - drop_ratio_);
    }
    cu_rand_.BinarizeProbs(prob_, &mask_);
    CopyInPlace(in, out);
    out->MulElements(mask_);
  }
  void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
    CopyInPlace(in_err, out_err);
    out_err->MulElements(mask_);
  }

 private:
  BaseFloat drop_ratio_;
  CuRand<BaseFloat> cu_rand_;
  CuMatrix<BaseFloat> prob_, mask_;
};

class Sigmoid : public Component {
 public:
  Sigmoid(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet) {}
  ComponentType GetType() const { return kSigmoid; }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) { cu::Sigmoid(in, out); }
  /// ey = y(1-y)ex with y = this layer's output, looked up in the owning Nnet's buffers
  void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err);
};

class Softmax : public Component {
 public:
  Softmax(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet) {}
  ComponentType GetType() const { return kSoftmax; }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) { cu::Softmax(in, out); }
  /// The error is passed through: with the cross-entropy objective the incoming error
  /// (posterior - target) already is the derivative w.r.t. the softmax input.
  void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) { CopyInPlace(in_err, out_err); }
};

class Relu : public Component {
 public:
  Relu(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet) {}
  ComponentType GetType() const { return kRelu; }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) { cu::Relu(in, out); }
  void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err);
};

class SoftRelu : public Component {
 public:
  SoftRelu(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet) {}
  ComponentType GetType() const { return kSoftRelu; }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) { cu::SoftRelu(in, out); }
  void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err);
};

}  // namespace kaldi
#endif
