// Component / UpdatableComponent: the building blocks of an Nnet.  Public interface of the
// reference's nnet/nnet-component.h:49-277 (same type ids, markers, dimension checks and error
// texts); the factory (nnet-component.cc) knows the components of the nnet1 hot path.
#ifndef KALDI_AMD_NNET_NNET_COMPONENT_H_
#define KALDI_AMD_NNET_NNET_COMPONENT_H_
#include <iostream>
#include <string>

#include "base/kaldi-common.h"
#include "cudamatrix/cu-matrix.h"
#include "cudamatrix/cu-stlvector.h"
#include "cudamatrix/cu-vector.h"
#include "matrix/matrix-lib.h"

namespace kaldi {

class Nnet;

class Component {
 public:
  /// numeric values are those of the reference so that code switching on them keeps working
  typedef enum {
    kUnknown = 0x0,

    kUpdatableComponent = 0x0100,
    kBiasedLinearity,
    kSharedLinearity,
    kKrylovLinearity,
    kCMVNBL,
    kPosNegBL,
    kGaussBL,
    kMaskedBL,
    kLinBL,
    kHMMBL,
    kCodeBL,

    kActivationFunction = 0x0200,
    kSoftmax,
    kSigmoid,
    kRelu,
    kSoftRelu,
    kDropout,
    kScale,

    kTranform = 0x0400,
    kRbm,
    kExpand,
    kCopy,
    kTranspose,
    kBlockLinearity,
    kBias,
    kWindow,
    kLog,
    kMaskedRbm,
    kRoRbm,
    kGRbm,
    kLinRbm
  } ComponentType;

  struct key_value {
    const Component::ComponentType key;
    const char *value;
  };
  static const struct key_value kMarkerMap[];
  static const char *TypeToMarker(ComponentType t);
  static ComponentType MarkerToType(const std::string &s);

  Component(MatrixIndexT input_dim, MatrixIndexT output_dim, Nnet *nnet)
      : input_dim_(input_dim), output_dim_(output_dim), nnet_(nnet) {}
  virtual ~Component() {}

  virtual ComponentType GetType() const = 0;
  virtual bool IsUpdatable() const { return false; }
  MatrixIndexT InputDim() const { return input_dim_; }
  MatrixIndexT OutputDim() const { return output_dim_; }

  /// checks the input width, sizes *out (in.NumRows() x OutputDim()) and runs PropagateFnc
  void Propagate(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out);
  /// checks the error width, sizes *out_err (rows x InputDim()) and runs BackpropagateFnc
  void Backpropagate(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err);

  /// reads "<marker> dim_out dim_in" + data; NULL at end of stream
  static Component *Read(std::istream &is, bool binary, Nnet *nnet);
  void Write(std::ostream &os, bool binary) const;

 protected:
  virtual void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) = 0;
  virtual void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) = 0;
  virtual void ReadData(std::istream &is, bool binary) {}
  virtual void WriteData(std::ostream &os, bool binary) const {}

  MatrixIndexT input_dim_;
  MatrixIndexT output_dim_;
  Nnet *nnet_;

 private:
  KALDI_DISALLOW_COPY_AND_ASSIGN(Component);
};

class UpdatableComponent : public Component {
 public:
  UpdatableComponent(MatrixIndexT input_dim, MatrixIndexT output_dim, Nnet *nnet)
      : Component(input_dim, output_dim, nnet),
        learn_rate_(0.0), momentum_(0.0), l2_penalty_(0.0), l2_upper_bound_(0.0), l1_penalty_(0.0), average_grad_(false) {}
  virtual ~UpdatableComponent() {}

  bool IsUpdatable() const { return true; }
  /// gradient from (input, err) and SGD step
  virtual void Update(const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err) = 0;

  void SetLearnRate(BaseFloat lrate) { learn_rate_ = lrate; }
  BaseFloat GetLearnRate() { return learn_rate_; }
  void SetMomentum(BaseFloat mmt) { momentum_ = mmt; }
  BaseFloat GetMomentum() { return momentum_; }
  void SetL2Penalty(BaseFloat l2) { l2_penalty_ = l2; }
  BaseFloat GetL2Penalty() { return l2_penalty_; }
  void SetL2UpperBound(BaseFloat bound) { l2_upper_bound_ = bound; }
  BaseFloat GetL2UpperBound() { return l2_upper_bound_; }
  void SetL1Penalty(BaseFloat l1) { l1_penalty_ = l1; }
  BaseFloat GetL1Penalty() { return l1_penalty_; }
  void SetAverageGrad(bool flag) { average_grad_ = flag; }
  bool GetAverageGrad() { return average_grad_; }

 protected:
  BaseFloat learn_rate_;
  BaseFloat momentum_;
  BaseFloat l2_penalty_;
  BaseFloat l2_upper_bound_;
  BaseFloat l1_penalty_;
  bool average_grad_;
};

inline void Component::Propagate(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) {
  if (input_dim_ != in.NumCols()) KALDI_ERR << "Nonmatching dims, component:" << input_dim_ << " data:" << in.NumCols();
  if (output_dim_ != out->NumCols() || in.NumRows() != out->NumRows()) out->Resize(in.NumRows(), output_dim_);
  PropagateFnc(in, out);
}

inline void Component::Backpropagate(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
  if (output_dim_ != in_err.NumCols())
    KALDI_ERR << "Nonmatching dims, component:" << output_dim_ << " data:" << in_err.NumCols();
  if (input_dim_ != out_err->NumCols() || in_err.NumRows() != out_err->NumRows())
    out_err->Resize(in_err.NumRows(), input_dim_);
  BackpropagateFnc(in_err, out_err);
}

}  // namespace kaldi
#endif
