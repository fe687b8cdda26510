// Restricted Boltzmann machine with Bernoulli or Gaussian units on either side, trained by CD-1.
// Public interface of the reference's RbmBase / Rbm (nnet/nnet-rbm.h:26-333); vis_hid_ is stored
// (hidden x visible).  Propagate / Reconstruct are one fused GEMM each (bias and sigmoid in the
// epilogue), RbmUpdate is two GEMMs whose second epilogue applies the step.
#ifndef KALDI_AMD_NNET_NNET_RBM_H_
#define KALDI_AMD_NNET_NNET_RBM_H_
#include "cudamatrix/cu-math.h"
#include "cudamatrix/cu-rand.h"
#include "nnet/nnet-component.h"

namespace kaldi {

class RbmBase : public UpdatableComponent {
 public:
  typedef enum { BERNOULLI, GAUSSIAN } RbmNodeType;
  RbmBase(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : UpdatableComponent(dim_in, dim_out, nnet) {}
  virtual void Reconstruct(const CuMatrix<BaseFloat> &hid_state, CuMatrix<BaseFloat> *vis_probs) = 0;
  virtual void RbmUpdate(const CuMatrix<BaseFloat> &pos_vis, const CuMatrix<BaseFloat> &pos_hid,
                         const CuMatrix<BaseFloat> &neg_vis, const CuMatrix<BaseFloat> &neg_hid) = 0;
  /// the same step, which also adds the reconstruction error 1/2 * sum((neg_vis - pos_vis)^2) to a device counter
  virtual void RbmUpdate(const CuMatrix<BaseFloat> &pos_vis, const CuMatrix<BaseFloat> &pos_hid,
                         const CuMatrix<BaseFloat> &neg_vis, const CuMatrix<BaseFloat> &neg_hid, double *mse_loss) = 0;
  /// Propagate(in, probs) followed by cu_rand->BinarizeProbs(*probs, states) -- the positive phase of a CD-1 step
  /// with Bernoulli hidden units (nnetbin/rbm-train-cd1-frmshuff.cc:141-146) -- in one launch: same probabilities,
  /// same states, same generator state afterwards.  Not in the reference.
  virtual void PropagateSample(const CuMatrix<BaseFloat> &in, CuRand<BaseFloat> *cu_rand, CuMatrix<BaseFloat> *probs,
                               CuMatrix<BaseFloat> *states) = 0;
  virtual RbmNodeType VisType() const = 0;
  virtual RbmNodeType HidType() const = 0;
  virtual void WriteAsNnet(std::ostream &os, bool binary) const = 0;
  virtual void WriteAsAutoEncoder(std::ostream &os, bool isEncoder, bool binary) const = 0;
};

class Rbm : public RbmBase {
 public:
  Rbm(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet)
      : RbmBase(dim_in, dim_out, nnet), vis_type_(BERNOULLI), hid_type_(BERNOULLI) {}
  ~Rbm() {}

  ComponentType GetType() const { return kRbm; }

  void ReadData(std::istream &is, bool binary) {
    std::string vis, hid;
    ReadToken(is, binary, &vis);
    ReadToken(is, binary, &hid);
    vis_type_ = ParseType(vis);
    hid_type_ = ParseType(hid);
    vis_hid_.Read(is, binary);
    vis_bias_.Read(is, binary);
    hid_bias_.Read(is, binary);
    KALDI_ASSERT(vis_hid_.NumRows() == output_dim_);
    KALDI_ASSERT(vis_hid_.NumCols() == input_dim_);
    KALDI_ASSERT(static_cast<MatrixIndexT>(vis_bias_.Dim()) == input_dim_);
    KALDI_ASSERT(static_cast<MatrixIndexT>(hid_bias_.Dim()) == output_dim_);
  }
  void WriteData(std::ostream &os, bool binary) const {
    WriteToken(os, binary, TypeName(vis_type_));
    WriteToken(os, binary, TypeName(hid_type_));
    vis_hid_.Write(os, binary);
    vis_bias_.Write(os, binary);
    hid_bias_.Write(os, binary);
  }

  /// hid = [sigmoid](vis * W^T + hid_bias)
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) {
    CuProfileScope p("Rbm::Propagate");
    CU_SAFE_CALL(hipF_affine_forward(S(), hid_type_ == BERNOULLI ? KH_ACT_SIGMOID : KH_ACT_NONE, KH_TRANS, in.NumRows(),
                                     output_dim_, input_dim_, in.Data(), in.Stride(), vis_hid_.Data(), vis_hid_.Stride(),
                                     hid_bias_.Data(), out->Data(), out->Stride()));
  }
  void PropagateSample(const CuMatrix<BaseFloat> &in, CuRand<BaseFloat> *cu_rand, CuMatrix<BaseFloat> *probs,
                       CuMatrix<BaseFloat> *states) {
    KALDI_ASSERT(hid_type_ == BERNOULLI && cu_rand != NULL && probs != states);
    if (input_dim_ != in.NumCols()) KALDI_ERR << "Nonmatching dims, component:" << input_dim_ << " data:" << in.NumCols();
    if (output_dim_ != probs->NumCols() || in.NumRows() != probs->NumRows()) probs->Resize(in.NumRows(), output_dim_);
    if (output_dim_ != states->NumCols() || in.NumRows() != states->NumRows()) states->Resize(in.NumRows(), output_dim_);
    static const bool two_calls = getenv("KALDI_AMD_UNFUSED_SAMPLING") != NULL;  // for A/B measurements
    if (two_calls || states->Stride() != probs->Stride()) {
      PropagateFnc(in, probs);
      cu_rand->BinarizeProbs(*probs, states);
      return;
    }
    CuProfileScope p("Rbm::PropagateSample");
    uint32 *z1, *z2, *z3, *z4;
    cu_rand->States(*probs, &z1, &z2, &z3, &z4);
    CU_SAFE_CALL(hipF_affine_forward_sample(S(), in.NumRows(), output_dim_, input_dim_, in.Data(), in.Stride(), vis_hid_.Data(),
                                            vis_hid_.Stride(), hid_bias_.Data(), probs->Data(), probs->Stride(), states->Data(),
                                            states->Stride(), z1, z2, z3, z4));
  }
  void BackpropagateFnc(const CuMatrix<BaseFloat> &, CuMatrix<BaseFloat> *) {
    KALDI_ERR << "Cannot backpropagate through RBM!"
              << "Better convert it to <BiasedLinearity>";
  }
  virtual void Update(const CuMatrix<BaseFloat> &, const CuMatrix<BaseFloat> &) {
    KALDI_ERR << "Cannot update RBM by backprop!"
              << "Better convert it to <BiasedLinearity>";
  }

  /// vis = [sigmoid](hid * W + vis_bias)
  void Reconstruct(const CuMatrix<BaseFloat> &hid_state, CuMatrix<BaseFloat> *vis_probs) {
    CuProfileScope p("Rbm::Reconstruct");
    if (output_dim_ != hid_state.NumCols())
      KALDI_ERR << "Nonmatching dims, component:" << output_dim_ << " data:" << hid_state.NumCols();
    if (input_dim_ != vis_probs->NumCols() || hid_state.NumRows() != vis_probs->NumRows())
      vis_probs->Resize(hid_state.NumRows(), input_dim_);
    CU_SAFE_CALL(hipF_affine_forward(S(), vis_type_ == BERNOULLI ? KH_ACT_SIGMOID : KH_ACT_NONE, KH_NO_TRANS,
                                     hid_state.NumRows(), input_dim_, output_dim_, hid_state.Data(), hid_state.Stride(),
                                     vis_hid_.Data(), vis_hid_.Stride(), vis_bias_.Data(), vis_probs->Data(),
                                     vis_probs->Stride()));
  }

  /// CD-1 step from the positive and negative statistics (weights, then both bias vectors)
  void RbmUpdate(const CuMatrix<BaseFloat> &pos_vis, const CuMatrix<BaseFloat> &pos_hid,
                 const CuMatrix<BaseFloat> &neg_vis, const CuMatrix<BaseFloat> &neg_hid) {
    CuProfileScope p("Rbm::RbmUpdate");
    KALDI_ASSERT(pos_vis.NumRows() == pos_hid.NumRows() && pos_vis.NumRows() == neg_vis.NumRows() &&
                 pos_vis.NumRows() == neg_hid.NumRows() && pos_vis.NumCols() == neg_vis.NumCols() &&
                 pos_hid.NumCols() == neg_hid.NumCols() && pos_vis.NumCols() == input_dim_ &&
                 pos_hid.NumCols() == output_dim_);
    vis_hid_corr_.Resize(vis_hid_.NumRows(), vis_hid_.NumCols());  // no-op after the first call
    vis_bias_corr_.Resize(vis_bias_.Dim());
    hid_bias_corr_.Resize(hid_bias_.Dim());
    // One stream: both weight statistics in one launch where the operands allow it, the bias statistics inside it
    // (else in one small launch).  (The bias half and the reconstruction error used to run on a second stream beside the
    // GEMMs: the event packets that fork and join the streams left ~6 us gaps in the main stream's queue, three
    // per step, more than the small kernels take on their own.)
    CU_SAFE_CALL(hipF_rbm_update(S(), pos_vis.NumRows(), output_dim_, input_dim_, pos_vis.Data(), pos_vis.Stride(),
                                 pos_hid.Data(), pos_hid.Stride(), neg_vis.Data(), neg_vis.Stride(), neg_hid.Data(),
                                 neg_hid.Stride(), learn_rate_, momentum_, l2_penalty_, vis_hid_.Data(),
                                 vis_hid_.Stride(), vis_hid_corr_.Data(), vis_hid_corr_.Stride(), vis_bias_.Data(),
                                 vis_bias_corr_.Data(), hid_bias_.Data(), hid_bias_corr_.Data()));
  }

  /// RbmUpdate and the reconstruction error of the same step, 1/2 * sum((neg_vis - pos_vis)^2) added to the device
  /// counter `mse_loss` (MseProgress::DeviceLoss): what the CD-1 loop's mse.Eval(neg_vis, pos_vis, ...) adds, without
  /// a launch of its own where the weight launch can carry it (hipF_rbm_update_mse).
  void RbmUpdate(const CuMatrix<BaseFloat> &pos_vis, const CuMatrix<BaseFloat> &pos_hid,
                 const CuMatrix<BaseFloat> &neg_vis, const CuMatrix<BaseFloat> &neg_hid, double *mse_loss) {
    CuProfileScope p("Rbm::RbmUpdate");
    KALDI_ASSERT(pos_vis.NumRows() == pos_hid.NumRows() && pos_vis.NumRows() == neg_vis.NumRows() &&
                 pos_vis.NumRows() == neg_hid.NumRows() && pos_vis.NumCols() == neg_vis.NumCols() &&
                 pos_hid.NumCols() == neg_hid.NumCols() && pos_vis.NumCols() == input_dim_ &&
                 pos_hid.NumCols() == output_dim_ && pos_vis.Stride() == neg_vis.Stride());
    vis_hid_corr_.Resize(vis_hid_.NumRows(), vis_hid_.NumCols());  // no-op after the first call
    vis_bias_corr_.Resize(vis_bias_.Dim());
    hid_bias_corr_.Resize(hid_bias_.Dim());
    CU_SAFE_CALL(hipF_rbm_update_mse(S(), pos_vis.NumRows(), output_dim_, input_dim_, pos_vis.Data(), pos_vis.Stride(),
                                     pos_hid.Data(), pos_hid.Stride(), neg_vis.Data(), neg_vis.Stride(), neg_hid.Data(),
                                     neg_hid.Stride(), learn_rate_, momentum_, l2_penalty_, vis_hid_.Data(),
                                     vis_hid_.Stride(), vis_hid_corr_.Data(), vis_hid_corr_.Stride(), vis_bias_.Data(),
                                     vis_bias_corr_.Data(), hid_bias_.Data(), hid_bias_corr_.Data(), mse_loss));
  }

  RbmNodeType VisType() const { return vis_type_; }
  RbmNodeType HidType() const { return hid_type_; }

  /// the RBM as the first layer(s) of a DNN: <biasedlinearity> [+ <sigmoid>]
  void WriteAsNnet(std::ostream &os, bool binary) const {
    WriteAffine(os, binary, OutputDim(), InputDim(), false, hid_bias_, HidType() == BERNOULLI);
  }
  void WriteAsAutoEncoder(std::ostream &os, bool isEncoder, bool binary) const {
    if (isEncoder) WriteAffine(os, binary, OutputDim(), InputDim(), false, hid_bias_, HidType() == BERNOULLI);
    else WriteAffine(os, binary, InputDim(), OutputDim(), true, vis_bias_, VisType() == BERNOULLI);
  }

  void SetWeight(const Matrix<BaseFloat> &mat) { vis_hid_.CopyFromMat(mat); }
  void GetWeight(Matrix<BaseFloat> *mat) { vis_hid_.CopyToMat(mat); }
  void GetWeight(CuMatrix<BaseFloat> *mat) { mat->CopyFromMat(vis_hid_); }
  void GetVisibleBias(Vector<BaseFloat> *vec) { vis_bias_.CopyToVec(vec); }
  void SetVisibleBias(const Vector<BaseFloat> &vec) { vis_bias_.CopyFromVec(vec); }
  void GetHiddenBias(Vector<BaseFloat> *vec) { hid_bias_.CopyToVec(vec); }
  void SetHiddenBias(const Vector<BaseFloat> &vec) { hid_bias_.CopyFromVec(vec); }

 protected:
  static khStream_t S() { return CuDevice::Instantiate().Stream(); }
  static RbmNodeType ParseType(const std::string &s) {
    if (s == "bern") return BERNOULLI;
    if (s == "gauss") return GAUSSIAN;
    KALDI_ERR << "Unknown RBM unit type " << s;
    return BERNOULLI;
  }
  static const char *TypeName(RbmNodeType t) { return t == BERNOULLI ? "bern" : "gauss"; }
  void WriteAffine(std::ostream &os, bool binary, int32 dim_out, int32 dim_in, bool transposed,
                   const CuVector<BaseFloat> &bias, bool sigmoid) const {
    WriteToken(os, binary, Component::TypeToMarker(Component::kBiasedLinearity));
    WriteBasicType(os, binary, dim_out);
    WriteBasicType(os, binary, dim_in);
    if (!binary) os << "\n";
    if (transposed) {
      Matrix<BaseFloat> tmp;
      vis_hid_.CopyToMat(&tmp);
      tmp.Transpose();
      tmp.Write(os, binary);
    } else {
      vis_hid_.Write(os, binary);
    }
    bias.Write(os, binary);
    if (sigmoid) {
      WriteToken(os, binary, Component::TypeToMarker(Component::kSigmoid));
      WriteBasicType(os, binary, dim_out);
      WriteBasicType(os, binary, dim_out);
    }
    if (!binary) os << "\n";
  }

  CuMatrix<BaseFloat> vis_hid_;
  CuVector<BaseFloat> vis_bias_;
  CuVector<BaseFloat> hid_bias_;
  CuMatrix<BaseFloat> vis_hid_corr_;
  CuVector<BaseFloat> vis_bias_corr_;
  CuVector<BaseFloat> hid_bias_corr_;
  RbmNodeType vis_type_;
  RbmNodeType hid_type_;
};

}  // namespace kaldi
#endif
