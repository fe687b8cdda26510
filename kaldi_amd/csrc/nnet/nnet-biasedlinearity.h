// <biasedlinearity>: out = in * W^T + b.  Public interface of the reference's BiasedLinearity
// (nnet/nnet-biasedlinearity.h:26-157) with the same update rule (:71-102); on the device the
// bias add, the activation that follows, the sigmoid derivative of the layer below and the SGD
// weight pass ride in the epilogues of the three GEMMs.
#ifndef KALDI_AMD_NNET_NNET_BIASEDLINEARITY_H_
#define KALDI_AMD_NNET_NNET_BIASEDLINEARITY_H_
#include "cudamatrix/cu-math.h"
#include "cudamatrix/cu-stlvector.h"
#include "nnet/nnet-component.h"

namespace kaldi {

class BiasedLinearity : public UpdatableComponent {
 public:
  BiasedLinearity(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet)
      : UpdatableComponent(dim_in, dim_out, nnet),
        linearity_(dim_out, dim_in), bias_(dim_out), linearity_corr_(dim_out, dim_in), bias_corr_(dim_out) {}
  ~BiasedLinearity() {}

  ComponentType GetType() const { return kBiasedLinearity; }

  void ReadData(std::istream &is, bool binary) {
    linearity_.Read(is, binary);
    bias_.Read(is, binary);
    KALDI_ASSERT(linearity_.NumRows() == output_dim_);
    KALDI_ASSERT(linearity_.NumCols() == input_dim_);
    KALDI_ASSERT(static_cast<MatrixIndexT>(bias_.Dim()) == output_dim_);
  }
  void WriteData(std::ostream &os, bool binary) const {
    linearity_.Write(os, binary);
    bias_.Write(os, binary);
  }

  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) { Forward(in, KH_ACT_NONE, out); }

  void BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
    CuProfileScope p("BiasedLinearity::Backpropagate");
    CU_SAFE_CALL(hipF_affine_backward(S(), in_err.NumRows(), output_dim_, input_dim_, in_err.Data(), in_err.Stride(),
                                      linearity_.Data(), linearity_.Stride(), NULL, 0, out_err->Data(), out_err->Stride()));
  }

  /// out = act(in * W^T + b), act = KH_ACT_NONE | KH_ACT_SIGMOID | KH_ACT_SOFTMAX, one pass
  void PropagateFused(const CuMatrix<BaseFloat> &in, int act, CuMatrix<BaseFloat> *out) {
    if (input_dim_ != in.NumCols()) KALDI_ERR << "Nonmatching dims, component:" << input_dim_ << " data:" << in.NumCols();
    if (output_dim_ != out->NumCols() || in.NumRows() != out->NumRows()) out->Resize(in.NumRows(), output_dim_);
    Forward(in, act, out);
  }
  /// The output layer of a training step: softmax(in * W^T + b) and Xent::EvalVec against the labels
  /// in one pass (diff = posteriors - onehot).  `out` gets the posteriors only if write_post.
  void PropagateFusedXent(const CuMatrix<BaseFloat> &in, const CuStlVector<int32> &target, bool write_post,
                          CuMatrix<BaseFloat> *out, CuMatrix<BaseFloat> *diff, BaseFloat *log_post, int32 *max_id,
                          long long *counters, double *loss) {
    CuProfileScope p("BiasedLinearity::PropagateXent");
    if (input_dim_ != in.NumCols()) KALDI_ERR << "Nonmatching dims, component:" << input_dim_ << " data:" << in.NumCols();
    KALDI_ASSERT(static_cast<int32>(target.Dim()) == in.NumRows() && out != diff);
    if (output_dim_ != out->NumCols() || in.NumRows() != out->NumRows()) out->Resize(in.NumRows(), output_dim_);
    if (output_dim_ != diff->NumCols() || in.NumRows() != diff->NumRows()) diff->Resize(in.NumRows(), output_dim_);
    KALDI_ASSERT(out->Stride() == diff->Stride());
    CU_SAFE_CALL(hipF_affine_forward_xent(S(), KH_TRANS, in.NumRows(), output_dim_, input_dim_, in.Data(), in.Stride(),
                                          linearity_.Data(), linearity_.Stride(), bias_.Data(), out->Data(), out->Stride(),
                                          write_post ? 1 : 0, target.Data(), diff->Data(), diff->Stride(), log_post, max_id,
                                          counters, loss));
  }
  /// out_err = (in_err * W) .* y .* (1-y): this layer's backprop followed by that of the sigmoid
  /// that produced this layer's input y
  void BackpropagateFused(const CuMatrix<BaseFloat> &in_err, const CuMatrix<BaseFloat> &y, CuMatrix<BaseFloat> *out_err) {
    CuProfileScope p("BiasedLinearity::BackpropagateFused");
    if (output_dim_ != in_err.NumCols())
      KALDI_ERR << "Nonmatching dims, component:" << output_dim_ << " data:" << in_err.NumCols();
    KALDI_ASSERT(y.NumCols() == input_dim_ && y.NumRows() == in_err.NumRows());
    if (input_dim_ != out_err->NumCols() || in_err.NumRows() != out_err->NumRows())
      out_err->Resize(in_err.NumRows(), input_dim_);
    CU_SAFE_CALL(hipF_affine_backward(S(), in_err.NumRows(), output_dim_, input_dim_, in_err.Data(), in_err.Stride(),
                                      linearity_.Data(), linearity_.Stride(), y.Data(), y.Stride(), out_err->Data(),
                                      out_err->Stride()));
  }

  void Update(const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err) {
    if (l1_penalty_ == 0.0 && !double_buffer_) {
      // the whole of nnet-biasedlinearity.h:71-102 (minus the L1 shrink) in one launch, weights in place
      CuProfileScope p("BiasedLinearity::Update");
      const int32 n = input.NumRows();
      KALDI_ASSERT(err.NumRows() == n && err.NumCols() == output_dim_ && input.NumCols() == input_dim_);
      const BaseFloat gscale = average_grad_ ? 1.0 / n : 1.0;
      const BaseFloat l2 = (l2_penalty_ != 0.0) ? learn_rate_ * l2_penalty_ * n : 0.0;
      CU_SAFE_CALL(hipF_affine_update_fused(S(), n, output_dim_, input_dim_, err.Data(), err.Stride(), input.Data(),
                                            input.Stride(), gscale, momentum_, learn_rate_, l2, linearity_.Data(),
                                            linearity_.Stride(), linearity_.Data(), linearity_.Stride(),
                                            KeepCorr() ? linearity_corr_.Data() : NULL, linearity_corr_.Stride(),
                                            bias_.Data(), KeepCorr() ? bias_corr_.Data() : NULL));
      if (l2_upper_bound_ > 0.0) linearity_.ApplyRowL2UpperBound(l2_upper_bound_);
      return;
    }
    UpdateWeights(input, err);
    UpdateBias(err);
  }

  /// The same step as Update(), written down as one item of hipF_affine_update_group instead of launched, so
  /// that a training step can run the updates of all its layers in one launch behind the back-propagation
  /// pass.  False (nothing written) when this layer's update is not the single fused launch (L1 penalty,
  /// second weight buffer); after the grouped launch has been queued the caller calls AfterGroupedUpdate().
  bool DescribeUpdate(const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err, khAffineUpdate *item) {
    if (l1_penalty_ != 0.0 || double_buffer_) return false;
    const int32 n = input.NumRows();
    KALDI_ASSERT(err.NumRows() == n && err.NumCols() == output_dim_ && input.NumCols() == input_dim_);
    item->N = n;
    item->O = output_dim_;
    item->I = input_dim_;
    item->e = err.Data();
    item->lde = err.Stride();
    item->in = input.Data();
    item->ldin = input.Stride();
    item->gscale = average_grad_ ? 1.0 / n : 1.0;
    item->momentum = momentum_;
    item->lr = learn_rate_;
    item->l2 = (l2_penalty_ != 0.0) ? learn_rate_ * l2_penalty_ * n : 0.0;
    item->W_in = linearity_.Data();
    item->ldw_in = linearity_.Stride();
    item->W_out = linearity_.Data();
    item->ldw_out = linearity_.Stride();
    item->corr = KeepCorr() ? linearity_corr_.Data() : NULL;
    item->ldcorr = linearity_corr_.Stride();
    item->bias = bias_.Data();
    item->bcorr = KeepCorr() ? bias_corr_.Data() : NULL;
    return true;
  }
  void AfterGroupedUpdate() {
    if (l2_upper_bound_ > 0.0) linearity_.ApplyRowL2UpperBound(l2_upper_bound_);
  }

  /// True when the two halves below are independent: the weight half then reads only `input`,
  /// `err` and the old weights, the bias half only `err`, and they may be queued on different
  /// streams.  (With an L1 penalty the reference's pass order is kept and everything happens in
  /// UpdateWeights.)
  bool SplitUpdate() const { return l1_penalty_ == 0.0; }

  /// Weight half of Update.  With SetDoubleBuffer(true) the new weights are written to a second
  /// buffer, which then becomes the component's weight matrix: work queued earlier keeps reading
  /// the old values from the old buffer while this step runs.
  void UpdateWeights(const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err) {
    CuProfileScope p("BiasedLinearity::Update");
    const int32 n = input.NumRows();
    KALDI_ASSERT(err.NumRows() == n && err.NumCols() == output_dim_ && input.NumCols() == input_dim_);
    const BaseFloat gscale = average_grad_ ? 1.0 / n : 1.0;
    const BaseFloat l2 = (l2_penalty_ != 0.0) ? learn_rate_ * l2_penalty_ * n : 0.0;
    if (l1_penalty_ == 0.0) {
      // gradient GEMM with the momentum / weight-decay / SGD step in its epilogue
      CuMatrix<BaseFloat> *dst = &linearity_;
      if (double_buffer_) {
        if (linearity_next_.NumRows() != output_dim_ || linearity_next_.NumCols() != input_dim_)
          linearity_next_.Resize(output_dim_, input_dim_);
        dst = &linearity_next_;
      }
      CU_SAFE_CALL(hipF_affine_update_weights(S(), n, output_dim_, input_dim_, err.Data(), err.Stride(), input.Data(),
                                              input.Stride(), gscale, momentum_, learn_rate_, l2, linearity_.Data(),
                                              linearity_.Stride(), dst->Data(), dst->Stride(),
                                              KeepCorr() ? linearity_corr_.Data() : NULL, linearity_corr_.Stride()));
      if (double_buffer_) linearity_.Swap(&linearity_next_);
    } else {
      // the L1 shrink sits between the gradient and the step, so the passes stay separate
      linearity_corr_.AddMatMat(gscale, err, kTrans, input, kNoTrans, momentum_);
      bias_corr_.AddRowSumMat(gscale, err, momentum_);
      if (l2 != 0.0) linearity_.AddMat(-l2, linearity_);
      const BaseFloat l1 = learn_rate_ * n * l1_penalty_;
      cu::RegularizeL1(&linearity_, &linearity_corr_, l1, learn_rate_);
      linearity_.AddMat(-learn_rate_, linearity_corr_);
      bias_.AddVec(-learn_rate_, bias_corr_);
    }
    if (l2_upper_bound_ > 0.0) linearity_.ApplyRowL2UpperBound(l2_upper_bound_);
  }
  /// Bias half of Update (nothing to do when UpdateWeights has handled the bias: L1 path).
  void UpdateBias(const CuMatrix<BaseFloat> &err) {
    if (l1_penalty_ != 0.0) return;
    const int32 n = err.NumRows();
    const BaseFloat gscale = average_grad_ ? 1.0 / n : 1.0;
    CU_SAFE_CALL(hipF_affine_update_bias(S(), n, output_dim_, err.Data(), err.Stride(), gscale, momentum_, learn_rate_,
                                         bias_.Data(), KeepCorr() ? bias_corr_.Data() : NULL));
  }
  void SetDoubleBuffer(bool on) { double_buffer_ = on; }
  /// True when UpdateWeights leaves the buffer that earlier work reads untouched.
  bool UpdateKeepsOldWeights() const { return double_buffer_ && l1_penalty_ == 0.0; }

  /// Data-parallel form of Update, first half: the raw sums  gW = err^T * input  and
  /// gb = colsum(err)  into one contiguous buffer [gW (dim_out x dim_in, dense) | gb].
  void ComputeGradient(const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err, BaseFloat *grad_buf) {
    CuProfileScope p("BiasedLinearity::ComputeGradient");
    khAffineGradient item = {input.NumRows(), output_dim_, input_dim_, err.Data(), err.Stride(), input.Data(), input.Stride(),
                             grad_buf, input_dim_, grad_buf + static_cast<size_t>(output_dim_) * input_dim_};
    CU_SAFE_CALL(hipF_affine_gradient_group(S(), 1, &item));
  }
  size_t GradientSize() const { return static_cast<size_t>(output_dim_) * input_dim_ + output_dim_; }
  /// second half: momentum, weight decay and the step from the (all-reduced) sums over
  /// total_frames frames
  void ApplyGradient(const BaseFloat *grad_buf, int32 total_frames) {
    CuProfileScope p("BiasedLinearity::ApplyGradient");
    KALDI_ASSERT(l1_penalty_ == 0.0);
    const BaseFloat gscale = average_grad_ ? 1.0 / total_frames : 1.0;
    const BaseFloat l2 = (l2_penalty_ != 0.0) ? learn_rate_ * l2_penalty_ * total_frames : 0.0;
    CU_SAFE_CALL(hipF_affine_apply(S(), output_dim_, input_dim_, grad_buf, input_dim_,
                                   grad_buf + static_cast<size_t>(output_dim_) * input_dim_, gscale, momentum_,
                                   learn_rate_, l2, linearity_.Data(), linearity_.Stride(),
                                   KeepCorr() ? linearity_corr_.Data() : NULL, linearity_corr_.Stride(), bias_.Data(),
                                   KeepCorr() ? bias_corr_.Data() : NULL));
    if (l2_upper_bound_ > 0.0) linearity_.ApplyRowL2UpperBound(l2_upper_bound_);
  }

  /// Sharded data-parallel form.  The parameters of the network's layers live one after another in arenas owned
  /// by the Nnet (weights, gradients and, with momentum, corrections: the same layout in each): a layer takes
  /// ParameterSpan() floats, the memory of its weight matrix as it is (dim_out rows of Stride() floats, padding
  /// included) followed by its bias padded to a multiple of 64 floats.  The exchange between the ranks and the
  /// element-wise step then work on runs of layers at a time.  Not with an L1 penalty, whose shrink needs the
  /// reference's pass order.
  size_t WeightSpan() const { return static_cast<size_t>(output_dim_) * linearity_.Stride(); }
  size_t BiasSpan() const { return (static_cast<size_t>(output_dim_) + 63) & ~static_cast<size_t>(63); }
  size_t ParameterSpan() const { return WeightSpan() + BiasSpan(); }
  bool CanUseArena() const { return l1_penalty_ == 0.0 && linearity_corr_.Stride() == linearity_.Stride(); }
  /// Moves weights and bias into `mem` (ParameterSpan() floats, 256-byte aligned, zero-filled by the caller) and
  /// makes the component's matrix and vector windows onto it.
  void MoveParametersTo(BaseFloat *mem) {
    khStream_t s = S();
    CU_SAFE_CALL(hipK_memcpy(s, mem, linearity_.Data(), WeightSpan() * sizeof(BaseFloat), 2));
    CU_SAFE_CALL(hipK_memcpy(s, mem + WeightSpan(), bias_.Data(), output_dim_ * sizeof(BaseFloat), 2));
    const MatrixIndexT stride = linearity_.Stride();
    linearity_.AttachView(mem, output_dim_, input_dim_, stride);
    bias_.AttachView(mem + WeightSpan(), output_dim_);
  }
  /// The same for the momentum state.
  void MoveCorrectionsTo(BaseFloat *mem) {
    khStream_t s = S();
    CU_SAFE_CALL(hipK_memcpy(s, mem, linearity_corr_.Data(), WeightSpan() * sizeof(BaseFloat), 2));
    CU_SAFE_CALL(hipK_memcpy(s, mem + WeightSpan(), bias_corr_.Data(), output_dim_ * sizeof(BaseFloat), 2));
    const MatrixIndexT stride = linearity_corr_.Stride();
    linearity_corr_.AttachView(mem, output_dim_, input_dim_, stride);
    bias_corr_.AttachView(mem + WeightSpan(), output_dim_);
  }
  /// Gives the parameters back memory of their own (the arena is about to go away).
  void OwnParametersAgain() {
    if (linearity_.IsView()) {
      CuMatrix<BaseFloat> w;
      w.CopyFromMat(linearity_);
      linearity_.Swap(&w);
      CuVector<BaseFloat> b;
      b.CopyFromVec(bias_);
      bias_.AttachView(NULL, 0);
      bias_.CopyFromVec(b);
    }
    if (linearity_corr_.IsView()) {
      CuMatrix<BaseFloat> w;
      w.CopyFromMat(linearity_corr_);
      linearity_corr_.Swap(&w);
      CuVector<BaseFloat> b;
      b.CopyFromVec(bias_corr_);
      bias_corr_.AttachView(NULL, 0);
      bias_corr_.CopyFromVec(b);
    }
  }
  /// This layer's item of the grouped gradient launch: gW = err^T * input laid out like the weight matrix and
  /// gb = colsum(err) behind it, into `grad` (ParameterSpan() floats; the padding is never written).
  void DescribeGradient(const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err, BaseFloat *grad,
                        khAffineGradient *item) const {
    const int32 n = input.NumRows();
    KALDI_ASSERT(err.NumRows() == n && err.NumCols() == output_dim_ && input.NumCols() == input_dim_);
    item->N = n;
    item->O = output_dim_;
    item->I = input_dim_;
    item->e = err.Data();
    item->lde = err.Stride();
    item->in = input.Data();
    item->ldin = input.Stride();
    item->gW = grad;
    item->ldg = linearity_.Stride();
    item->gb = grad + WeightSpan();
  }
  /// The two segments of the element-wise step (weights, bias) for this layer's place `offset` in the arenas,
  /// for a step on the sums over total_frames frames.
  void DescribeStep(size_t offset, int32 total_frames, khSgdSegment *w_seg, khSgdSegment *b_seg) const {
    KALDI_ASSERT(l1_penalty_ == 0.0);
    w_seg->begin = offset;
    w_seg->end = offset + WeightSpan();
    w_seg->gscale = average_grad_ ? 1.0 / total_frames : 1.0;
    w_seg->momentum = momentum_;
    w_seg->lr = learn_rate_;
    w_seg->l2 = (l2_penalty_ != 0.0) ? learn_rate_ * l2_penalty_ * total_frames : 0.0;
    *b_seg = *w_seg;
    b_seg->begin = w_seg->end;
    b_seg->end = offset + ParameterSpan();
    b_seg->l2 = 0.0;
  }
  bool NeedsCorrections() const { return KeepCorr(); }
  /// after the new weights have been gathered
  void FinishShardedStep() {
    if (l2_upper_bound_ > 0.0) linearity_.ApplyRowL2UpperBound(l2_upper_bound_);
  }

  void SetLinearityWeight(const CuMatrix<BaseFloat> &weight, bool trans) {
    if (trans) {
      Matrix<BaseFloat> mat;
      weight.CopyToMat(&mat);
      mat.Transpose();
      AssignWeights(mat);
    } else {
      AssignWeights(weight);
    }
  }
  void SetLinearityWeight(const Matrix<BaseFloat> &weight, bool trans) {
    if (trans) {
      Matrix<BaseFloat> mat(weight);
      mat.Transpose();
      AssignWeights(mat);
    } else {
      AssignWeights(weight);
    }
  }
  const CuMatrix<BaseFloat> &GetLinearityWeight() { return linearity_; }
  void SetBiasWeight(const CuVector<BaseFloat> &bias) { bias_.CopyFromVec(bias); }
  void SetBiasWeight(const Vector<BaseFloat> &bias) { bias_.CopyFromVec(bias); }
  const CuVector<BaseFloat> &GetBiasWeight() { return bias_; }
  /// Momentum state (zero after construction; never written to disk).  As in the reference
  /// (nnet/nnet-biasedlinearity.h:76-82) every update stores it, so that a momentum switched on in the middle of
  /// training starts from the gradient of the update before.  With momentum 0 nothing on the training path reads
  /// it, and a caller whose momentum stays what it is for the rest of the run (the trainers: one value per
  /// process) saves one full pass over the weight matrix per step with SetKeepGradient(false); a momentum that
  /// such a caller turns on later all the same starts from the last stored correction (zero if there was none).
  const CuMatrix<BaseFloat> &GetLinearityCorr() { return linearity_corr_; }
  const CuVector<BaseFloat> &GetBiasCorr() { return bias_corr_; }
  void SetKeepGradient(bool keep) { keep_gradient_ = keep; }
  void SetToIdentity() {
    Matrix<BaseFloat> mat(linearity_.NumRows(), linearity_.NumCols());
    mat.SetUnit();
    linearity_.CopyFromMat(mat);
    bias_.SetZero();
  }

 protected:
  static khStream_t S() { return CuDevice::Instantiate().Stream(); }
  bool KeepCorr() const { return momentum_ != 0.0 || keep_gradient_; }
  /// (a weight matrix that is a window onto the network's parameter arena is written in place)
  void AssignWeights(const CuMatrix<BaseFloat> &w) {
    if (!linearity_.IsView()) {
      linearity_.CopyFromMat(w);
      return;
    }
    KALDI_ASSERT(w.NumRows() == linearity_.NumRows() && w.NumCols() == linearity_.NumCols());
    linearity_.AssignInPlace(w);
  }
  void AssignWeights(const Matrix<BaseFloat> &w) {
    if (!linearity_.IsView()) {
      linearity_.CopyFromMat(w);
      return;
    }
    CuMatrix<BaseFloat> tmp;
    tmp.CopyFromMat(w);
    AssignWeights(tmp);
  }
  void Forward(const CuMatrix<BaseFloat> &in, int act, CuMatrix<BaseFloat> *out) {
    CuProfileScope p("BiasedLinearity::Propagate");
    CU_SAFE_CALL(hipF_affine_forward(S(), act, KH_TRANS, in.NumRows(), output_dim_, input_dim_, in.Data(), in.Stride(),
                                     linearity_.Data(), linearity_.Stride(), bias_.Data(), out->Data(), out->Stride()));
  }

  CuMatrix<BaseFloat> linearity_;
  CuVector<BaseFloat> bias_;
  CuMatrix<BaseFloat> linearity_corr_;
  CuVector<BaseFloat> bias_corr_;
  CuMatrix<BaseFloat> linearity_next_;  // second weight buffer (SetDoubleBuffer)
  bool keep_gradient_ = true;
  bool double_buffer_ = false;
};

}  // namespace kaldi
#endif
