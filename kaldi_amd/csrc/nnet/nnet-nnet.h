// Nnet: an ordered list of Components plus the forward / backward buffers.  Public interface of
// the reference's Nnet (nnet/nnet-nnet.h:36-111).
//
// Forward and backward passes recognise the sigmoid-DNN pattern of the hot path and run it fused
// (SetFusion(false) restores one kernel per component):
//   <biasedlinearity> + <sigmoid>|<softmax>  -> one GEMM whose epilogue adds the bias and applies
//       the activation; the pre-activation buffer [i+1] of the forward pass is not written;
//   backward through <biasedlinearity> preceded by <sigmoid> -> one GEMM whose epilogue multiplies
//       by y(1-y); the buffer [i-1] of the backward pass is not written;
//   <softmax> backward (a copy in the reference) -> no work, the error is used in place.
// A caller that looks at the buffers gets the reference's: see PropagateBuffer() below.
// The order of operations that matters for the result is the reference's (nnet-nnet.cc:50-118):
// per layer, top down, first back-propagate through the OLD weights, then update them.
#ifndef KALDI_AMD_NNET_NNET_NNET_H_
#define KALDI_AMD_NNET_NNET_NNET_H_
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "base/kaldi-common.h"
#include "matrix/matrix-lib.h"
#include "nnet/nnet-component.h"
#include "util/kaldi-io.h"

#include <cstdlib>

namespace kaldi {

class Xent;

class Nnet {
 public:
  Nnet() : learn_rate_(0.0), fusion_(true), overlap_(true), borrow_input_(false), dp_ranks_(1), dp_grad_(NULL), dp_grad_size_(0), join_event_(NULL), aux_join_event_(NULL) {}
  ~Nnet();

  /// forward pass, keeping every layer's output for Backpropagate
  void Propagate(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out);
  /// Propagate followed by xent->EvalVec(*out, target, diff).  When the network ends in
  /// <biasedlinearity><softmax> and fusion is on, the output layer, the softmax and the loss are one
  /// GEMM + one row kernel, and with keep_output == false the posteriors are never written (*out is
  /// then unspecified).  Not in the reference.
  void PropagateXent(const CuMatrix<BaseFloat> &in, const CuStlVector<int32> &target, Xent *xent,
                     CuMatrix<BaseFloat> *out, CuMatrix<BaseFloat> *diff, bool keep_output);
  /// backward pass + parameter update; out_err == NULL skips the input gradient of layer 0
  void Backpropagate(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err);
  /// forward pass that keeps two ping-pong buffers only (inference)
  void Feedforward(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out);

  MatrixIndexT InputDim() const;
  MatrixIndexT OutputDim() const;
  MatrixIndexT LayerCount() const { return nnet_.size(); }
  Component *Layer(MatrixIndexT index) {
    FinishExchange();  // (the caller may be after the layer's parameters)
    return nnet_[index];
  }
  int IndexOfLayer(const Component &comp) const;

  /// Every layer's input/output of the last forward pass ([0] the input, [i+1] the output of layer i) and the
  /// errors of the last backward pass ([i-1] w.r.t. the input of layer i), nnet/nnet-nnet.h:61-69.  The fused
  /// passes do not write all of them, so the first call of either accessor turns fusion off: every later pass
  /// writes every buffer, one kernel per component as the reference does (SetFusion(true) turns it back on).
  /// For the pass that has already run, PropagateBuffer() fills in what was left out by running those components
  /// again on their stored inputs (what the pass would have written, unless parameters changed since);
  /// BackpropagateBuffer() cannot -- the weights that pass went through are gone -- and leaves those entries
  /// empty until the next backward pass.
  const std::vector<CuMatrix<BaseFloat> > &PropagateBuffer() const {
    const_cast<Nnet *>(this)->BuffersWanted(true);
    return propagate_buf_;
  }
  const std::vector<CuMatrix<BaseFloat> > &BackpropagateBuffer() const {
    const_cast<Nnet *>(this)->BuffersWanted(false);
    return backpropagate_buf_;
  }
  /// For the network's own components: entry i of the forward buffers as the last pass left it, without the
  /// side effects of PropagateBuffer().
  const CuMatrix<BaseFloat> &BufferInFrontOfLayer(int32 i) const { return propagate_buf_[i]; }
  bool GetFusion() const { return fusion_; }

  void Read(const std::string &file);
  void Read(std::istream &in, bool binary);
  void Write(const std::string &file, bool binary);
  void Write(std::ostream &out, bool binary);
  void WriteFrontLayers(std::ostream &out, bool binary, int32 num_layers);

  /// lrate_factors: optional per-updatable-layer multipliers separated by ",:; "
  void SetLearnRate(BaseFloat lrate, const char *lrate_factors);
  BaseFloat GetLearnRate() { return learn_rate_; }
  std::string GetLearnRateString();
  void SetMomentum(BaseFloat mmt);
  /// SetMomentum for a caller that will not change the momentum again (not in the reference): with momentum 0
  /// the affine layers then drop the store of the correction that only a later, non-zero momentum would read.
  void SetConstantMomentum(BaseFloat mmt);
  void SetL2Penalty(BaseFloat l2);
  void SetL2UpperBound(BaseFloat bound);
  void SetL1Penalty(BaseFloat l1);
  void SetAverageGrad(bool flag);

  /// fused forward/backward for the sigmoid-DNN pattern (on by default)
  void SetFusion(bool on) { fusion_ = on; }
  /// Queue each layer's parameter update (and, data-parallel, its gradient all-reduce) on a second
  /// stream so that it runs beside the backward pass of the layers below (on by default; the
  /// arithmetic and its order inside every layer are unchanged).
  void SetOverlapUpdates(bool on) { overlap_ = on; }
  /// Propagate keeps a window onto `in` instead of the reference's copy into its first buffer.
  /// The caller must then leave `in` alone until Backpropagate has returned.  Off by default.
  void SetBorrowInput(bool on) { borrow_input_ = on; }
  /// Single GPU: launch the updates of all <biasedlinearity> layers together behind the back-propagation pass
  /// instead of one by one between its steps (on by default; the results are the same, see Backpropagate).
  void SetGroupUpdates(bool on) { group_updates_ = on; }
  /// Synchronous data-parallel training over `nranks` processes (one GPU each, communicator set up
  /// with hipK_comm_init): every <biasedlinearity> update becomes one exchange between the ranks and the
  /// same new weights on all of them, equal to one process training on the concatenated bunch.  The
  /// exchange (see SetDataParallel's definition) is chosen by the number of ranks; the environment variable
  /// KALDI_AMD_DP_EXCHANGE = gather | shard | sum overrides it.
  void SetDataParallel(int32 nranks);
  /// Sharded data-parallel training leaves the last step's exchange of new weights running beside whatever
  /// the caller does next; the next Propagate waits layer by layer for what it needs.  Anything else that reads
  /// parameters on the main stream calls this first (Write, Feedforward and Layer do): the main stream then
  /// continues only behind the exchange.  No host wait; nothing to do otherwise.
  void FinishExchange();

 private:
  typedef std::vector<Component *> NnetType;
  void UpdateLayer(int32 i, const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err);
  void DataParallelGradient(int32 i, class BiasedLinearity *bl, const CuMatrix<BaseFloat> &input,
                            const CuMatrix<BaseFloat> &err);

  /// A run of <biasedlinearity> layers (top-down) whose parameters lie one after another in the arenas and are
  /// exchanged together in the sharded data-parallel step.
  struct DpChunk {
    std::vector<int32> layers;    // component indices, top-down
    std::vector<size_t> offsets;  // of each layer's parameters in the arenas (floats)
    size_t begin, end;            // [begin, end) in the arenas; the length is a multiple of ranks * 64
    khEvent_t grad_ready, weights_read, weights_ready;
    bool pending;                 // weights_ready has been recorded and the main stream has not waited for it yet
    DpChunk() : begin(0), end(0), grad_ready(NULL), weights_read(NULL), weights_ready(NULL), pending(false) {}
  };
  void BuffersWanted(bool forward);
  bool SetupArenas();
  void ReleaseArenas();
  void EnsureCorrectionArena();
  void WaitForLayerWeights(int32 i);

  NnetType nnet_;
  std::vector<CuMatrix<BaseFloat> > propagate_buf_;
  std::vector<CuMatrix<BaseFloat> > backpropagate_buf_;
  std::vector<char> propagate_skipped_, backpropagate_skipped_;  // buffers the last (fused) pass did not write
  BaseFloat learn_rate_;
  bool fusion_, overlap_, borrow_input_;
  int32 dp_ranks_;
  BaseFloat *dp_grad_;   // device staging buffer for one layer's gradient
  size_t dp_grad_size_;
  std::vector<khEvent_t> layer_events_;  // "error of layer i ready, its weights no longer read"
  std::vector<khEvent_t> grad_events_;   // data-parallel: "local gradient of layer i complete"
  std::vector<khEvent_t> read_events_;   // data-parallel: "back-propagation of layer i has read its weights"
  bool dp_gather_ = false;  // data-parallel by gathering every rank's (input, error) rows instead of summing gradients
  bool dp_shard_ = false;   // summed gradients, but each rank steps one shard of every layer and the new weights are gathered
  std::vector<CuMatrix<BaseFloat> > dp_in_all_, dp_err_all_;  // the gathered rows, per layer
  CuMatrix<BaseFloat> dp_in_repack_;
  std::vector<BaseFloat *> dp_layer_grad_;  // data-parallel: one gradient buffer per layer
  std::vector<size_t> dp_layer_grad_size_;
  khEvent_t join_event_, aux_join_event_;
  khEvent_t dp_first_gather_ = NULL;  // gather exchange: behind the gathers of the top layer's rows (owned by grad_events_)
  // sharded exchange: the parameter arenas (weights, gradients, momentum state: one layout) and their chunks
  std::vector<DpChunk> dp_chunks_;
  std::vector<int32> dp_chunk_of_layer_;
  BaseFloat *dp_w_arena_ = NULL, *dp_g_arena_ = NULL, *dp_c_arena_ = NULL;
  size_t dp_arena_floats_ = 0;
  bool dp_arena_failed_ = false;  // this network cannot use the arenas (an L1 penalty, another updatable component)
  std::vector<khAffineGradient> dp_grad_items_;
  bool group_updates_ = getenv("KALDI_AMD_NO_GROUP_UPDATES") == NULL;  // (the variable: for A/B measurements)
  std::vector<khAffineUpdate> deferred_updates_;  // the updates of this Backpropagate call, launched at its end
  std::vector<int32> deferred_layers_;
  KALDI_DISALLOW_COPY_AND_ASSIGN(Nnet);
};

}  // namespace kaldi
#endif
