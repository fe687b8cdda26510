#include "nnet/nnet-nnet.h"

#include "nnet/nnet-loss.h"

#include <stdlib.h>

#include <string>
#include <string.h>

#include "nnet/nnet-activation.h"
#include "nnet/nnet-biasedlinearity.h"
#include "nnet/nnet-various.h"

namespace kaldi {

Nnet::~Nnet() {
  for (size_t i = 0; i < nnet_.size(); ++i) delete nnet_[i];  // (windows onto the arenas free nothing)
  if (dp_w_arena_ || dp_g_arena_ || dp_c_arena_) hipK_device_synchronize();
  if (dp_w_arena_) hipK_free(dp_w_arena_);
  if (dp_g_arena_) hipK_free(dp_g_arena_);
  if (dp_c_arena_) hipK_free(dp_c_arena_);
  for (size_t c = 0; c < dp_chunks_.size(); ++c) {
    if (dp_chunks_[c].grad_ready) hipK_event_destroy(dp_chunks_[c].grad_ready);
    if (dp_chunks_[c].weights_read) hipK_event_destroy(dp_chunks_[c].weights_read);
    if (dp_chunks_[c].weights_ready) hipK_event_destroy(dp_chunks_[c].weights_ready);
  }
  if (dp_grad_) hipK_free(dp_grad_);
  for (size_t i = 0; i < dp_layer_grad_.size(); ++i)
    if (dp_layer_grad_[i]) hipK_free(dp_layer_grad_[i]);
  for (size_t i = 0; i < grad_events_.size(); ++i)
    if (grad_events_[i]) hipK_event_destroy(grad_events_[i]);
  for (size_t i = 0; i < read_events_.size(); ++i)
    if (read_events_[i]) hipK_event_destroy(read_events_[i]);
  for (size_t i = 0; i < layer_events_.size(); ++i)
    if (layer_events_[i]) hipK_event_destroy(layer_events_[i]);
  if (join_event_) hipK_event_destroy(join_event_);
  if (aux_join_event_) hipK_event_destroy(aux_join_event_);
}

MatrixIndexT Nnet::InputDim() const {
  if (LayerCount() == 0) KALDI_ERR << "No layers in MLP";
  return nnet_.front()->InputDim();
}

MatrixIndexT Nnet::OutputDim() const {
  if (LayerCount() == 0) KALDI_ERR << "No layers in MLP";
  return nnet_.back()->OutputDim();
}

int32 Nnet::IndexOfLayer(const Component &comp) const {
  for (int32 i = 0; i < LayerCount(); ++i)
    if (&comp == nnet_[i]) return i;
  KALDI_ERR << "Component:" << &comp << " type:" << comp.GetType() << " not found in the MLP";
  return -1;
}

static inline int ActivationAfter(const std::vector<Component *> &net, int32 i) {
  if (i + 1 >= static_cast<int32>(net.size())) return -1;
  if (net[i]->GetType() != Component::kBiasedLinearity) return -1;
  if (net[i + 1]->InputDim() != net[i + 1]->OutputDim()) return -1;
  if (net[i + 1]->GetType() == Component::kSigmoid) return KH_ACT_SIGMOID;
  if (net[i + 1]->GetType() == Component::kSoftmax) return KH_ACT_SOFTMAX;
  return -1;
}

void Nnet::Propagate(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) {
  KALDI_ASSERT(NULL != out);
  const int32 L = LayerCount();
  if (L == 0) {
    out->CopyFromMat(in);
    return;
  }
  KALDI_ASSERT(static_cast<int32>(propagate_buf_.size()) >= L + 1);
  if (borrow_input_) propagate_buf_[0].AttachRows(in, 0, in.NumRows());
  else propagate_buf_[0].CopyFromMat(in);
  propagate_skipped_.assign(L + 1, 0);
  for (int32 i = 0; i < L; ++i) {
    const int act = fusion_ ? ActivationAfter(nnet_, i) : -1;
    WaitForLayerWeights(i);
    if (act >= 0) {
      static_cast<BiasedLinearity *>(nnet_[i])->PropagateFused(propagate_buf_[i], act, &propagate_buf_[i + 2]);
      propagate_skipped_[i + 1] = 1;
      ++i;  // the activation has been applied
    } else {
      nnet_[i]->Propagate(propagate_buf_[i], &propagate_buf_[i + 1]);
    }
  }
  CuMatrix<BaseFloat> &last = propagate_buf_[L];
  if (fusion_ && nnet_[L - 1]->GetType() == Component::kSoftmax && out->NumRows() == last.NumRows() &&
      out->NumCols() == last.NumCols()) {
    // Nothing reads the softmax output from the buffers during backprop (its Backpropagate only
    // forwards the error), so the result is handed over by exchanging the two allocations.
    out->Swap(&last);
    propagate_skipped_[L] = 1;
  } else {
    out->CopyFromMat(last);
  }
}

void Nnet::BuffersWanted(bool forward) {
  fusion_ = false;
  const int32 L = LayerCount();
  if (forward) {
    if (static_cast<int32>(propagate_skipped_.size()) != L + 1) return;  // no forward pass yet
    FinishExchange();
    for (int32 i = 0; i < L; ++i) {
      if (!propagate_skipped_[i + 1]) continue;
      nnet_[i]->Propagate(propagate_buf_[i], &propagate_buf_[i + 1]);
      propagate_skipped_[i + 1] = 0;
    }
  } else {
    for (size_t k = 0; k < backpropagate_skipped_.size() && k < backpropagate_buf_.size(); ++k) {
      if (backpropagate_skipped_[k]) backpropagate_buf_[k].Resize(0, 0);  // (nothing stale from an earlier pass)
      backpropagate_skipped_[k] = 0;
    }
  }
}

void Nnet::PropagateXent(const CuMatrix<BaseFloat> &in, const CuStlVector<int32> &target, Xent *xent,
                         CuMatrix<BaseFloat> *out, CuMatrix<BaseFloat> *diff, bool keep_output) {
  KALDI_ASSERT(NULL != out && NULL != diff && NULL != xent);
  const int32 L = LayerCount();
  static const bool two_calls = getenv("KALDI_AMD_UNFUSED_LOSS") != NULL;  // for A/B measurements
  const bool fusable = !two_calls && fusion_ && L >= 2 && nnet_[L - 1]->GetType() == Component::kSoftmax &&
                       nnet_[L - 2]->GetType() == Component::kBiasedLinearity;
  if (!fusable) {
    Propagate(in, out);
    xent->EvalVec(*out, target, diff);
    return;
  }
  KALDI_ASSERT(static_cast<int32>(propagate_buf_.size()) >= L + 1);
  if (borrow_input_) propagate_buf_[0].AttachRows(in, 0, in.NumRows());
  else propagate_buf_[0].CopyFromMat(in);
  propagate_skipped_.assign(L + 1, 0);
  propagate_skipped_[L - 1] = 1;
  for (int32 i = 0; i < L - 2; ++i) {
    const int act = ActivationAfter(nnet_, i);
    WaitForLayerWeights(i);
    if (act >= 0) {
      static_cast<BiasedLinearity *>(nnet_[i])->PropagateFused(propagate_buf_[i], act, &propagate_buf_[i + 2]);
      propagate_skipped_[i + 1] = 1;
      ++i;
    } else {
      nnet_[i]->Propagate(propagate_buf_[i], &propagate_buf_[i + 1]);
    }
  }
  BaseFloat *log_post;
  int32 *max_id;
  long long *counters;
  double *loss;
  xent->FusedBuffers(in.NumRows(), &log_post, &max_id, &counters, &loss);
  CuMatrix<BaseFloat> &last = propagate_buf_[L];
  WaitForLayerWeights(L - 2);
  static_cast<BiasedLinearity *>(nnet_[L - 2])
      ->PropagateFusedXent(propagate_buf_[L - 2], target, keep_output, &last, diff, log_post, max_id, counters, loss);
  if (keep_output) {
    if (out->NumRows() == last.NumRows() && out->NumCols() == last.NumCols()) {
      out->Swap(&last);
      propagate_skipped_[L] = 1;
    } else {
      out->CopyFromMat(last);
    }
  } else {
    propagate_skipped_[L] = 1;
  }
}

void Nnet::DataParallelGradient(int32 i, BiasedLinearity *bl, const CuMatrix<BaseFloat> &input,
                                const CuMatrix<BaseFloat> &err) {
  // one staging buffer for all layers: the queue that uses it is in order, so a layer's step has
  // consumed it before the next layer's gradient overwrites it
  const size_t need = bl->GradientSize();
  if (need > dp_grad_size_) {
    if (dp_grad_) {
      CU_SAFE_CALL(hipK_device_synchronize());
      CU_SAFE_CALL(hipK_free(dp_grad_));
    }
    void *p = NULL;
    CU_SAFE_CALL(hipK_malloc(&p, need * sizeof(BaseFloat)));
    dp_grad_ = static_cast<BaseFloat *>(p);
    dp_grad_size_ = need;
  }
  bl->ComputeGradient(input, err, dp_grad_);
  CU_SAFE_CALL(hipF_allreduce_sum(CuDevice::Instantiate().Stream(), dp_grad_, need));
}

void Nnet::UpdateLayer(int32 i, const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err) {
  if (!nnet_[i]->IsUpdatable()) return;
  UpdatableComponent *uc = dynamic_cast<UpdatableComponent *>(nnet_[i]);
  if (!(uc->GetLearnRate() > 0.0)) return;
  if (dp_ranks_ > 1) {
    if (nnet_[i]->GetType() != Component::kBiasedLinearity)
      KALDI_ERR << "Data-parallel training supports <biasedlinearity> layers only";
    BiasedLinearity *bl = static_cast<BiasedLinearity *>(nnet_[i]);
    DataParallelGradient(i, bl, input, err);
    bl->ApplyGradient(dp_grad_, input.NumRows() * dp_ranks_);
  } else {
    uc->Update(input, err);
  }
}

void Nnet::Backpropagate(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
  const int32 L = LayerCount();
  if (L == 0) KALDI_ERR << "Cannot backpropagate on empty network";
  KALDI_ASSERT(static_cast<int32>(propagate_buf_.size()) >= L + 1);
  KALDI_ASSERT(static_cast<int32>(backpropagate_buf_.size()) >= L - 1);

  // Two queues.  The main stream carries the chain of error back-propagations and the (small)
  // bias updates; a layer's weight update needs only its input, its output error and the old
  // weights, so it is queued on the side stream and runs beside the back-propagation of this
  // layer and of those below.  A <biasedlinearity> writes its new weights to a second buffer, so
  // its update waits only for the error (event recorded before the layer's own
  // back-propagation is queued); any other update waits until that back-propagation has read the
  // weights.  The side stream is joined back before returning: callers keep seeing one in-order
  // stream.
  CuDevice &dev = CuDevice::Instantiate();
  // Single GPU: everything stays on the main stream, in the reference's order.  Measured on the
  // Switchboard-sized net: the update GEMMs fill the GPU on their own, and running them on a second stream
  // beside the back-propagation cost more (event packets in the main stream's queue, two kernels halving each
  // other's occupancy) than the gaps they filled were worth: 0.747 -> 0.715 ms per step without the streams,
  // and each layer's bias update rides in its weight update's launch.  The update streams remain for
  // data-parallel training, where they carry the communication.
  static const bool force_overlap = getenv("KALDI_AMD_OVERLAP_UPDATES") != NULL;  // for A/B measurements
  const bool overlap = overlap_ && !dev.Profiling() && (dp_ranks_ > 1 || force_overlap);
  if (overlap && static_cast<int32>(layer_events_.size()) < L) layer_events_.resize(L, NULL);
  bool side_used = false, aux_used = false;
  // Sharded exchange (see SetDataParallel): gradients of a chunk of layers in one launch on the main stream as
  // soon as the error of the chunk's lowest layer exists, the chunk's reduce-scatter and the step on this rank's
  // shard on the exchange stream beside the rest of the pass; the all-gathers of the new weights follow bottom-up
  // behind the pass and are waited for layer by layer by the NEXT forward pass.
  const bool arena = overlap && dp_ranks_ > 1 && dp_shard_ && SetupArenas();
  if (arena) EnsureCorrectionArena();
  const int32 total_frames = in_err.NumRows() * dp_ranks_;
  dp_grad_items_.clear();
  // to be called before the back-propagation of layer i is queued
  auto err_ready = [&](int32 i, const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err_i) {
    if (!overlap || !nnet_[i]->IsUpdatable()) return;
    if (nnet_[i]->GetType() != Component::kBiasedLinearity) return;
    BiasedLinearity *bl = static_cast<BiasedLinearity *>(nnet_[i]);
    if (arena) {
      DpChunk &ch = dp_chunks_[dp_chunk_of_layer_[i]];
      size_t k = 0;
      while (ch.layers[k] != i) ++k;
      khAffineGradient item;
      bl->DescribeGradient(input, err_i, dp_g_arena_ + ch.offsets[k], &item);
      dp_grad_items_.push_back(item);
      // The errors of all the chunk's layers exist once that of its lowest layer does.  The top chunk's gradients
      // are launched here, in front of that layer's back-propagation -- the largest message of the step starts
      // as early as it can -- and are marked on their own; the other chunks' follow that back-propagation
      // (update_or_defer below) and share its mark: every mark in the main stream's queue holds the next launch
      // back by ~6 us (kernel trace).
      if (i == ch.layers.back() && &ch == &dp_chunks_[0]) {
        CuProfileScope p("BiasedLinearity::ComputeGradient");
        CU_SAFE_CALL(hipF_affine_gradient_group_marked(dev.MainStream(), static_cast<int32>(dp_grad_items_.size()), &dp_grad_items_[0],
                                                       ch.grad_ready));
        dp_grad_items_.clear();
      }
      return;
    }
    // (gather exchange with grouped updates: the one update launch runs behind the whole pass, in place)
    bl->SetDoubleBuffer((dp_ranks_ == 1 || dp_gather_) && !(dp_ranks_ > 1 && dp_gather_ && group_updates_));
    if (dp_ranks_ > 1) {
      // data-parallel: only the mark in the main stream; whoever needs it waits for it when its work is queued
      if (layer_events_[i] == NULL) CU_SAFE_CALL(hipK_event_create_ordering(&layer_events_[i]));
      CU_SAFE_CALL(hipK_event_record(layer_events_[i], dev.MainStream()));
    } else if (bl->UpdateKeepsOldWeights()) {
      dev.Order(dev.MainStream(), dev.SideStream(), &layer_events_[i]);
    }
  };
  // to be called after the back-propagation of layer i has been queued (data-parallel): "layer i's weights read"
  auto weights_read = [&](int32 i) {
    if (static_cast<int32>(read_events_.size()) < L) read_events_.resize(L, NULL);
    if (read_events_[i] == NULL) CU_SAFE_CALL(hipK_event_create_ordering(&read_events_[i]));
    CU_SAFE_CALL(hipK_event_record(read_events_[i], dev.MainStream()));
  };
  // Single GPU: the updates of the <biasedlinearity> layers are written down here and launched together behind
  // the back-propagation pass (hipF_affine_update_group).  Same result as the reference's order: a layer's
  // update reads its input, its output error and its own weights, all of which stay as they are until then, and
  // no back-propagation below reads what it writes.
  deferred_updates_.clear();
  deferred_layers_.clear();
  dp_first_gather_ = NULL;
  auto update = [&](int32 i, const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err) {
    if (!overlap || !nnet_[i]->IsUpdatable()) {
      if (dp_ranks_ == 1 && group_updates_ && nnet_[i]->IsUpdatable() && nnet_[i]->GetType() == Component::kBiasedLinearity &&
          !dev.Profiling()) {
        BiasedLinearity *bl = static_cast<BiasedLinearity *>(nnet_[i]);
        khAffineUpdate item;
        if (bl->GetLearnRate() > 0.0 && bl->DescribeUpdate(input, err, &item)) {
          deferred_updates_.push_back(item);
          deferred_layers_.push_back(i);
          return;
        }
      }
      UpdateLayer(i, input, err);
      return;
    }
    UpdatableComponent *uc = dynamic_cast<UpdatableComponent *>(nnet_[i]);
    if (!(uc->GetLearnRate() > 0.0)) return;
    side_used = true;
    BiasedLinearity *bl = (nnet_[i]->GetType() == Component::kBiasedLinearity) ? static_cast<BiasedLinearity *>(nnet_[i]) : NULL;
    if (bl != NULL && dp_ranks_ > 1 && dp_gather_) {
      // The rows of every rank, gathered on the third stream as soon as this layer's error exists;
      // then the ordinary fused update on all of them, on the side stream.
      if (static_cast<int32>(dp_in_all_.size()) < L) {
        dp_in_all_.resize(L);
        dp_err_all_.resize(L);
      }
      if (static_cast<int32>(grad_events_.size()) < L) grad_events_.resize(L, NULL);
      const int32 rows = input.NumRows();
      {
        CuStreamScope scope(dev.AuxStream());  // (a first-time Resize zero-fills on this stream)
        dp_in_all_[i].Resize(rows * dp_ranks_, input.NumCols());
        dp_err_all_[i].Resize(rows * dp_ranks_, err.NumCols());
      }
      KALDI_ASSERT(dp_err_all_[i].Stride() == err.Stride());
      CU_SAFE_CALL(hipK_stream_wait_event(dev.AuxStream(), layer_events_[i]));
      const BaseFloat *in_rows = input.Data();
      if (dp_in_all_[i].Stride() != input.Stride()) {
        // a caller's bunch with another row pitch (only the first layer's input can be one): repack first
        CuStreamScope scope(dev.AuxStream());
        dp_in_repack_.CopyFromMat(input);
        in_rows = dp_in_repack_.Data();
      }
      {
        CommGroupScope both;  // the two gathers of a layer go out as one launch of the collective library
        CU_SAFE_CALL(hipF_allgather(dev.AuxStream(), in_rows, dp_in_all_[i].Data(),
                                    static_cast<size_t>(rows) * dp_in_all_[i].Stride()));
        CU_SAFE_CALL(hipF_allgather(dev.AuxStream(), err.Data(), dp_err_all_[i].Data(),
                                    static_cast<size_t>(rows) * err.Stride()));
      }
      {
        // the updates on the gathered rows as grouped launches behind the pass (below), where the layer's update
        // is the single fused launch
        khAffineUpdate item;
        if (group_updates_ && bl->DescribeUpdate(dp_in_all_[i], dp_err_all_[i], &item)) {
          // the top layer's rows arrive first: a mark behind them lets its update start while the rows of the
          // layers below are still on the links
          if (deferred_updates_.empty()) {
            dev.Record(dev.AuxStream(), &grad_events_[i]);
            dp_first_gather_ = grad_events_[i];
          }
          deferred_updates_.push_back(item);
          deferred_layers_.push_back(i);
          aux_used = true;
          return;
        }
      }
      dev.Order(dev.AuxStream(), dev.SideStream(), &grad_events_[i]);
      // without a second weight buffer (L1 penalty) the step must also wait for this layer's back-propagation
      if (!bl->UpdateKeepsOldWeights()) CU_SAFE_CALL(hipK_stream_wait_event(dev.SideStream(), read_events_[i]));
      CuStreamScope scope(dev.SideStream());
      bl->UpdateWeights(dp_in_all_[i], dp_err_all_[i]);
      bl->UpdateBias(dp_err_all_[i]);
      aux_used = true;
      return;
    }
    if (bl != NULL && dp_ranks_ > 1) {
      // Three queues: the gradient GEMMs follow one another on the side stream, each layer's
      // all-reduce and shared step go to the third stream, so that communication of layer i runs
      // beside the gradient GEMMs and back-propagation of the layers below (a buffer per layer:
      // the next GEMM must not overwrite a gradient that is still being reduced).  The step
      // rewrites the weights in place, so it also waits for this layer's back-propagation.
      if (static_cast<int32>(dp_layer_grad_.size()) < L) {
        dp_layer_grad_.resize(L, NULL);
        dp_layer_grad_size_.resize(L, 0);
        grad_events_.resize(L, NULL);
      }
      const size_t need = bl->GradientSize();
      if (need > dp_layer_grad_size_[i]) {
        if (dp_layer_grad_[i]) {
          CU_SAFE_CALL(hipK_device_synchronize());
          CU_SAFE_CALL(hipK_free(dp_layer_grad_[i]));
        }
        void *p = NULL;
        CU_SAFE_CALL(hipK_malloc(&p, need * sizeof(BaseFloat)));
        dp_layer_grad_[i] = static_cast<BaseFloat *>(p);
        dp_layer_grad_size_[i] = need;
      }
      CU_SAFE_CALL(hipK_stream_wait_event(dev.SideStream(), layer_events_[i]));  // this layer's error exists
      {
        CuStreamScope scope(dev.SideStream());
        bl->ComputeGradient(input, err, dp_layer_grad_[i]);
      }
      dev.Order(dev.SideStream(), dev.AuxStream(), &grad_events_[i]);
      CU_SAFE_CALL(hipK_stream_wait_event(dev.AuxStream(), read_events_[i]));  // its back-propagation has read the weights
      {
        CuStreamScope scope(dev.AuxStream());
        CU_SAFE_CALL(hipF_allreduce_sum(dev.AuxStream(), dp_layer_grad_[i], need));
        bl->ApplyGradient(dp_layer_grad_[i], input.NumRows() * dp_ranks_);
      }
      aux_used = true;
      return;
    }
    if (bl != NULL && bl->UpdateKeepsOldWeights()) {
      // both halves on the side stream: the main stream's chain (back-propagation GEMM, its
      // slab sum) stays as short as possible, since the next layer's pair waits for it
      {
        CuStreamScope scope(dev.SideStream());
        bl->UpdateWeights(input, err);
      }
      // the bias half (a small column-sum kernel) on a stream of its own, behind the same
      // "error ready" point: it neither delays the next update GEMM nor the back-propagation chain
      CU_SAFE_CALL(hipK_stream_wait_event(dev.AuxStream(), layer_events_[i]));
      CuStreamScope scope(dev.AuxStream());
      bl->UpdateBias(err);
      aux_used = true;
      return;
    }
    dev.Order(dev.MainStream(), dev.SideStream(), &layer_events_[i]);
    CuStreamScope scope(dev.SideStream());
    uc->Update(input, err);
  };

  // Data-parallel: the whole back-propagation chain is queued first -- with one mark in front of and one behind
  // every layer's step -- and only then each layer's gradient, exchange and step on the other two streams,
  // which wait for those marks.  Queuing a layer's exchange takes the host 100 us and more (a dozen calls into
  // HIP and RCCL); done between the back-propagation launches it starved the main stream of work (timeline of
  // the one-GPU rehearsal: 90-150 us gaps between back-propagation kernels).
  struct PendingUpdate {
    int32 i;
    const CuMatrix<BaseFloat> *input, *err;
  };
  std::vector<PendingUpdate> pending;
  auto update_or_defer = [&](int32 i, const CuMatrix<BaseFloat> &input, const CuMatrix<BaseFloat> &err) {
    if (arena && nnet_[i]->IsUpdatable()) {
      // (SetupArenas has checked that every updatable component is a <biasedlinearity>)
      DpChunk &ch = dp_chunks_[dp_chunk_of_layer_[i]];
      if (i != ch.layers.back()) return;
      // The back-propagation of the chunk's lowest layer is queued: nothing on the main stream reads these
      // weights any more.  Exchange stream: sum of the gradient shards, then the step on this rank's shard.
      khStream_t xs = dev.AuxStream();
      const size_t per_rank = (ch.end - ch.begin) / dp_ranks_;
      const bool top = &ch == &dp_chunks_[0];
      if (top) {
        // gradients launched and marked in front of the back-propagation: a second mark behind it for the step
        // (none where no back-propagation followed: the bottom layer of a pass without an input gradient)
        const bool bp_followed = !(i == 0 && out_err == NULL);
        if (bp_followed) CU_SAFE_CALL(hipK_event_record(ch.weights_read, dev.MainStream()));
        CU_SAFE_CALL(hipK_stream_wait_event(xs, ch.grad_ready));
        CU_SAFE_CALL(hipF_reduce_scatter_sum(xs, dp_g_arena_ + ch.begin, per_rank));
        if (bp_followed) CU_SAFE_CALL(hipK_stream_wait_event(xs, ch.weights_read));
      } else {
        {
          CuProfileScope p("BiasedLinearity::ComputeGradient");
          CU_SAFE_CALL(hipF_affine_gradient_group_marked(dev.MainStream(), static_cast<int32>(dp_grad_items_.size()),
                                                         &dp_grad_items_[0], ch.grad_ready));
          dp_grad_items_.clear();
        }
        CU_SAFE_CALL(hipK_stream_wait_event(xs, ch.grad_ready));
        CU_SAFE_CALL(hipF_reduce_scatter_sum(xs, dp_g_arena_ + ch.begin, per_rank));
      }
      khSgdSegment segs[KH_MAX_SGD_SEGMENTS];
      int32 nseg = 0;
      for (size_t k = 0; k < ch.layers.size(); ++k) {  // (ascending offsets)
        BiasedLinearity *bl = static_cast<BiasedLinearity *>(nnet_[ch.layers[k]]);
        if (!(bl->GetLearnRate() > 0.0)) continue;  // (the reference skips the update of such a layer)
        bl->DescribeStep(ch.offsets[k], total_frames, &segs[nseg], &segs[nseg + 1]);
        nseg += 2;
      }
      size_t first = 0, count = 0;
      CU_SAFE_CALL(hipK_comm_shard_range(per_rank, &first, &count));
      CU_SAFE_CALL(hipF_sgd_apply_segments(xs, nseg, segs, ch.begin + first, count, dp_g_arena_, dp_w_arena_, dp_c_arena_));
      aux_used = true;
      return;
    }
    if (overlap && dp_ranks_ > 1 && nnet_[i]->IsUpdatable() && nnet_[i]->GetType() == Component::kBiasedLinearity) {
      weights_read(i);
      const PendingUpdate pu = {i, &input, &err};
      pending.push_back(pu);
      return;
    }
    update(i, input, err);
  };

  const CuMatrix<BaseFloat> *err = &in_err;  // error w.r.t. the output of layer i
  int32 i = L - 1;
  backpropagate_skipped_.assign(L > 1 ? L - 1 : 0, 0);
  while (i >= 0) {
    Component *comp = nnet_[i];
    // where the error w.r.t. this layer's input goes (NULL: nobody wants it)
    CuMatrix<BaseFloat> *below = (i > 0) ? &backpropagate_buf_[i - 1] : out_err;

    if (fusion_ && comp->GetType() == Component::kSoftmax && i > 0 && comp->InputDim() == comp->OutputDim()) {
      backpropagate_skipped_[i - 1] = 1;
      --i;  // identity on the error
      continue;
    }
    if (fusion_ && comp->GetType() == Component::kBiasedLinearity && i >= 1 &&
        nnet_[i - 1]->GetType() == Component::kSigmoid && nnet_[i - 1]->InputDim() == nnet_[i - 1]->OutputDim()) {
      // this layer and the sigmoid below it in one step; y = propagate_buf_[i] is that sigmoid's output
      BiasedLinearity *bl = static_cast<BiasedLinearity *>(comp);
      CuMatrix<BaseFloat> *below2 = (i - 1 > 0) ? &backpropagate_buf_[i - 2] : out_err;
      err_ready(i, propagate_buf_[i], *err);
      if (below2 != NULL) bl->BackpropagateFused(*err, propagate_buf_[i], below2);
      update_or_defer(i, propagate_buf_[i], *err);
      backpropagate_skipped_[i - 1] = 1;
      err = below2;
      i -= 2;
      continue;
    }
    err_ready(i, propagate_buf_[i], *err);
    if (below != NULL) comp->Backpropagate(*err, below);
    update_or_defer(i, propagate_buf_[i], *err);
    err = below;
    --i;
  }
  for (size_t k = 0; k < pending.size(); ++k) update(pending[k].i, *pending[k].input, *pending[k].err);
  if (!deferred_updates_.empty()) {
    CuProfileScope p("BiasedLinearity::Update");
    size_t first = 0;
    if (dp_ranks_ > 1) {
      // Data-parallel gather exchange: the rows of every rank have to be in first.  Where the top layer is a large
      // part of the work (the output layer of an acoustic model), its update is a launch of its own that only
      // waits for its own rows -- gathered first -- and runs beside the gathers of the layers below.
      double all = 0.0;
      for (size_t k = 0; k < deferred_updates_.size(); ++k) all += static_cast<double>(deferred_updates_[k].O) * deferred_updates_[k].I;
      const double top = static_cast<double>(deferred_updates_[0].O) * deferred_updates_[0].I;
      if (deferred_updates_.size() > 1 && dp_first_gather_ != NULL && top >= 0.25 * all) {
        CU_SAFE_CALL(hipK_stream_wait_event(dev.MainStream(), dp_first_gather_));
        CU_SAFE_CALL(hipF_affine_update_group(dev.Stream(), 1, &deferred_updates_[0]));
        first = 1;
      }
      dev.Order(dev.AuxStream(), dev.MainStream(), &aux_join_event_);
    }
    CU_SAFE_CALL(hipF_affine_update_group(dev.Stream(), static_cast<int32>(deferred_updates_.size() - first), &deferred_updates_[first]));
    for (size_t k = 0; k < deferred_layers_.size(); ++k)
      static_cast<BiasedLinearity *>(nnet_[deferred_layers_[k]])->AfterGroupedUpdate();
  }
  if (arena) {
    // The new weights, bottom-up: the next forward pass needs the lowest layers first and waits per chunk
    // (WaitForLayerWeights); nothing here is joined into the main stream.
    khStream_t xs = dev.AuxStream();
    for (size_t c = dp_chunks_.size(); c-- > 0;) {
      DpChunk &ch = dp_chunks_[c];
      CU_SAFE_CALL(hipF_allgather_shards(xs, dp_w_arena_ + ch.begin, (ch.end - ch.begin) / dp_ranks_));
      {
        CuStreamScope scope(xs);
        for (size_t k = 0; k < ch.layers.size(); ++k) static_cast<BiasedLinearity *>(nnet_[ch.layers[k]])->FinishShardedStep();
      }
      CU_SAFE_CALL(hipK_event_record(ch.weights_ready, xs));
      ch.pending = true;
    }
    if (side_used) dev.Order(dev.SideStream(), dev.MainStream(), &join_event_);
    return;
  }
  if (side_used) dev.Order(dev.SideStream(), dev.MainStream(), &join_event_);
  if (aux_used) dev.Order(dev.AuxStream(), dev.MainStream(), &aux_join_event_);
}

void Nnet::WaitForLayerWeights(int32 i) {
  if (dp_chunk_of_layer_.empty() || dp_chunk_of_layer_[i] < 0) return;
  DpChunk &ch = dp_chunks_[dp_chunk_of_layer_[i]];
  if (!ch.pending) return;
  CU_SAFE_CALL(hipK_stream_wait_event(CuDevice::Instantiate().MainStream(), ch.weights_ready));
  ch.pending = false;
}

void Nnet::FinishExchange() {
  for (size_t c = 0; c < dp_chunks_.size(); ++c) {
    if (!dp_chunks_[c].pending) continue;
    CU_SAFE_CALL(hipK_stream_wait_event(CuDevice::Instantiate().MainStream(), dp_chunks_[c].weights_ready));
    dp_chunks_[c].pending = false;
  }
}

bool Nnet::SetupArenas() {
  if (dp_w_arena_ != NULL) return true;
  if (dp_arena_failed_) return false;
  // the <biasedlinearity> layers, top-down; anything else that is updatable, or an L1 penalty, rules the arenas out
  std::vector<int32> layers;
  size_t total = 0;
  for (int32 i = LayerCount() - 1; i >= 0; --i) {
    if (!nnet_[i]->IsUpdatable()) continue;
    if (nnet_[i]->GetType() != Component::kBiasedLinearity || !static_cast<BiasedLinearity *>(nnet_[i])->CanUseArena()) {
      dp_arena_failed_ = true;
      return false;
    }
    layers.push_back(i);
    total += static_cast<BiasedLinearity *>(nnet_[i])->ParameterSpan();
  }
  if (layers.empty()) {
    dp_arena_failed_ = true;
    return false;
  }
  // Chunks of roughly equal size, cut top-down (the output layer of a large net is a chunk of its own): few
  // enough that the host queues a handful of collectives per step, enough that the first sums travel while the
  // back-propagation is still running.  KALDI_AMD_DP_CHUNKS overrides the number (1: one exchange behind the pass).
  int32 nchunks = 3;
  if (const char *e = getenv("KALDI_AMD_DP_CHUNKS")) nchunks = atoi(e);
  if (nchunks < 1) nchunks = 1;
  if (nchunks > static_cast<int32>(layers.size())) nchunks = layers.size();
  const size_t quantum = static_cast<size_t>(dp_ranks_) * 64;  // every shard a whole number of 256-byte lines
  dp_chunks_.clear();
  dp_chunk_of_layer_.assign(LayerCount(), -1);
  size_t pos = 0, in_chunk = 0;
  dp_chunks_.push_back(DpChunk());
  for (size_t k = 0; k < layers.size(); ++k) {
    BiasedLinearity *bl = static_cast<BiasedLinearity *>(nnet_[layers[k]]);
    if (in_chunk > 0 && in_chunk >= total / nchunks && static_cast<int32>(dp_chunks_.size()) < nchunks) {
      pos = (pos + quantum - 1) / quantum * quantum;
      dp_chunks_.back().end = pos;
      dp_chunks_.push_back(DpChunk());
      dp_chunks_.back().begin = pos;
      in_chunk = 0;
    }
    // (top-down in the pass, but ascending addresses bottom-up inside the chunk are not needed: any order does)
    dp_chunks_.back().layers.push_back(layers[k]);
    dp_chunks_.back().offsets.push_back(pos);
    dp_chunk_of_layer_[layers[k]] = dp_chunks_.size() - 1;
    pos += bl->ParameterSpan();
    in_chunk += bl->ParameterSpan();
  }
  pos = (pos + quantum - 1) / quantum * quantum;
  dp_chunks_.back().end = pos;
  dp_arena_floats_ = pos;
  for (size_t c = 0; c < dp_chunks_.size(); ++c) {
    KALDI_ASSERT(2 * dp_chunks_[c].layers.size() <= KH_MAX_SGD_SEGMENTS);
    CU_SAFE_CALL(hipK_event_create_ordering(&dp_chunks_[c].grad_ready));
    CU_SAFE_CALL(hipK_event_create_ordering(&dp_chunks_[c].weights_read));
    CU_SAFE_CALL(hipK_event_create_ordering(&dp_chunks_[c].weights_ready));
  }
  CuDevice &dev = CuDevice::Instantiate();
  void *p = NULL;
  CU_SAFE_CALL(hipK_malloc(&p, dp_arena_floats_ * sizeof(BaseFloat)));
  dp_w_arena_ = static_cast<BaseFloat *>(p);
  CU_SAFE_CALL(hipK_malloc(&p, dp_arena_floats_ * sizeof(BaseFloat)));
  dp_g_arena_ = static_cast<BaseFloat *>(p);
  // (the padding of rows, biases and chunks stays zero for good: gradient 0, weight 0)
  CU_SAFE_CALL(hipK_memset_async(dev.MainStream(), dp_w_arena_, 0, dp_arena_floats_ * sizeof(BaseFloat)));
  CU_SAFE_CALL(hipK_memset_async(dev.MainStream(), dp_g_arena_, 0, dp_arena_floats_ * sizeof(BaseFloat)));
  CuStreamScope scope(dev.MainStream());
  for (size_t c = 0; c < dp_chunks_.size(); ++c)
    for (size_t k = 0; k < dp_chunks_[c].layers.size(); ++k)
      static_cast<BiasedLinearity *>(nnet_[dp_chunks_[c].layers[k]])->MoveParametersTo(dp_w_arena_ + dp_chunks_[c].offsets[k]);
  return true;
}

void Nnet::EnsureCorrectionArena() {
  if (dp_c_arena_ != NULL || dp_w_arena_ == NULL) return;
  bool need = false;
  for (size_t c = 0; c < dp_chunks_.size(); ++c)
    for (size_t k = 0; k < dp_chunks_[c].layers.size(); ++k)
      if (static_cast<BiasedLinearity *>(nnet_[dp_chunks_[c].layers[k]])->NeedsCorrections()) need = true;
  if (!need) return;
  CuDevice &dev = CuDevice::Instantiate();
  void *p = NULL;
  CU_SAFE_CALL(hipK_malloc(&p, dp_arena_floats_ * sizeof(BaseFloat)));
  dp_c_arena_ = static_cast<BaseFloat *>(p);
  CU_SAFE_CALL(hipK_memset_async(dev.MainStream(), dp_c_arena_, 0, dp_arena_floats_ * sizeof(BaseFloat)));
  CuStreamScope scope(dev.MainStream());
  for (size_t c = 0; c < dp_chunks_.size(); ++c)
    for (size_t k = 0; k < dp_chunks_[c].layers.size(); ++k)
      static_cast<BiasedLinearity *>(nnet_[dp_chunks_[c].layers[k]])->MoveCorrectionsTo(dp_c_arena_ + dp_chunks_[c].offsets[k]);
  // the exchange stream uses the arena from the next step on
  dev.Order(dev.MainStream(), dev.AuxStream(), &aux_join_event_);
}

void Nnet::ReleaseArenas() {
  if (dp_w_arena_ == NULL) {
    dp_arena_failed_ = false;
    return;
  }
  FinishExchange();
  CU_SAFE_CALL(hipK_device_synchronize());
  for (size_t c = 0; c < dp_chunks_.size(); ++c) {
    for (size_t k = 0; k < dp_chunks_[c].layers.size(); ++k)
      static_cast<BiasedLinearity *>(nnet_[dp_chunks_[c].layers[k]])->OwnParametersAgain();
    hipK_event_destroy(dp_chunks_[c].grad_ready);
    hipK_event_destroy(dp_chunks_[c].weights_read);
    hipK_event_destroy(dp_chunks_[c].weights_ready);
  }
  CU_SAFE_CALL(hipK_device_synchronize());
  dp_chunks_.clear();
  dp_chunk_of_layer_.clear();
  CU_SAFE_CALL(hipK_free(dp_w_arena_));
  CU_SAFE_CALL(hipK_free(dp_g_arena_));
  if (dp_c_arena_) CU_SAFE_CALL(hipK_free(dp_c_arena_));
  dp_w_arena_ = dp_g_arena_ = dp_c_arena_ = NULL;
  dp_arena_floats_ = 0;
  dp_arena_failed_ = false;
}

void Nnet::Feedforward(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) {
  KALDI_ASSERT(NULL != out);
  const int32 L = LayerCount();
  if (L == 0) {
    out->CopyFromMat(in);
    return;
  }
  KALDI_ASSERT(propagate_buf_.size() >= 2 || L == 1);
  FinishExchange();
  // steps: each is one component or a fused (linearity, activation) pair; intermediate results
  // alternate between the first two buffers, the last step writes into *out
  const CuMatrix<BaseFloat> *src = &in;
  int32 flip = 0;
  for (int32 i = 0; i < L;) {
    const int act = fusion_ ? ActivationAfter(nnet_, i) : -1;
    const int32 next = i + (act >= 0 ? 2 : 1);
    CuMatrix<BaseFloat> *dst = (next >= L) ? out : &propagate_buf_[flip];
    if (act >= 0) static_cast<BiasedLinearity *>(nnet_[i])->PropagateFused(*src, act, dst);
    else nnet_[i]->Propagate(*src, dst);
    src = dst;
    flip ^= 1;
    i = next;
  }
}

void Nnet::Read(const std::string &file) {
  bool binary;
  Input in(file, &binary);
  Read(in.Stream(), binary);
  in.Close();
}

void Nnet::Read(std::istream &in, bool binary) {
  Component *comp;
  while (NULL != (comp = Component::Read(in, binary, this))) {
    if (LayerCount() > 0 && nnet_.back()->OutputDim() != comp->InputDim()) {
      KALDI_ERR << "Dimensionality mismatch!"
                << " Previous layer output:" << nnet_.back()->OutputDim() << " Current layer input:" << comp->InputDim();
    }
    nnet_.push_back(comp);
  }
  // A stream without a component ends the reference too: it sizes its backward buffers to LayerCount() - 1
  // (nnet/nnet-nnet.cc:159), which throws std::length_error for -1.
  if (LayerCount() == 0) KALDI_ERR << "No component in the model stream";
  propagate_buf_.resize(LayerCount() + 1);
  backpropagate_buf_.resize(LayerCount() - 1);
  learn_rate_ = 0.0;
}

void Nnet::Write(const std::string &file, bool binary) {
  Output out(file, binary, true);
  Write(out.Stream(), binary);
  out.Close();
}

void Nnet::Write(std::ostream &out, bool binary) {
  FinishExchange();
  for (int32 i = 0; i < LayerCount(); ++i) nnet_[i]->Write(out, binary);
}

void Nnet::WriteFrontLayers(std::ostream &out, bool binary, int32 num_layers) {
  KALDI_ASSERT(num_layers <= LayerCount());
  FinishExchange();
  for (int32 i = 0; i < num_layers; ++i) nnet_[i]->Write(out, binary);
}

void Nnet::SetLearnRate(BaseFloat lrate, const char *lrate_factors) {
  std::vector<BaseFloat> factors;
  if (NULL != lrate_factors) {
    std::string s(lrate_factors);
    size_t pos = 0;
    while ((pos = s.find_first_not_of(",:; ", pos)) != std::string::npos) {
      const size_t end = s.find_first_of(",:; ", pos);
      factors.push_back(atof(s.substr(pos, end == std::string::npos ? std::string::npos : end - pos).c_str()));
      if (end == std::string::npos) break;
      pos = end;
    }
  }
  int32 updatable = 0;
  for (int32 i = 0; i < LayerCount(); ++i)
    if (nnet_[i]->IsUpdatable()) ++updatable;
  if (!factors.empty() && updatable != static_cast<int32>(factors.size()))
    KALDI_ERR << "Mismatch between number of trainable layers " << updatable << " and learn rate factors "
              << factors.size();
  int32 k = 0;
  for (int32 i = 0; i < LayerCount(); ++i) {
    if (!nnet_[i]->IsUpdatable()) continue;
    BaseFloat scaled = lrate;
    if (!factors.empty()) scaled *= factors[k++];
    dynamic_cast<UpdatableComponent *>(nnet_[i])->SetLearnRate(scaled);
  }
  learn_rate_ = lrate;
}

std::string Nnet::GetLearnRateString() {
  std::ostringstream oss;
  oss << "LEARN_RATE global: " << learn_rate_ << " individual: ";
  for (int32 i = 0; i < LayerCount(); ++i)
    if (nnet_[i]->IsUpdatable()) oss << dynamic_cast<UpdatableComponent *>(nnet_[i])->GetLearnRate() << " ";
  return oss.str();
}

#define KALDI_AMD_FOR_UPDATABLE(call)                                                  \
  for (int32 i = 0; i < LayerCount(); ++i)                                             \
    if (nnet_[i]->IsUpdatable()) dynamic_cast<UpdatableComponent *>(nnet_[i])->call

void Nnet::SetMomentum(BaseFloat mmt) { KALDI_AMD_FOR_UPDATABLE(SetMomentum(mmt)); }
void Nnet::SetConstantMomentum(BaseFloat mmt) {
  SetMomentum(mmt);
  for (int32 i = 0; i < LayerCount(); ++i)
    if (nnet_[i]->GetType() == Component::kBiasedLinearity) static_cast<BiasedLinearity *>(nnet_[i])->SetKeepGradient(false);
}
void Nnet::SetL2Penalty(BaseFloat l2) { KALDI_AMD_FOR_UPDATABLE(SetL2Penalty(l2)); }
void Nnet::SetL2UpperBound(BaseFloat bound) { KALDI_AMD_FOR_UPDATABLE(SetL2UpperBound(bound)); }
void Nnet::SetL1Penalty(BaseFloat l1) { KALDI_AMD_FOR_UPDATABLE(SetL1Penalty(l1)); }
void Nnet::SetAverageGrad(bool flag) { KALDI_AMD_FOR_UPDATABLE(SetAverageGrad(flag)); }
#undef KALDI_AMD_FOR_UPDATABLE

void Nnet::SetDataParallel(int32 nranks) {
  KALDI_ASSERT(nranks >= 1);
  ReleaseArenas();
  dp_ranks_ = nranks;
  // Three exchanges give the same step.
  //  shard:  (the default) the parameters of the <biasedlinearity> layers live in arenas (SetupArenas), cut into a
  //          few chunks of layers.  Per chunk: the gradients in one launch, a reduce-scatter of the chunk's
  //          gradient memory, the step on this rank's 1/n of the chunk, and -- bottom-up behind the pass -- an
  //          all-gather of the new parameters in place, for which the next forward pass waits layer by layer.
  //          2(n-1)/n x 185 MB over the links per step of the Switchboard net, as for an all-reduce, but the
  //          element-wise pass over gradient and weights is 1/n of the net per rank and the exchange runs beside
  //          the back-propagation and the next forward pass.
  //  gather: every rank's rows of (layer input, layer error) are gathered -- (n-1) x 40 MB per rank and step on
  //          the Switchboard net at a 256-frame bunch -- and each rank runs the single-GPU fused update on the
  //          n x 256 rows (n times the update FLOPs, no gradient buffers, replicas identical by construction).
  //  sum:    all-reduce of each layer's gradient, the whole step on every rank (also what a network takes that
  //          cannot use the arenas: an L1 penalty, an updatable component of another kind).
  // Default by the number of ranks, from the bytes a link carries per step (xGMI is point to point: n ranks of one
  // node have one link per pair).  Switchboard net, 256-frame bunch: gather sends 40 MB to every peer whatever n is;
  // shard and sum move 2/n x 185 MB over every link -- 185 MB for 2 ranks, 92 for 4, 46 for 8.  Gather pays with n
  // times the update FLOPs (a third of the step each), so it wins while the links are the bound: 2 to 4 ranks.
  dp_gather_ = nranks >= 2 && nranks <= 4;
  dp_shard_ = nranks > 4;
  if (const char *e = getenv("KALDI_AMD_DP_EXCHANGE")) {
    dp_gather_ = nranks >= 2 && std::string(e) == "gather";
    dp_shard_ = nranks >= 2 && std::string(e) == "shard";
  }
}

// ---- activations that look their buffers up in the owning network ------------------------------
// (through BufferInFrontOfLayer, not PropagateBuffer(): a caller's look at the buffers turns the fused passes off,
// the network's own components must not; what they read -- their own output, or for <softrelu> its input -- is
// written by every forward pass, fused or not)
void Sigmoid::BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
  const CuMatrix<BaseFloat> &y = nnet_->BufferInFrontOfLayer(nnet_->IndexOfLayer(*this) + 1);
  cu::DiffSigmoid(in_err, y, out_err);
}
void Relu::BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
  const CuMatrix<BaseFloat> &y = nnet_->BufferInFrontOfLayer(nnet_->IndexOfLayer(*this) + 1);
  cu::DiffRelu(in_err, y, out_err);
}
void SoftRelu::BackpropagateFnc(const CuMatrix<BaseFloat> &in_err, CuMatrix<BaseFloat> *out_err) {
  const CuMatrix<BaseFloat> &x = nnet_->BufferInFrontOfLayer(nnet_->IndexOfLayer(*this));
  cu::DiffSoftRelu(in_err, x, out_err);
}

}  // namespace kaldi
