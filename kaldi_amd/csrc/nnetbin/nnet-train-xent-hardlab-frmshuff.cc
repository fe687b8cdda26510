// nnet-train-xent-hardlab-frmshuff: one epoch of frame-shuffled minibatch SGD with the
// cross-entropy objective and hard (alignment) labels.  Command line, defaults, log lines and
// exit codes of the reference tool (nnetbin/nnet-train-xent-hardlab-frmshuff.cc:27-209); the
// multi-GPU options (--num-ranks ...) are additions: synchronous data-parallel training, each
// rank on its shard of the utterances, gradients summed with an RCCL all-reduce per layer.
//
// Command line, option names, usage text and log lines are derived from nnetbin/nnet-train-xent-hardlab-frmshuff.cc of troylee/kaldi
// (Copyright 2011 Karel Vesely), licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); this file is distributed under the same license, "AS IS",
// without warranties or conditions of any kind.
#include "base/kaldi-common.h"
#include "cudamatrix/cu-device.h"
#include "nnet/nnet-cache.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-nnet.h"
#include "nnetbin/dist-utils.h"
#include "nnetbin/feature-upload.h"
#include "util/common-utils.h"
#include "util/timer.h"

int main(int argc, char *argv[]) {
  using namespace kaldi;
  try {
    const char *usage =
        "Perform iteration of Neural Network training by stochastic gradient descent.\n"
        "Usage:  nnet-train-xent-hardlab-frmshuff [options] <feature-rspecifier> <alignments-rspecifier> <model-in> "
        "[<model-out>]\n"
        "e.g.: \n"
        " nnet-train-xent-hardlab-perutt scp:train.scp ark:train.ali nnet.init nnet.iter1\n";

    ParseOptions po(usage);
    bool binary = false, crossvalidate = false, randomize = true;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    po.Register("randomize", &randomize, "Perform the frame-level shuffling within the Cache::");

    BaseFloat learn_rate = 0.008, momentum = 0.0, l2_penalty = 0.0, l2_upper_bound = 0.0, l1_penalty = 0.0;
    po.Register("learn-rate", &learn_rate, "Learning rate");
    po.Register("momentum", &momentum, "Momentum");
    po.Register("l2-penalty", &l2_penalty, "L2 penalty (weight decay)");
    po.Register("l2-upper-bound", &l2_upper_bound, "L2 upper bound (>=1.0)");
    po.Register("l1-penalty", &l1_penalty, "L1 penalty (promote sparsity)");

    bool average_grad = false;
    po.Register("average-grad", &average_grad, "Whether to average the gradient in the bunch");
    std::string learn_factors = "";
    po.Register("learn-factors", &learn_factors,
                "Learning factor for each updatable layer, separated by ',' and work together with learn rate");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform Neural Network");

    int32 bunchsize = 512, cachesize = 32768;
    po.Register("bunchsize", &bunchsize, "Size of weight update block");
    po.Register("cachesize", &cachesize, "Size of cache for frame level shuffling");

    bool fused = true, profile = false;
    po.Register("fused", &fused, "Run the linearity+activation pairs as fused GPU kernels");
    po.Register("profile", &profile, "Synchronise after every GPU call and print a per-function time profile");
    DistOptions dist;
    dist.Register(&po);

    po.Read(argc, argv);

    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) {
      po.PrintUsage();
      exit(1);
    }

    std::string feature_rspecifier = po.GetArg(1), alignments_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);

    dist.Resolve();
    DistInit(dist);
    CuDevice::Instantiate().Profile(profile);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);

    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetLearnRate(learn_rate, learn_factors == "" ? NULL : learn_factors.c_str());
    nnet.SetConstantMomentum(momentum);
    nnet.SetL2Penalty(l2_penalty);
    nnet.SetL2UpperBound(l2_upper_bound);
    nnet.SetL1Penalty(l1_penalty);
    nnet.SetAverageGrad(average_grad);
    nnet.SetFusion(fused);
    // fused mode also drops the per-bunch copies: bunches are windows onto the cache and the
    // network reads its input in place (both stay untouched until Backpropagate has returned)
    nnet.SetBorrowInput(fused);
    nnet_transf.SetFusion(fused);
    if (dist.num_ranks > 1 && !crossvalidate) nnet.SetDataParallel(dist.num_ranks);

    kaldi::int64 tot_t = 0;

    // Every rank walks the same feature table and keeps the utterances the assigner gives it (the rank with the
    // fewest frames so far), decided on each entry's header: a binary matrix that belongs to another rank is
    // skipped unread.  The reader reads ahead on a thread of its own; the assigner is only used there.
    ShardAssigner shard(dist.num_ranks);
    const int32 my_rank = dist.rank;
    PrefetchingBaseFloatMatrixReader feature_reader(
        feature_rspecifier, [&shard, my_rank](const std::string &, int32 rows) { return shard.Assign(rows) == my_rank; });
    RandomAccessInt32VectorReader alignments_reader(alignments_rspecifier);

    Cache cache;
    // (the reference divides by it first and dies of the division, nnetbin/nnet-train-xent-hardlab-frmshuff.cc:115)
    if (bunchsize <= 0) KALDI_ERR << "--bunchsize must be positive, got " << bunchsize;
    cachesize = (cachesize / bunchsize) * bunchsize;  // a whole number of bunches
    cache.Init(cachesize, bunchsize);

    Xent xent;

    CuMatrix<BaseFloat> feats_transf, nnet_in, nnet_out, glob_err;
    FeatureUploader uploader(cachesize);
    CuStlVector<int32> targets;

    Timer tim;
    double time_next = 0;
    KALDI_LOG << (crossvalidate ? "CROSSVALIDATE" : "TRAINING") << " STARTED";

    int32 num_done = 0, num_no_alignment = 0, num_other_error = 0, num_cache = 0;
    while (1) {
      // fill the cache with this rank's utterances; with several ranks an error here is told to the others
      // below instead of leaving them waiting in a collective for a rank that has gone
      std::string fill_error;
      try {
        while (!cache.Full() && !feature_reader.Done()) {
          std::string key = feature_reader.Key();
          if (!alignments_reader.HasKey(key)) {
            num_no_alignment++;
          } else {
            const Matrix<BaseFloat> &mat = feature_reader.Value();
            const std::vector<int32> &alignment = alignments_reader.Value(key);
            if ((int32)alignment.size() != mat.NumRows()) {
              KALDI_WARN << "Alignment has wrong size " << (alignment.size()) << " vs. " << (mat.NumRows());
              num_other_error++;
              // step to the next utterance (the reference retried the same one forever)
            } else {
              for (size_t i = 0; i < alignment.size(); ++i)
                if (alignment[i] < 0 || alignment[i] >= nnet.OutputDim())
                  KALDI_ERR << "Alignment of " << key << " has label " << alignment[i] << " outside the network's "
                            << nnet.OutputDim() << " outputs";
              const CuMatrix<BaseFloat> &feats = uploader.Upload(mat);  // host -> device, beside the training
              nnet_transf.Feedforward(feats, &feats_transf);     // optional transform
              cache.AddData(feats_transf, alignment);
              num_done++;
            }
          }
          Timer t_features;
          feature_reader.Next();
          time_next += t_features.Elapsed();
        }
      } catch (const std::exception &e) {
        if (dist.num_ranks == 1) throw;
        fill_error = e.what();
      }
      // (Like the reference, also a last cache that holds less than one bunch is shuffled -- its line says [RND] --
      // but a rank whose shard has run dry holds nothing to shuffle: it takes part in the collectives with 0 bunches.)
      if (!crossvalidate && randomize && fill_error.empty() && cache.HoldsRows()) cache.Randomize();
      // Nothing to train on at all (an empty table, no utterance with an alignment): the reference stops in
      // Cache::Randomize's assertion (nnet-cache.cc:141); a run that "succeeds" with a nan accuracy helps nobody.
      // (A cache left empty BEHIND trained ones -- trailing utterances without alignments -- ends the run normally.)
      if (dist.num_ranks == 1 && !crossvalidate && randomize && num_done == 0 && !cache.HoldsRows())
        KALDI_ERR << "No utterance to train on: the feature table is empty or no utterance has an alignment";
      std::cerr << "Cache #" << ++num_cache << " " << (cache.Randomized() ? "[RND]" : "[NO-RND]")
                << " segments: " << num_done << " frames: " << tot_t << "\n";

      if (dist.num_ranks == 1) {
        while (!cache.Empty()) {
          if (fused) cache.GetBunchView(&nnet_in, &targets);
          else cache.GetBunch(&nnet_in, &targets);
          nnet.PropagateXent(nnet_in, targets, &xent, &nnet_out, &glob_err, false);  // nothing reads the posteriors here
          if (!crossvalidate) nnet.Backpropagate(glob_err, NULL);
          tot_t += nnet_in.NumRows();
        }
        if (feature_reader.Done()) break;
      } else {
        // every rank must take the same number of steps: agree once per cache on the bunch count (the
        // minimum over the ranks; -1 from a rank that failed makes every rank stop with an error)
        const int64 bunches = DistAllReduce(dist, fill_error.empty() ? cache.NumBunches() : -1, 1 /*min*/);
        if (bunches < 0)
          KALDI_ERR << (fill_error.empty() ? std::string("Another rank failed while reading its data; stopping")
                                           : "Rank " + std::to_string(dist.rank) + " failed: " + fill_error);
        for (int64 b = 0; b < bunches; ++b) {
          if (fused) cache.GetBunchView(&nnet_in, &targets);
          else cache.GetBunch(&nnet_in, &targets);
          nnet.PropagateXent(nnet_in, targets, &xent, &nnet_out, &glob_err, false);  // nothing reads the posteriors here
          if (!crossvalidate) nnet.Backpropagate(glob_err, NULL);
          tot_t += nnet_in.NumRows();
        }
        // bunches this rank still holds when another rank had fewer are dropped (last cache only: full caches
        // hold the same number everywhere), as a single process drops its last partial bunch
        while (!cache.Empty()) cache.GetBunch(&nnet_in, &targets);
        // Stop once any rank's shard has run dry: shards are balanced by frames, so the others hold at most a few
        // utterances more, which no round of equal steps could use.
        if (DistAllReduce(dist, feature_reader.Done() ? 1 : 0, 2 /*max*/)) break;
      }
    }

    CuDevice::Instantiate().Synchronize();
    if (!crossvalidate && dist.rank == 0) nnet.Write(target_model_filename, binary);

    std::cout << "\n" << std::flush;

    KALDI_LOG << (crossvalidate ? "CROSSVALIDATE" : "TRAINING") << " FINISHED " << tim.Elapsed() << "s, fps"
              << tot_t / tim.Elapsed() << ", feature wait " << time_next << "s";
    KALDI_LOG << "Done " << num_done << " files, " << num_no_alignment << " with no alignments, " << num_other_error
              << " with other errors.";
    if (dist.num_ranks == 1) {
      KALDI_LOG << xent.Report();
    } else {
      // One report for the whole job, on rank 0, in the reference's format (the recipes' scheduler takes the
      // accuracy from the FRAME_ACCURACY line): frames, correct frames and loss summed over the ranks.  The other
      // ranks log their shard without that keyword.
      int64 frames, correct;
      double loss;
      xent.GetStats(&frames, &correct, &loss);
      const int64 all_frames = DistAllReduce(dist, frames, 0), all_correct = DistAllReduce(dist, correct, 0);
      const double all_loss = DistAllReduceSum(dist, loss);
      KALDI_LOG << "rank " << dist.rank << " of " << dist.num_ranks << ": shard frames:" << frames << " correct:" << correct
                << " xent:" << loss << ", feature payload read " << feature_reader.PayloadBytesRead() << " bytes";
      if (dist.rank == 0) {
        KALDI_LOG << "all " << dist.num_ranks << " ranks: fps " << all_frames / tim.Elapsed();
        KALDI_LOG << Xent::FormatReport(all_loss, all_frames, all_correct);
      }
    }
    CuDevice::Instantiate().PrintProfile();
    DistFinalize(dist);
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
