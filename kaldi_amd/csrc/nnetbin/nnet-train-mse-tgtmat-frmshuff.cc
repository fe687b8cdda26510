// nnet-train-mse-tgtmat-frmshuff: one epoch of frame-shuffled minibatch SGD with the mean-square
// error objective and a matrix of targets per utterance (autoencoders, regression front ends).
// Command line, defaults and log lines of the reference tool
// (nnetbin/nnet-train-mse-tgtmat-frmshuff.cc:26-222); shares every kernel with the
// cross-entropy trainer (Mse::Eval instead of Xent::EvalVec).  A feature / target pair with
// different lengths is skipped (the reference never advanced past it).
//
// Command line, option names, usage text and log lines are derived from nnetbin/nnet-train-mse-tgtmat-frmshuff.cc of troylee/kaldi
// (Copyright 2012 Karel Vesely), licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); this file is distributed under the same license, "AS IS",
// without warranties or conditions of any kind.
#include "base/kaldi-common.h"
#include "cudamatrix/cu-device.h"
#include "nnet/nnet-cache.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-nnet.h"
#include "util/common-utils.h"
#include "util/timer.h"

int main(int argc, char *argv[]) {
  using namespace kaldi;
  try {
    const char *usage =
        "Perform iteration of Neural Network training by stochastic gradient descent.\n"
        "Usage:  nnet-train-mse-tgtmat-frmshuff [options] <feature-rspecifier> <targets-rspecifier> <model-in> "
        "[<model-out>]\n"
        "e.g.: \n"
        " nnet-train-mse-tgtmat-frmshuff scp:train.scp ark:targets.scp nnet.init nnet.iter1\n";

    ParseOptions po(usage);
    bool binary = false, crossvalidate = false;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    BaseFloat learn_rate = 0.008, momentum = 0.0, l2_penalty = 0.0, l1_penalty = 0.0;
    po.Register("learn-rate", &learn_rate, "Learning rate");
    po.Register("momentum", &momentum, "Momentum");
    po.Register("l2-penalty", &l2_penalty, "L2 penalty (weight decay)");
    po.Register("l1-penalty", &l1_penalty, "L1 penalty (promote sparsity)");
    std::string learn_factors = "";
    po.Register("learn-factors", &learn_factors,
                "Learning factor for each updatable layer, separated by ',' and work together with learn rate");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform Neural Network");
    int32 bunchsize = 512, cachesize = 32768;
    po.Register("bunchsize", &bunchsize, "Size of weight update block");
    po.Register("cachesize", &cachesize, "Size of cache for frame level shuffling");
    bool average_grad = false;
    po.Register("average-grad", &average_grad, "Whether to average the gradient in the bunch");
    bool fused = true, randomize = true;
    int32 gpu_id = -1;
    po.Register("fused", &fused, "Run the linearity+activation pairs as fused GPU kernels");
    po.Register("randomize", &randomize, "Perform the frame-level shuffling within the cache");
    po.Register("gpu-id", &gpu_id, "GPU ordinal (-1: the GPU with most free memory)");
    po.Read(argc, argv);

    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) {
      po.PrintUsage();
      exit(1);
    }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);

    CuDevice::Instantiate().SelectGpuId(gpu_id);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetLearnRate(learn_rate, learn_factors == "" ? NULL : learn_factors.c_str());
    nnet.SetConstantMomentum(momentum);
    nnet.SetL2Penalty(l2_penalty);
    nnet.SetL1Penalty(l1_penalty);
    nnet.SetAverageGrad(average_grad);
    nnet.SetFusion(fused);
    nnet_transf.SetFusion(fused);

    kaldi::int64 tot_t = 0;
    // both tables are read in step: the lists must be in the same order
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    SequentialBaseFloatMatrixReader targets_reader(targets_rspecifier);

    CacheTgtMat cache;
    // (the reference divides by it first and dies of the division, nnetbin/nnet-train-xent-hardlab-frmshuff.cc:115)
    if (bunchsize <= 0) KALDI_ERR << "--bunchsize must be positive, got " << bunchsize;
    cachesize = (cachesize / bunchsize) * bunchsize;
    cache.Init(cachesize, bunchsize);
    Mse mse;
    CuMatrix<BaseFloat> feats, feats_transf, targets, nnet_in, nnet_out, nnet_tgt, glob_err;

    Timer tim;
    double time_next = 0;
    KALDI_LOG << (crossvalidate ? "CROSSVALIDATE" : "TRAINING") << " STARTED";

    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0, num_cache = 0;
    while (1) {
      while (!cache.Full() && !feature_reader.Done() && !targets_reader.Done()) {
        std::string tgt_key = targets_reader.Key();
        std::string fea_key = feature_reader.Key();
        // skip feature matrices without targets
        while (fea_key != tgt_key && !feature_reader.Done()) {
          KALDI_WARN << "No targets for: " << fea_key;
          num_no_tgt_mat++;
          feature_reader.Next();
          if (!feature_reader.Done()) fea_key = feature_reader.Key();
        }
        // (targets whose features never come: the reference asks the exhausted feature reader for its key here and
        // dies of that, nnetbin/nnet-train-mse-tgtmat-frmshuff.cc:136-142, util/kaldi-table-inl.h:423)
        if (feature_reader.Done())
          KALDI_ERR << "Key() called on TableReader object at the wrong time: no features for the targets of " << tgt_key;
        const Matrix<BaseFloat> &fea_mat = feature_reader.Value();
        const Matrix<BaseFloat> &tgt_mat = targets_reader.Value();
        if (tgt_mat.NumRows() != fea_mat.NumRows()) {
          KALDI_WARN << "Alignment has wrong size " << (tgt_mat.NumRows()) << " vs. " << (fea_mat.NumRows());
          num_other_error++;
        } else {
          feats.CopyFromMat(fea_mat);
          targets.CopyFromMat(tgt_mat);
          nnet_transf.Feedforward(feats, &feats_transf);
          cache.AddData(feats_transf, targets);
          num_done++;
        }
        Timer t_features;
        feature_reader.Next();
        targets_reader.Next();
        time_next += t_features.Elapsed();
      }
      if (!crossvalidate && randomize) cache.Randomize();
      std::cerr << "Cache #" << ++num_cache << " " << (cache.Randomized() ? "[RND]" : "[NO-RND]")
                << " segments: " << num_done << " frames: " << tot_t << "\n";
      while (!cache.Empty()) {
        cache.GetBunch(&nnet_in, &nnet_tgt);
        nnet.Propagate(nnet_in, &nnet_out);
        mse.Eval(nnet_out, nnet_tgt, &glob_err);
        if (!crossvalidate) nnet.Backpropagate(glob_err, NULL);
        tot_t += nnet_in.NumRows();
      }
      if (feature_reader.Done() || targets_reader.Done()) break;
    }

    CuDevice::Instantiate().Synchronize();
    if (!crossvalidate) nnet.Write(target_model_filename, binary);
    std::cout << "\n" << std::flush;

    KALDI_LOG << (crossvalidate ? "CROSSVALIDATE" : "TRAINING") << " FINISHED " << tim.Elapsed() << "s, fps"
              << tot_t / tim.Elapsed() << ", feature wait " << time_next << "s";
    KALDI_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no tgt_mats, " << num_other_error
              << " with other errors.";
    KALDI_LOG << mse.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
