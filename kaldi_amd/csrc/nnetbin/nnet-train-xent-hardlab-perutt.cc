// nnet-train-xent-hardlab-perutt: SGD with one update per utterance (no frame shuffling), the
// cross-entropy objective and hard labels.  Command line, defaults and log lines of the reference
// tool (nnetbin/nnet-train-xent-hardlab-perutt.cc:26-170); it shares every kernel with the
// frame-shuffled trainer.  An utterance whose alignment has the wrong length is skipped (the
// reference never advanced past it).
//
// Command line, option names, usage text and log lines are derived from nnetbin/nnet-train-xent-hardlab-perutt.cc of troylee/kaldi
// (Copyright 2011 Karel Vesely), licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); this file is distributed under the same license, "AS IS",
// without warranties or conditions of any kind.
#include "base/kaldi-common.h"
#include "cudamatrix/cu-device.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-nnet.h"
#include "util/common-utils.h"
#include "util/timer.h"

int main(int argc, char *argv[]) {
  using namespace kaldi;
  try {
    const char *usage =
        "Perform iteration of Neural Network training by stochastic gradient descent.\n"
        "Usage:  nnet-train-xent-hardlab-perutt [options] <feature-rspecifier> <alignments-rspecifier> <model-in> "
        "[<model-out>]\n"
        "e.g.: \n"
        " nnet-train-xent-hardlab-perutt scp:train.scp ark:train.ali nnet.init nnet.iter1\n";

    ParseOptions po(usage);
    bool binary = false, crossvalidate = false;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    BaseFloat learn_rate = 0.008, momentum = 0.0, l2_penalty = 0.0, l1_penalty = 0.0;
    po.Register("learn-rate", &learn_rate, "Learning rate");
    po.Register("momentum", &momentum, "Momentum");
    po.Register("l2-penalty", &l2_penalty, "L2 penalty (weight decay)");
    po.Register("l1-penalty", &l1_penalty, "L1 penalty (promote sparsity)");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform Neural Network");
    bool fused = true;
    int32 gpu_id = -1;
    po.Register("fused", &fused, "Run the linearity+activation pairs as fused GPU kernels");
    po.Register("gpu-id", &gpu_id, "GPU ordinal (-1: the GPU with most free memory)");
    po.Read(argc, argv);

    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) {
      po.PrintUsage();
      exit(1);
    }
    std::string feature_rspecifier = po.GetArg(1), alignments_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);

    CuDevice::Instantiate().SelectGpuId(gpu_id);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetLearnRate(learn_rate, NULL);
    nnet.SetConstantMomentum(momentum);
    nnet.SetL2Penalty(l2_penalty);
    nnet.SetL1Penalty(l1_penalty);
    nnet.SetFusion(fused);
    nnet_transf.SetFusion(fused);

    kaldi::int64 tot_t = 0;
    PrefetchingBaseFloatMatrixReader feature_reader(feature_rspecifier);  // reads ahead on a thread of its own
    RandomAccessInt32VectorReader alignments_reader(alignments_rspecifier);
    Xent xent;
    CuMatrix<BaseFloat> feats, feats_transf, nnet_out, glob_err;

    Timer tim;
    double time_next = 0;
    KALDI_LOG << (crossvalidate ? "CROSSVALIDATE" : "TRAINING") << " STARTED";

    int32 num_done = 0, num_no_alignment = 0, num_other_error = 0;
    for (; !feature_reader.Done();) {
      std::string key = feature_reader.Key();
      if (!alignments_reader.HasKey(key)) {
        num_no_alignment++;
      } else {
        const Matrix<BaseFloat> &mat = feature_reader.Value();
        const std::vector<int32> &alignment = alignments_reader.Value(key);
        if ((int32)alignment.size() != mat.NumRows()) {
          KALDI_WARN << "Alignment has wrong size " << (alignment.size()) << " vs. " << (mat.NumRows());
          num_other_error++;
        } else {
          for (size_t i = 0; i < alignment.size(); ++i)
            if (alignment[i] < 0 || alignment[i] >= nnet.OutputDim())
              KALDI_ERR << "Alignment of " << key << " has label " << alignment[i] << " outside the network's "
                        << nnet.OutputDim() << " outputs";
          if (num_done % 10000 == 0) std::cout << num_done << ", " << std::flush;
          num_done++;
          feats.CopyFromMat(mat);  // host -> device
          nnet_transf.Feedforward(feats, &feats_transf);
          nnet.Propagate(feats_transf, &nnet_out);
          xent.EvalVec(nnet_out, alignment, &glob_err);
          if (!crossvalidate) nnet.Backpropagate(glob_err, NULL);
          tot_t += mat.NumRows();
        }
      }
      Timer t_features;
      feature_reader.Next();
      time_next += t_features.Elapsed();
    }

    CuDevice::Instantiate().Synchronize();
    if (!crossvalidate) nnet.Write(target_model_filename, binary);
    std::cout << "\n" << std::flush;

    KALDI_LOG << (crossvalidate ? "CROSSVALIDATE" : "TRAINING") << " FINISHED " << tim.Elapsed() << "s, fps"
              << tot_t / tim.Elapsed() << ", feature wait " << time_next << "s";
    KALDI_LOG << "Done " << num_done << " files, " << num_no_alignment << " with no alignments, " << num_other_error
              << " with other errors.";
    KALDI_LOG << xent.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
