// rbm-train-cd1-frmshuff: one epoch of contrastive-divergence (CD-1) training of a single RBM
// on frame-shuffled minibatches.  Command line, defaults and log lines of the reference tool
// (nnetbin/rbm-train-cd1-frmshuff.cc:29-191).
//
// Command line, option names, usage text and log lines are derived from nnetbin/rbm-train-cd1-frmshuff.cc of troylee/kaldi
// (Copyright 2012 Karel Vesely), licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); this file is distributed under the same license, "AS IS",
// without warranties or conditions of any kind.
#include "base/kaldi-common.h"
#include "cudamatrix/cu-device.h"
#include "cudamatrix/cu-rand.h"
#include "nnet/nnet-cache.h"
#include "nnetbin/feature-upload.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-nnet.h"
#include "nnet/nnet-rbm.h"
#include "util/common-utils.h"
#include "util/timer.h"

int main(int argc, char *argv[]) {
  using namespace kaldi;
  try {
    const char *usage =
        "Perform iteration of RBM training by contrastive divergence alg.\n"
        "Usage:  rbm-train-cd1-frmshuff [options] <model-in> <feature-rspecifier> <model-out>\n"
        "e.g.: \n"
        " rbm-train-cd1-frmshuff rbm.init scp:train.scp rbm.iter1\n";

    ParseOptions po(usage);
    bool binary = false;
    po.Register("binary", &binary, "Write output in binary mode");
    BaseFloat learn_rate = 0.008, momentum = 0.0, l2_penalty = 0.0;
    po.Register("learn-rate", &learn_rate, "Learning rate");
    po.Register("momentum", &momentum, "Momentum");
    po.Register("l2-penalty", &l2_penalty, "L2 penalty (weight decay)");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform Neural Network");
    int32 bunchsize = 512, cachesize = 32768;
    po.Register("bunchsize", &bunchsize, "Size of weight update block");
    po.Register("cachesize", &cachesize, "Size of cache for frame level shuffling");
    int32 gpu_id = -1;
    po.Register("gpu-id", &gpu_id, "GPU ordinal (-1: the GPU with most free memory)");
    bool profile = false;
    po.Register("profile", &profile, "Synchronise after every GPU call and print a per-function time profile");

    po.Read(argc, argv);
    if (po.NumArgs() != 3) {
      po.PrintUsage();
      exit(1);
    }
    std::string model_filename = po.GetArg(1), feature_rspecifier = po.GetArg(2), target_model_filename = po.GetArg(3);

    CuDevice::Instantiate().SelectGpuId(gpu_id);
    CuDevice::Instantiate().Profile(profile);

    Nnet rbm_transf;
    if (feature_transform != "") rbm_transf.Read(feature_transform);

    Nnet nnet;
    nnet.Read(model_filename);
    KALDI_ASSERT(nnet.LayerCount() == 1);
    KALDI_ASSERT(nnet.Layer(0)->GetType() == Component::kRbm);
    RbmBase &rbm = dynamic_cast<RbmBase &>(*nnet.Layer(0));
    rbm.SetLearnRate(learn_rate);
    rbm.SetMomentum(momentum);
    rbm.SetL2Penalty(l2_penalty);

    kaldi::int64 tot_t = 0;
    PrefetchingBaseFloatMatrixReader feature_reader(feature_rspecifier);  // reads ahead on a thread of its own

    Cache cache;
    // (the reference divides by it first and dies of the division, nnetbin/nnet-train-xent-hardlab-frmshuff.cc:115)
    if (bunchsize <= 0) KALDI_ERR << "--bunchsize must be positive, got " << bunchsize;
    cachesize = (cachesize / bunchsize) * bunchsize;
    cache.Init(cachesize, bunchsize);

    CuRand<BaseFloat> cu_rand;
    MseProgress mse;

    FeatureUploader uploader(cachesize);
    CuMatrix<BaseFloat> feats_transf, pos_vis, pos_hid, neg_vis, neg_hid;
    std::vector<int32> dummy_cache_vec;
    CuStlVector<int32> dummy_targets;

    Timer tim;
    double time_next = 0;
    KALDI_LOG << "RBM TRAINING STARTED";

    int32 num_done = 0, num_cache = 0;
    while (1) {
      while (!cache.Full() && !feature_reader.Done()) {
        const Matrix<BaseFloat> &mat = feature_reader.Value();
        dummy_cache_vec.resize(mat.NumRows());  // the cache wants labels; RBM training has none
        const CuMatrix<BaseFloat> &feats = uploader.Upload(mat);  // host -> device, beside the training
        rbm_transf.Feedforward(feats, &feats_transf);
        cache.AddData(feats_transf, dummy_cache_vec);
        num_done++;
        Timer t_features;
        feature_reader.Next();
        time_next += t_features.Elapsed();
      }
      cache.Randomize();
      std::cerr << "Cache #" << ++num_cache << " " << (cache.Randomized() ? "[RND]" : "[NO-RND]")
                << " segments: " << num_done << " frames: " << tot_t << "\n";
      while (!cache.Empty()) {
        cache.GetBunch(&pos_vis, &dummy_targets);
        // positive phase, and the sample of the hidden layer that drives the reconstruction
        if (rbm.HidType() == Rbm::BERNOULLI) {
          // (rbm.Propagate(pos_vis, &pos_hid); cu_rand.BinarizeProbs(pos_hid, &neg_hid); in one launch)
          rbm.PropagateSample(pos_vis, &cu_rand, &pos_hid, &neg_hid);
        } else {
          rbm.Propagate(pos_vis, &pos_hid);
          neg_hid.CopyFromMat(pos_hid);
          cu_rand.AddGaussNoise(&neg_hid);
        }
        // negative phase
        rbm.Reconstruct(neg_hid, &neg_vis);
        rbm.Propagate(neg_vis, &neg_hid);
        // the update and the reconstruction error mse.Eval(neg_vis, pos_vis, ...) of the reference's loop in one
        // call: the error only reads the two visible matrices and rides in the update's weight launch
        rbm.RbmUpdate(pos_vis, pos_hid, neg_vis, neg_hid, mse.DeviceLoss());
        mse.Evaluated(neg_vis.NumRows());
        tot_t += pos_vis.NumRows();
      }
      if (feature_reader.Done()) break;
    }

    CuDevice::Instantiate().Synchronize();
    nnet.Write(target_model_filename, binary);

    std::cout << "\n" << std::flush;
    KALDI_LOG << "RBM TRAINING FINISHED " << tim.Elapsed() << "s, fps" << tot_t / tim.Elapsed() << ", feature wait "
              << time_next << "s";
    KALDI_LOG << "Done " << num_done << " files.";
    KALDI_LOG << mse.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
