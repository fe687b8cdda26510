// rbm-convert-to-nnet: writes a trained RBM as <biasedlinearity> (+ <sigmoid> for Bernoulli hidden
// units), the step between RBM pre-training of one layer and stacking the next
// (nnetbin/rbm-convert-to-nnet.cc:22-75, Rbm::WriteAsNnet nnet/nnet-rbm.h:233-249).
//
// Command line, option names, usage text and log lines are derived from nnetbin/rbm-convert-to-nnet.cc of troylee/kaldi
// (Copyright 2009-2011 Microsoft Corporation), licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); this file is distributed under the same license, "AS IS",
// without warranties or conditions of any kind.
#include "base/kaldi-common.h"
#include "nnet/nnet-nnet.h"
#include "nnet/nnet-rbm.h"
#include "util/common-utils.h"

int main(int argc, char *argv[]) {
  try {
    using namespace kaldi;
    const char *usage =
        "Convert RBM to <biasedlinearity> and <sigmoid>\n"
        "Usage:  rbm-convert-to-nnet [options] <rbm-in> <nnet-out>\n"
        "e.g.:\n"
        " rbm-convert-to-nnet --binary=false rbm.mdl nnet.mdl\n";
    bool binary_write = false;
    ParseOptions po(usage);
    po.Register("binary", &binary_write, "Write output in binary mode");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    std::string model_in_filename = po.GetArg(1), model_out_filename = po.GetArg(2);
    Nnet nnet;
    nnet.Read(model_in_filename);
    KALDI_ASSERT(nnet.LayerCount() == 1);
    KALDI_ASSERT(nnet.Layer(0)->GetType() == Component::kRbm);
    RbmBase &rbm = dynamic_cast<RbmBase &>(*nnet.Layer(0));
    {
      Output ko(model_out_filename, binary_write);
      rbm.WriteAsNnet(ko.Stream(), binary_write);
    }
    KALDI_LOG << "Written model to " << model_out_filename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
