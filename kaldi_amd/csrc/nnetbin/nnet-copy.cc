// nnet-copy: reads a model and writes it back, possibly changing between the binary and the text
// format (nnetbin/nnet-copy.cc:22-68).  The parameters pass through the device matrices of the
// components, as in the reference.
//
// Command line, option names, usage text and log lines are derived from nnetbin/nnet-copy.cc of troylee/kaldi
// (Copyright 2012 Karel Vesely), licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); this file is distributed under the same license, "AS IS",
// without warranties or conditions of any kind.
#include "base/kaldi-common.h"
#include "nnet/nnet-nnet.h"
#include "util/common-utils.h"

int main(int argc, char *argv[]) {
  try {
    using namespace kaldi;
    const char *usage =
        "Copy Neural Network model (and possibly change binary/text format)\n"
        "Usage:  nnet-copy [options] <model-in> <model-out>\n"
        "e.g.:\n"
        " nnet-copy --binary=false nnet.mdl nnet_txt.mdl\n";
    bool binary_write = true;
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
    nnet.Write(model_out_filename, binary_write);
    KALDI_LOG << "Written model to " << model_out_filename;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
