#include "nnet/nnet-component.h"

#include <string.h>

#include "nnet/nnet-activation.h"
#include "nnet/nnet-biasedlinearity.h"
#include "nnet/nnet-nnet.h"
#include "nnet/nnet-rbm.h"
#include "nnet/nnet-various.h"

namespace kaldi {

// Markers as they appear in model files.  The research components of the fork (<cmvnbl>,
// <posnegbl>, <gaussbl>, <maskedbl>, <maskedrbm>, <rorbm>, <grbm>, <linbl>, <linrbm>, <hmmbl>,
// <codebl>) keep their marker so that a model containing one fails with "Missing type" rather
// than "Unknown marker".
const struct Component::key_value Component::kMarkerMap[] = {
    {Component::kBiasedLinearity, "<biasedlinearity>"}, {Component::kSigmoid, "<sigmoid>"},
    {Component::kSoftmax, "<softmax>"}, {Component::kRbm, "<rbm>"}, {Component::kExpand, "<expand>"},
    {Component::kCopy, "<copy>"}, {Component::kDropout, "<dropout>"}, {Component::kScale, ",<scale>"},
    {Component::kRelu, "<relu>"}, {Component::kSoftRelu, "<softrelu>"}, {Component::kCMVNBL, "<cmvnbl>"},
    {Component::kPosNegBL, "<posnegbl>"}, {Component::kGaussBL, "<gaussbl>"}, {Component::kMaskedBL, "<maskedbl>"},
    {Component::kMaskedRbm, "<maskedrbm>"}, {Component::kRoRbm, "<rorbm>"}, {Component::kGRbm, "<grbm>"},
    {Component::kLinBL, "<linbl>"}, {Component::kLinRbm, "<linrbm>"}, {Component::kHMMBL, "<hmmbl>"},
    {Component::kCodeBL, "<codebl>"}};

const char *Component::TypeToMarker(ComponentType t) {
  const int32 n = sizeof(kMarkerMap) / sizeof(kMarkerMap[0]);
  for (int32 i = 0; i < n; ++i)
    if (kMarkerMap[i].key == t) return kMarkerMap[i].value;
  KALDI_ERR << "Unknown type" << t;
  return NULL;
}

Component::ComponentType Component::MarkerToType(const std::string &s) {
  const int32 n = sizeof(kMarkerMap) / sizeof(kMarkerMap[0]);
  for (int32 i = 0; i < n; ++i)
    if (s == kMarkerMap[i].value) return kMarkerMap[i].key;
  KALDI_ERR << "Unknown marker" << s;
  return kUnknown;
}

Component *Component::Read(std::istream &is, bool binary, Nnet *nnet) {
  if (Peek(is, binary) == EOF) return NULL;
  std::string token;
  ReadToken(is, binary, &token);
  const ComponentType type = MarkerToType(token);
  int32 dim_out, dim_in;
  ReadBasicType(is, binary, &dim_out);
  ReadBasicType(is, binary, &dim_in);
  Component *c = NULL;
  switch (type) {
    case kBiasedLinearity: c = new BiasedLinearity(dim_in, dim_out, nnet); break;
    case kSigmoid: c = new Sigmoid(dim_in, dim_out, nnet); break;
    case kSoftmax: c = new Softmax(dim_in, dim_out, nnet); break;
    case kRbm: c = new Rbm(dim_in, dim_out, nnet); break;
    case kExpand: c = new Expand(dim_in, dim_out, nnet); break;
    case kCopy: c = new Copy(dim_in, dim_out, nnet); break;
    case kRelu: c = new Relu(dim_in, dim_out, nnet); break;
    case kSoftRelu: c = new SoftRelu(dim_in, dim_out, nnet); break;
    case kDropout: c = new Dropout(dim_in, dim_out, nnet); break;
    case kScale: c = new Scale(dim_in, dim_out, nnet); break;
    default: KALDI_ERR << "Missing type: " << token;
  }
  c->ReadData(is, binary);
  return c;
}

void Component::Write(std::ostream &os, bool binary) const {
  WriteToken(os, binary, TypeToMarker(GetType()));
  WriteBasicType(os, binary, OutputDim());
  WriteBasicType(os, binary, InputDim());
  if (!binary) os << "\n";
  WriteData(os, binary);
}

}  // namespace kaldi
