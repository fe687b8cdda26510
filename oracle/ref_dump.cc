// ref_dump: drives the REFERENCE's own classes (linked from oracle/_ref/libkaldi_ref.a, built from
// /root/reference/src with HAVE_CUDA off, so every CuMatrix op runs its CPU branch) and dumps the
// tensors of the nnet1 hot path into a Kaldi matrix archive.  tests/golden/make_golden.py turns
// the archive into the committed .npz fixtures.  TEST INFRASTRUCTURE ONLY; this file is our own
// code, the reference is only called.
//
//   ref_dump dnn <model> <feats.ark> <ali.ark> <out.ark> <nsteps> <lr> <mom> <l2> <l1> <l2ub> <avg>
//       first utterance of feats.ark = one bunch; nsteps x {Propagate, Xent::EvalVec, Backpropagate}
//   ref_dump rbm <model> <feats.ark> <rand.ark> <out.ark> <nsteps> <lr> <mom> <l2>
//       CD-1 steps with the uniform / gaussian draws read from rand.ark (keys u0,u1.. / g0,g1..)
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "cudamatrix/cu-math.h"
#include "nnet/nnet-biasedlinearity.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-nnet.h"
#include "nnet/nnet-rbm.h"
#include "util/common-utils.h"

using namespace kaldi;

static BaseFloatMatrixWriter *g_out = NULL;

static void Dump(const std::string &key, const Matrix<BaseFloat> &m) { g_out->Write(key, m); }
static void Dump(const std::string &key, const CuMatrix<BaseFloat> &m) {
  Matrix<BaseFloat> h;
  m.CopyToMat(&h);
  Dump(key, h);
}
static void Dump(const std::string &key, const CuVector<BaseFloat> &v) {
  Vector<BaseFloat> h;
  v.CopyToVec(&h);
  Matrix<BaseFloat> m(1, h.Dim());
  for (int32 i = 0; i < h.Dim(); i++) m(0, i) = h(i);
  Dump(key, m);
}
static void DumpScalar(const std::string &key, double v) {
  // two floats (hi, lo) so that a double survives the float archive
  Matrix<BaseFloat> m(1, 2);
  m(0, 0) = static_cast<float>(v);
  m(0, 1) = static_cast<float>(v - static_cast<double>(m(0, 0)));
  Dump(key, m);
}
static std::string K(const std::string &a, int i) {
  std::ostringstream os;
  os << a << i;
  return os.str();
}
template <class T>
static std::string K(const std::string &a, int i, const T &b) {
  std::ostringstream os;
  os << a << i << b;
  return os.str();
}

// reach the protected momentum buffers of a component the factory built
struct BiasedLinearityPeek : public BiasedLinearity {
  const CuMatrix<BaseFloat> &Corr() const { return linearity_corr_; }
  const CuVector<BaseFloat> &BiasCorr() const { return bias_corr_; }
};

// "Xent:<loss> frames:<n> err/frm:..\nFRAME_ACCURACY >> <acc>% <<"
static void ParseReport(const std::string &rep, double *loss, double *frames, double *acc) {
  *loss = atof(rep.c_str() + rep.find("Xent:") + 5);
  *frames = atof(rep.c_str() + rep.find("frames:") + 7);
  size_t p = rep.find(">> ");
  *acc = (p == std::string::npos) ? -1.0 : atof(rep.c_str() + p + 3);
}

static int RunDnn(int argc, char **argv) {
  if (argc != 13) {
    std::cerr << "usage: ref_dump dnn model feats.ark ali.ark out.ark nsteps lr mom l2 l1 l2ub avg\n";
    return 1;
  }
  std::string model = argv[2], feats_rs = std::string("ark:") + argv[3], ali_rs = std::string("ark:") + argv[4];
  BaseFloatMatrixWriter out(std::string("ark:") + argv[5]);
  g_out = &out;
  const int nsteps = atoi(argv[6]);
  Nnet nnet;
  nnet.Read(model);
  nnet.SetLearnRate(atof(argv[7]), NULL);
  nnet.SetMomentum(atof(argv[8]));
  nnet.SetL2Penalty(atof(argv[9]));
  nnet.SetL1Penalty(atof(argv[10]));
  nnet.SetL2UpperBound(atof(argv[11]));
  nnet.SetAverageGrad(atoi(argv[12]) != 0);

  SequentialBaseFloatMatrixReader feature_reader(feats_rs);
  RandomAccessInt32VectorReader ali_reader(ali_rs);
  const Matrix<BaseFloat> &mat = feature_reader.Value();
  const std::vector<int32> &targets = ali_reader.Value(feature_reader.Key());

  CuMatrix<BaseFloat> nnet_in, nnet_out, glob_err;
  nnet_in.CopyFromMat(mat);
  Xent xent;
  for (int s = 0; s < nsteps; s++) {
    nnet.Propagate(nnet_in, &nnet_out);
    for (int32 i = 0; i <= nnet.LayerCount(); i++) Dump(K("s", s, "_prop") + K("", i), nnet.PropagateBuffer()[i]);
    Dump(K("s", s, "_post"), nnet_out);
    xent.EvalVec(nnet_out, targets, &glob_err);
    Dump(K("s", s, "_globerr"), glob_err);
    double loss, frames, acc;
    ParseReport(xent.Report(), &loss, &frames, &acc);
    DumpScalar(K("s", s, "_cumloss"), loss);
    DumpScalar(K("s", s, "_cumframes"), frames);
    DumpScalar(K("s", s, "_cumacc"), acc);
    nnet.Backpropagate(glob_err, NULL);
    for (int32 i = 0; i < nnet.LayerCount() - 1; i++) Dump(K("s", s, "_bprop") + K("", i), nnet.BackpropagateBuffer()[i]);
    for (int32 i = 0; i < nnet.LayerCount(); i++) {
      if (nnet.Layer(i)->GetType() != Component::kBiasedLinearity) continue;
      BiasedLinearity *bl = dynamic_cast<BiasedLinearity *>(nnet.Layer(i));
      Dump(K("s", s, "_W") + K("", i), bl->GetLinearityWeight());
      Dump(K("s", s, "_b") + K("", i), bl->GetBiasWeight());
      const BiasedLinearityPeek *pk = static_cast<const BiasedLinearityPeek *>(bl);
      Dump(K("s", s, "_Wcorr") + K("", i), pk->Corr());
      Dump(K("s", s, "_bcorr") + K("", i), pk->BiasCorr());
    }
  }
  std::cerr << xent.Report() << "\n";
  return 0;
}

static int RunRbm(int argc, char **argv) {
  if (argc != 10) {
    std::cerr << "usage: ref_dump rbm model feats.ark rand.ark out.ark nsteps lr mom l2\n";
    return 1;
  }
  Nnet nnet;
  nnet.Read(argv[2]);
  KALDI_ASSERT(nnet.LayerCount() == 1 && nnet.Layer(0)->GetType() == Component::kRbm);
  Rbm &rbm = dynamic_cast<Rbm &>(*nnet.Layer(0));
  SequentialBaseFloatMatrixReader feature_reader(std::string("ark:") + argv[3]);
  RandomAccessBaseFloatMatrixReader rand_reader(std::string("ark:") + argv[4]);
  BaseFloatMatrixWriter out(std::string("ark:") + argv[5]);
  g_out = &out;
  const int nsteps = atoi(argv[6]);
  rbm.SetLearnRate(atof(argv[7]));
  rbm.SetMomentum(atof(argv[8]));
  rbm.SetL2Penalty(atof(argv[9]));

  CuMatrix<BaseFloat> pos_vis, pos_hid, neg_vis, neg_hid, noise, dummy;
  pos_vis.CopyFromMat(feature_reader.Value());
  MseProgress mse;
  for (int s = 0; s < nsteps; s++) {
    rbm.Propagate(pos_vis, &pos_hid);
    Dump(K("s", s, "_pos_hid"), pos_hid);
    if (rbm.HidType() == Rbm::BERNOULLI) {
      // state = (uniform < prob) ? 1 : 0 with the uniform matrix injected
      // (cu-rand-inl.h:163-167; the reference's CPU branch forgets to size `states`)
      const Matrix<BaseFloat> &u = rand_reader.Value(K("u", s));
      Matrix<BaseFloat> probs, st(u.NumRows(), u.NumCols());
      pos_hid.CopyToMat(&probs);
      for (int32 r = 0; r < u.NumRows(); r++)
        for (int32 c = 0; c < u.NumCols(); c++) st(r, c) = (u(r, c) < probs(r, c)) ? 1 : 0;
      neg_hid.CopyFromMat(st);
    } else {
      // AddGaussNoise (cu-rand-inl.h:173-177) with the gaussian matrix injected
      noise.CopyFromMat(rand_reader.Value(K("g", s)));
      neg_hid.CopyFromMat(pos_hid);
      neg_hid.AddMat(1.0, noise, 1.0);
    }
    Dump(K("s", s, "_neg_hid_sample"), neg_hid);
    rbm.Reconstruct(neg_hid, &neg_vis);
    Dump(K("s", s, "_neg_vis"), neg_vis);
    rbm.Propagate(neg_vis, &neg_hid);
    Dump(K("s", s, "_neg_hid"), neg_hid);
    rbm.RbmUpdate(pos_vis, pos_hid, neg_vis, neg_hid);
    mse.Eval(neg_vis, pos_vis, &dummy);
    Dump(K("s", s, "_mse_diff"), dummy);
    Matrix<BaseFloat> W;
    Vector<BaseFloat> vb, hb;
    rbm.GetWeight(&W);
    rbm.GetVisibleBias(&vb);
    rbm.GetHiddenBias(&hb);
    Dump(K("s", s, "_W"), W);
    Matrix<BaseFloat> mv(1, vb.Dim()), mh(1, hb.Dim());
    for (int32 i = 0; i < vb.Dim(); i++) mv(0, i) = vb(i);
    for (int32 i = 0; i < hb.Dim(); i++) mh(0, i) = hb(i);
    Dump(K("s", s, "_vis_bias"), mv);
    Dump(K("s", s, "_hid_bias"), mh);
  }
  // the RBM as a DNN front layer, text form (rbm-convert-to-nnet path)
  std::ostringstream os;
  rbm.WriteAsNnet(os, false);
  std::cout << os.str();
  return 0;
}

int main(int argc, char **argv) {
  try {
    if (argc >= 2 && std::string(argv[1]) == "dnn") return RunDnn(argc, argv);
    if (argc >= 2 && std::string(argv[1]) == "rbm") return RunRbm(argc, argv);
    std::cerr << "usage: ref_dump dnn|rbm ...\n";
    return 1;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
