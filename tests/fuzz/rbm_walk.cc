// A seeded random walk over the public Rbm API (Propagate through the component, Reconstruct, RbmUpdate, the
// learning knobs, the weight / bias accessors and the three ways of writing it out): the SAME source is compiled
// against the reference's classes (CPU branch; oracle/_ref/bin/rbm_walk) and against this build's
// (kaldi_amd/bin/rbm_walk).  tests/fuzz/rbm_walk.py writes the model files (one or more <rbm> components), runs
// both programs and compares the dumps record by record.  Sampling is left out on purpose: hidden states and the
// Gaussian noise come from each code base's own generator; the walk feeds probabilities where the trainer feeds
// samples, which goes through the same calls.
//
// usage: rbm_walk <model-file> <seed> <steps> <dump-file>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "base/kaldi-common.h"
#include "cudamatrix/cu-matrix.h"
#include "cudamatrix/cu-vector.h"
#include "nnet/nnet-component.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-nnet.h"
#include "nnet/nnet-rbm.h"

using namespace kaldi;

namespace {
struct Lcg {
  unsigned long long s;
  explicit Lcg(unsigned long long seed) : s(seed * 2862933555777941757ULL + 3037000493ULL) {}
  unsigned next() {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return static_cast<unsigned>(s >> 33);
  }
  int below(int n) { return static_cast<int>(next() % static_cast<unsigned>(n)); }
  float unit() { return (next() & 0xffffff) / 16777216.0f; }
  float sym() { return 2.0f * unit() - 1.0f; }
};

FILE *g_out = NULL;
void DumpMat(const std::string &what, const CuMatrix<BaseFloat> &m) {
  Matrix<BaseFloat> h;
  m.CopyToMat(&h);
  const int r = h.NumRows(), c = h.NumCols();
  fprintf(g_out, "mat %s %d %d\n", what.c_str(), r, c);
  for (int i = 0; i < r; ++i) fwrite(h.RowData(i), sizeof(BaseFloat), c, g_out);
  fputc('\n', g_out);
}
void DumpHostMat(const std::string &what, const Matrix<BaseFloat> &h) {
  const int r = h.NumRows(), c = h.NumCols();
  fprintf(g_out, "mat %s %d %d\n", what.c_str(), r, c);
  for (int i = 0; i < r; ++i) fwrite(h.RowData(i), sizeof(BaseFloat), c, g_out);
  fputc('\n', g_out);
}
void DumpHostVec(const std::string &what, const Vector<BaseFloat> &v) {
  fprintf(g_out, "mat %s 1 %d\n", what.c_str(), static_cast<int>(v.Dim()));
  fwrite(v.Data(), sizeof(BaseFloat), v.Dim(), g_out);
  fputc('\n', g_out);
}
void DumpText(const std::string &what, const std::string &text) {
  fprintf(g_out, "text %s %d 0\n", what.c_str(), static_cast<int>(text.size()));
  fwrite(text.data(), 1, text.size(), g_out);
  fputc('\n', g_out);
}
void FillMat(Lcg *g, CuMatrix<BaseFloat> *m, int r, int c, bool probabilities) {
  Matrix<BaseFloat> h(r, c);
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) h(i, j) = probabilities ? g->unit() : 1.5f * g->sym();
  m->CopyFromMat(h);
}
std::string Tag(int step, const char *name) {
  char buf[96];
  snprintf(buf, sizeof(buf), "step%d_%s", step, name);
  return buf;
}
}  // namespace

int main(int argc, char *argv[]) {
  if (argc != 5) {
    std::cerr << "usage: rbm_walk <model-file> <seed> <steps> <dump-file>\n";
    return 2;
  }
  try {
    Lcg g(strtoull(argv[2], NULL, 10));
    const int steps = atoi(argv[3]);
    g_out = fopen(argv[4], "wb");
    if (!g_out) return 2;

    Nnet nnet;
    nnet.Read(argv[1]);
    const int L = nnet.LayerCount();
    // the stack below the top one is a fixed transform (as rbm-train-cd1-frmshuff's --feature-transform); the top
    // component is the one in training
    if (nnet.Layer(L - 1)->GetType() != Component::kRbm) KALDI_ERR << "the last component must be an <rbm>";
    RbmBase *rbm = dynamic_cast<RbmBase *>(nnet.Layer(L - 1));
    Rbm *top = dynamic_cast<Rbm *>(nnet.Layer(L - 1));
    const int vis = top->InputDim(), hid = top->OutputDim();
    const bool vis_bern = rbm->VisType() == RbmBase::BERNOULLI, hid_bern = rbm->HidType() == RbmBase::BERNOULLI;
    {
      std::ostringstream os;
      os << "layers " << L << " vis " << vis << (vis_bern ? " bern" : " gauss") << " hid " << hid << (hid_bern ? " bern" : " gauss")
         << " updatable " << top->IsUpdatable() << " marker " << Component::TypeToMarker(top->GetType());
      DumpText("shape", os.str());
    }
    top->SetLearnRate(vis_bern || hid_bern ? 0.002f : 0.0002f);
    Mse mse;
    const int row_choices[] = {1, 3, 32, 100, 256};
    for (int step = 0; step < steps; ++step) {
      const int op = g.below(10);
      const int rows = row_choices[g.below(5)];
      if (getenv("WALK_TRACE")) fprintf(stderr, "step %d op %d rows %d\n", step, op, rows);
      CuMatrix<BaseFloat> pos_vis, pos_hid, neg_vis, neg_hid, err;
      switch (op) {
        case 0: {  // the knobs of the component in training (momentum may change at any time here)
          // (Gaussian units on both sides have no bound: at the step sizes the other pairs take, contrastive divergence
          // on random data blows up and the comparison would be of amplified rounding)
          top->SetLearnRate((vis_bern || hid_bern ? 0.0005f : 0.00005f) * (1 + g.below(8)));
          top->SetMomentum(g.below(2) ? 0.0f : 0.1f * (1 + g.below(9)));
          top->SetL2Penalty(g.below(2) ? 0.0f : 2e-4f * (1 + g.below(5)));
          std::ostringstream os;
          os << top->GetLearnRate() << " " << top->GetMomentum() << " " << top->GetL2Penalty();
          DumpText(Tag(step, "knobs"), os.str());
          break;
        }
        case 1: {  // visible -> hidden through the component interface
          FillMat(&g, &pos_vis, rows, vis, vis_bern);
          top->Propagate(pos_vis, &pos_hid);
          DumpMat(Tag(step, "propagate"), pos_hid);
          break;
        }
        case 2: {  // hidden -> visible, into a matrix of the right size, of another size, or an empty one
          FillMat(&g, &pos_hid, rows, hid, hid_bern);
          const int pre = g.below(3);
          if (pre == 1) neg_vis.Resize(rows, vis);
          if (pre == 2) neg_vis.Resize(rows + 1, vis);
          rbm->Reconstruct(pos_hid, &neg_vis);
          DumpMat(Tag(step, "reconstruct"), neg_vis);
          break;
        }
        case 3:
        case 4:
        case 5: {  // one contrastive-divergence step on probabilities
          FillMat(&g, &pos_vis, rows, vis, vis_bern);
          top->Propagate(pos_vis, &pos_hid);
          rbm->Reconstruct(pos_hid, &neg_vis);
          top->Propagate(neg_vis, &neg_hid);
          rbm->RbmUpdate(pos_vis, pos_hid, neg_vis, neg_hid);
          mse.Eval(neg_vis, pos_vis, &err);
          DumpMat(Tag(step, "cd1_neg_hid"), neg_hid);
          DumpMat(Tag(step, "cd1_err"), err);
          break;
        }
        case 6: {  // an update from four unrelated matrices
          FillMat(&g, &pos_vis, rows, vis, vis_bern);
          FillMat(&g, &neg_vis, rows, vis, vis_bern);
          FillMat(&g, &pos_hid, rows, hid, hid_bern);
          FillMat(&g, &neg_hid, rows, hid, hid_bern);
          rbm->RbmUpdate(pos_vis, pos_hid, neg_vis, neg_hid);
          Matrix<BaseFloat> w;
          top->GetWeight(&w);
          DumpHostMat(Tag(step, "update_weight"), w);
          break;
        }
        case 7: {  // parameters out (and now and then back in, scaled)
          Matrix<BaseFloat> w;
          Vector<BaseFloat> vb, hb;
          top->GetWeight(&w);
          top->GetVisibleBias(&vb);
          top->GetHiddenBias(&hb);
          DumpHostMat(Tag(step, "weight"), w);
          DumpHostVec(Tag(step, "vis_bias"), vb);
          DumpHostVec(Tag(step, "hid_bias"), hb);
          CuMatrix<BaseFloat> dw;
          top->GetWeight(&dw);
          DumpMat(Tag(step, "weight_device"), dw);
          if (g.below(3) == 0) {
            for (int i = 0; i < w.NumRows(); ++i)
              for (int j = 0; j < w.NumCols(); ++j) w(i, j) *= 0.9f;
            for (int i = 0; i < vb.Dim(); ++i) vb(i) *= 0.5f;
            for (int i = 0; i < hb.Dim(); ++i) hb(i) *= 1.1f;
            top->SetWeight(w);
            top->SetVisibleBias(vb);
            top->SetHiddenBias(hb);
          }
          break;
        }
        case 8: {  // the three written forms, as text
          std::ostringstream a, b, c, d, e;
          top->Write(a, false);
          DumpText(Tag(step, "write"), a.str());
          rbm->WriteAsNnet(b, false);
          DumpText(Tag(step, "write_as_nnet"), b.str());
          rbm->WriteAsAutoEncoder(c, true, false);
          DumpText(Tag(step, "write_as_encoder"), c.str());
          rbm->WriteAsAutoEncoder(d, false, false);
          DumpText(Tag(step, "write_as_decoder"), d.str());
          rbm->WriteAsNnet(e, true);
          std::ostringstream len;
          len << e.str().size();
          DumpText(Tag(step, "write_as_nnet_binary_bytes"), len.str());
          DumpText(Tag(step, "mse_report"), mse.Report());
          break;
        }
        case 9: {  // the whole stack forward, as the trainer's feature transform and as nnet-forward would
          CuMatrix<BaseFloat> in, out;
          FillMat(&g, &in, rows, nnet.InputDim(), nnet.Layer(0)->GetType() == Component::kRbm &&
                                                     dynamic_cast<RbmBase *>(nnet.Layer(0))->VisType() == RbmBase::BERNOULLI);
          nnet.Feedforward(in, &out);
          DumpMat(Tag(step, "stack_feedforward"), out);
          nnet.Propagate(in, &out);
          DumpMat(Tag(step, "stack_propagate"), out);
          break;
        }
      }
    }
    fclose(g_out);
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return 1;
  }
}
