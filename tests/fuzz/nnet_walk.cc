// A seeded random walk over the public Nnet / Component / Xent / Mse / MseProgress / Cache API: the SAME source is
// compiled against the reference's classes (CPU branch; oracle/Makefile -> oracle/_ref/bin/nnet_walk) and against
// this build's classes (kaldi_amd/csrc/Makefile -> kaldi_amd/bin/nnet_walk).  Every step makes one call (or one
// training step) and appends what it changed to a dump file; tests/fuzz/nnet_walk.py writes the model files, runs
// both programs and compares the dumps record by record.
//
// usage: nnet_walk <model-file> <seed> <steps> <dump-file>
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
#include "nnet/nnet-nnet.h"
// (the reference's activations need the network's definition in front of them, nnet/nnet-activation.h:163)
#include "nnet/nnet-activation.h"
#include "nnet/nnet-biasedlinearity.h"
#include "nnet/nnet-loss.h"
#include "nnet/nnet-cache.h"
// the reference's two cache headers share one include guard (nnet/nnet-cache.h:19, nnet/nnet-cache-tgtmat.h:19): a
// program that wants both has to drop it in between
#undef KALDI_NNET_CACHE_H
#include "nnet/nnet-cache-tgtmat.h"

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
void DumpText(const std::string &what, const std::string &text) {
  fprintf(g_out, "text %s %d 0\n", what.c_str(), static_cast<int>(text.size()));
  fwrite(text.data(), 1, text.size(), g_out);
  fputc('\n', g_out);
}
void DumpInts(const std::string &what, const std::vector<int32> &v) {
  std::ostringstream os;
  for (size_t i = 0; i < v.size(); ++i) os << v[i] << " ";
  DumpText(what, os.str());
}
void FillMat(Lcg *g, CuMatrix<BaseFloat> *m, int r, int c, float scale = 1.0f) {
  Matrix<BaseFloat> h(r, c);
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) h(i, j) = scale * g->sym();
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
    std::cerr << "usage: nnet_walk <model-file> <seed> <steps> <dump-file>\n";
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
    {
      std::ostringstream os;
      os << "layers " << L;
      if (L > 0) os << " in " << nnet.InputDim() << " out " << nnet.OutputDim();
      for (int i = 0; i < L; ++i) {
        Component *c = nnet.Layer(i);
        os << " | " << Component::TypeToMarker(c->GetType()) << " " << c->InputDim() << " " << c->OutputDim() << " "
           << (c->IsUpdatable() ? "updatable" : "fixed") << " at " << nnet.IndexOfLayer(*c);
      }
      DumpText("shape", os.str());
    }
    if (L == 0) {  // an empty network passes its input through
      CuMatrix<BaseFloat> in, out;
      FillMat(&g, &in, 5, 7);
      nnet.Propagate(in, &out);
      DumpMat("empty_propagate", out);
      nnet.Feedforward(in, &out);
      DumpMat("empty_feedforward", out);
      fclose(g_out);
      return 0;
    }
    const int in_dim = nnet.InputDim(), out_dim = nnet.OutputDim();
    const bool posteriors = nnet.Layer(L - 1)->GetType() == Component::kSoftmax;
    int updatable = 0;
    bool trainable = true;  // <expand> and <copy> have no backward pass (nnet/nnet-various.h:75,130)
    for (int i = 0; i < L; ++i) {
      updatable += nnet.Layer(i)->IsUpdatable() ? 1 : 0;
      const Component::ComponentType t = nnet.Layer(i)->GetType();
      if (t == Component::kExpand || t == Component::kCopy) trainable = false;
    }
    int xent_frames = 0;

    Xent xent;
    Mse mse;
    MseProgress mse_progress(300);
    const int row_choices[] = {1, 2, 5, 32, 100, 256};
    nnet.SetLearnRate(0.001f, NULL);
    // The momentum changes along the walk too (op 0), between zero and non-zero: the reference's first momentum step
    // reuses the gradient of the update before it, which it keeps even with momentum 0
    // (nnet/nnet-biasedlinearity.h:76-82).
    nnet.SetMomentum(g.below(2) ? 0.0f : 0.1f * (1 + g.below(8)));
    // How this walk looks at the pass buffers: 0 never; 1 at the forward buffers only, whenever it likes; 2 at both,
    // having asked for them once before the first pass (a backward buffer is only there for a pass that ran after
    // the first look, see Nnet::BackpropagateBuffer of this build).
    const int looks = g.below(3);
    if (looks == 2) DumpText("buffers_before", nnet.BackpropagateBuffer().size() + 1 >= static_cast<size_t>(L) ? "enough" : "short");
    for (int step = 0; step < steps; ++step) {
      const int op = g.below(14);
      const int rows = row_choices[g.below(6)];
      if (getenv("WALK_TRACE")) fprintf(stderr, "step %d op %d rows %d\n", step, op, rows);
      CuMatrix<BaseFloat> in, out, err, in_err;
      switch (op) {
        case 0: {  // the trainer's knobs
          const float lr = 0.0002f * (1 + g.below(8));
          std::string factors;
          if (g.below(2) && updatable > 0) {
            const char *f[] = {"1.0", "0.5", "0.0", "2"};
            const char *sep[] = {":", ",", ";", " "};
            for (int i = 0; i < updatable; ++i) factors += std::string(i ? sep[g.below(4)] : "") + f[g.below(4)];
          }
          nnet.SetLearnRate(lr, factors.empty() ? NULL : factors.c_str());
          nnet.SetMomentum(g.below(2) ? 0.0f : 0.1f * (1 + g.below(8)));
          // (no L2 penalty here: the reference's CPU branch of that update asserts for a weight matrix that is not
          // square -- it adds the matrix to itself through the transposed-alias check, matrix/kaldi-matrix.cc:281)
          nnet.SetL2Penalty(0.0f);
          nnet.SetL1Penalty(g.below(3) ? 0.0f : 1e-5f * (1 + g.below(5)));
          nnet.SetAverageGrad(g.below(4) == 0);
          // (a bound on the L2 norm of each weight row after the step, nnet/nnet-biasedlinearity.h:98-101)
          nnet.SetL2UpperBound(g.below(3) ? 0.0f : 0.25f * (1 + g.below(8)));
          std::ostringstream os;
          os << nnet.GetLearnRateString() << " | " << nnet.GetLearnRate();
          for (int i = 0; i < L; ++i)
            if (nnet.Layer(i)->IsUpdatable()) {
              UpdatableComponent *uc = dynamic_cast<UpdatableComponent *>(nnet.Layer(i));
              os << " | " << uc->GetLearnRate() << " " << uc->GetMomentum() << " " << uc->GetL2Penalty() << " "
                 << uc->GetL1Penalty() << " " << uc->GetAverageGrad() << " " << uc->GetL2UpperBound();
            }
          DumpText(Tag(step, "knobs"), os.str());
          break;
        }
        case 1: {  // forward with the buffers kept
          FillMat(&g, &in, rows, in_dim);
          nnet.Propagate(in, &out);
          DumpMat(Tag(step, "propagate"), out);
          if (looks == 0) break;
          const std::vector<CuMatrix<BaseFloat> > &buf = nnet.PropagateBuffer();
          const int k = g.below(L + 1);
          std::ostringstream os;
          os << "buffers " << (static_cast<int>(buf.size()) >= L + 1 ? "enough" : "short");
          DumpText(Tag(step, "propagate_buffers"), os.str());
          DumpMat(Tag(step, "propagate_buffer"), buf[k]);
          break;
        }
        case 2:
        case 3:
        case 4: {  // one training step on hard labels (posteriors) or towards a target matrix
          if (!trainable) break;
          FillMat(&g, &in, rows, in_dim);
          nnet.Propagate(in, &out);
          if (posteriors) {
            std::vector<int32> tgt(rows);
            for (int i = 0; i < rows; ++i) tgt[i] = g.below(out_dim);
            xent.EvalVec(out, tgt, &err);
            xent_frames += rows;
          } else {
            CuMatrix<BaseFloat> target;
            FillMat(&g, &target, rows, out_dim, 0.5f);
            if (g.below(2)) mse.Eval(out, target, &err);
            else mse_progress.Eval(out, target, &err);
          }
          DumpMat(Tag(step, "train_err"), err);
          if (g.below(3) == 0) {
            nnet.Backpropagate(err, &in_err);
            DumpMat(Tag(step, "train_in_err"), in_err);
          } else {
            nnet.Backpropagate(err, NULL);
          }
          if (L > 1 && looks == 2) {
            const std::vector<CuMatrix<BaseFloat> > &buf = nnet.BackpropagateBuffer();
            DumpMat(Tag(step, "backpropagate_buffer"), buf[g.below(L - 1)]);
          }
          break;
        }
        case 5: {  // forward without buffers
          FillMat(&g, &in, rows, in_dim);
          nnet.Feedforward(in, &out);
          DumpMat(Tag(step, "feedforward"), out);
          break;
        }
        case 6: {  // the model as text, whole and the front layers only
          std::ostringstream os;
          nnet.Write(os, false);
          DumpText(Tag(step, "write_text"), os.str());
          std::ostringstream front;
          nnet.WriteFrontLayers(front, false, g.below(L + 1));
          DumpText(Tag(step, "write_front"), front.str());
          std::ostringstream bin;
          nnet.Write(bin, true);
          std::ostringstream len;
          len << bin.str().size();
          DumpText(Tag(step, "write_binary_bytes"), len.str());
          break;
        }
        case 7: {  // the loss reports
          DumpText(Tag(step, "xent_report"), xent.Report());
          if (xent_frames > 0) {  // with no frame seen the call is an error (nnet/nnet-loss.cc:115)
            std::ostringstream os;
            os << xent.GetFrameAccuracy();
            DumpText(Tag(step, "xent_accuracy"), os.str());
          }
          DumpText(Tag(step, "mse_report"), mse.Report());
          DumpText(Tag(step, "mse_progress_report"), mse_progress.Report());
          if (g.below(3) == 0) {
            xent.Reset();
            xent_frames = 0;
          }
          break;
        }
        case 8: {  // cross-entropy against soft labels
          if (!posteriors) break;
          FillMat(&g, &in, rows, in_dim);
          nnet.Feedforward(in, &out);
          Matrix<BaseFloat> t(rows, out_dim);
          for (int i = 0; i < rows; ++i) t(i, g.below(out_dim)) = 1.0f;
          CuMatrix<BaseFloat> target;
          target.CopyFromMat(t);
          Xent soft;
          soft.Eval(out, target, &err);
          DumpMat(Tag(step, "soft_err"), err);
          DumpText(Tag(step, "soft_report"), soft.Report());
          break;
        }
        case 9: {  // one component on its own
          Component *c = nnet.Layer(g.below(L));
          FillMat(&g, &in, rows, c->InputDim());
          c->Propagate(in, &out);
          DumpMat(Tag(step, "component_propagate"), out);
          std::ostringstream os;
          c->Write(os, false);
          DumpText(Tag(step, "component_write"), os.str());
          break;
        }
        case 10: {  // the frame cache, filled from a few "utterances", emptied bunch by bunch
          Cache cache;
          const int bunch = 1 + g.below(64), cachesize = bunch * (1 + g.below(6));
          cache.Init(cachesize, bunch);
          const bool shuffle = g.below(2);
          std::ostringstream log;
          int next_label = 0;
          for (int u = 0; u < 6 && !cache.Full(); ++u) {
            const int n = 1 + g.below(cachesize);
            CuMatrix<BaseFloat> feats;
            FillMat(&g, &feats, n, 9);
            std::vector<int32> labels(n);
            for (int i = 0; i < n; ++i) labels[i] = next_label++;
            cache.AddData(feats, labels);
            log << "add " << n << " full " << cache.Full() << " empty " << cache.Empty() << " | ";
          }
          // (the permutation comes from the process-wide rand() stream, which neither code base seeds)
          if (shuffle && !cache.Empty()) cache.Randomize();
          log << "randomized " << cache.Randomized() << " | ";
          int bunches = 0;
          while (!cache.Empty() && bunches < 8) {
            CuMatrix<BaseFloat> feats;
            std::vector<int32> labels;
            cache.GetBunch(&feats, &labels);
            DumpMat(Tag(step, "cache_bunch"), feats);
            DumpInts(Tag(step, "cache_labels"), labels);
            ++bunches;
          }
          log << "bunches " << bunches << " empty " << cache.Empty() << " full " << cache.Full();
          DumpText(Tag(step, "cache_log"), log.str());
          break;
        }
        case 11: {  // a round trip of the model through a binary stream into a second network
          std::ostringstream os;
          nnet.Write(os, true);
          std::istringstream is(os.str());
          Nnet copy;
          copy.Read(is, true);
          FillMat(&g, &in, rows, in_dim);
          copy.Feedforward(in, &out);
          DumpMat(Tag(step, "reread_feedforward"), out);
          break;
        }
        case 12: {  // an affine layer's weight accessors (nnet/nnet-biasedlinearity.h:104-148), the factor of a <scale>
          std::vector<int> affine, scale;
          for (int i = 0; i < L; ++i) {
            if (nnet.Layer(i)->GetType() == Component::kBiasedLinearity) affine.push_back(i);
            if (nnet.Layer(i)->GetType() == Component::kScale) scale.push_back(i);
          }
          if (!scale.empty() && (affine.empty() || g.below(3) == 0)) {
            dynamic_cast<Scale *>(nnet.Layer(scale[g.below(scale.size())]))->SetScale(0.25f * (1 + g.below(8)));
          } else if (!affine.empty()) {
            BiasedLinearity *bl = dynamic_cast<BiasedLinearity *>(nnet.Layer(affine[g.below(affine.size())]));
            const int O = bl->OutputDim(), I = bl->InputDim();
            DumpMat(Tag(step, "get_linearity"), bl->GetLinearityWeight());
            Vector<BaseFloat> bias;
            bl->GetBiasWeight().CopyToVec(&bias);
            std::ostringstream os;
            for (int i = 0; i < bias.Dim(); ++i) os << bias(i) << " ";
            DumpText(Tag(step, "get_bias"), os.str());
            const int how = g.below(6);
            const bool trans = how & 1;
            if (how < 4) {  // new weights from the host or from the device, as they are or transposed
              Matrix<BaseFloat> w(trans ? I : O, trans ? O : I);
              for (int i = 0; i < w.NumRows(); ++i)
                for (int j = 0; j < w.NumCols(); ++j) w(i, j) = 0.3f * g.sym();
              if (how < 2) {
                bl->SetLinearityWeight(w, trans);
              } else {
                CuMatrix<BaseFloat> dw;
                dw.CopyFromMat(w);
                bl->SetLinearityWeight(dw, trans);
              }
              Vector<BaseFloat> b(O);
              for (int i = 0; i < O; ++i) b(i) = 0.3f * g.sym();
              if (how & 2) {
                CuVector<BaseFloat> db;
                db.CopyFromVec(b);
                bl->SetBiasWeight(db);
              } else {
                bl->SetBiasWeight(b);
              }
            } else if (how == 4) {
              bl->SetToIdentity();
            }
            DumpMat(Tag(step, "set_linearity"), bl->GetLinearityWeight());
          }
          FillMat(&g, &in, rows, in_dim);
          nnet.Feedforward(in, &out);
          DumpMat(Tag(step, "accessor_feedforward"), out);
          break;
        }
        case 13: {  // the cache of the trainer towards target matrices (nnet/nnet-cache-tgtmat.h)
          CacheTgtMat cache;
          const int bunch = 1 + g.below(64), cachesize = bunch * (1 + g.below(6));
          cache.Init(cachesize, bunch);
          const bool shuffle = g.below(2);
          std::ostringstream log;
          for (int u = 0; u < 6 && !cache.Full(); ++u) {
            const int n = 1 + g.below(cachesize);
            CuMatrix<BaseFloat> feats, targets;
            FillMat(&g, &feats, n, 9);
            FillMat(&g, &targets, n, 5);
            cache.AddData(feats, targets);
            log << "add " << n << " full " << cache.Full() << " empty " << cache.Empty() << " | ";
          }
          if (shuffle && !cache.Empty()) cache.Randomize();
          log << "randomized " << cache.Randomized() << " | ";
          int bunches = 0;
          while (!cache.Empty() && bunches < 8) {
            CuMatrix<BaseFloat> feats, targets;
            cache.GetBunch(&feats, &targets);
            DumpMat(Tag(step, "tgtcache_bunch"), feats);
            DumpMat(Tag(step, "tgtcache_targets"), targets);
            ++bunches;
          }
          log << "bunches " << bunches << " empty " << cache.Empty() << " full " << cache.Full();
          DumpText(Tag(step, "tgtcache_log"), log.str());
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
