// A seeded random walk over the public CuMatrix / CuVector / cu:: API: the SAME source is compiled against the
// reference's classes (CPU branch; oracle/Makefile -> oracle/_ref/bin/cumatrix_walk) and against this build's
// classes (kaldi_amd/csrc/Makefile -> kaldi_amd/bin/cumatrix_walk).  Every step applies one method to a small pool of
// matrices and vectors and appends the changed object to a dump file; tests/fuzz/cumatrix_walk.py runs both and
// compares the dumps step by step.  What it checks is the class boundary itself: argument order, which operand is
// scaled by alpha and which by beta, transposes, row / column conventions of the vector operands, Resize semantics.
//
// usage: cumatrix_walk <seed> <steps> <dump-file>   |   cumatrix_walk refused
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "base/kaldi-common.h"
#include "cudamatrix/cu-device.h"
#include "cudamatrix/cu-math.h"
#include "cudamatrix/cu-matrix.h"
#include "cudamatrix/cu-stlvector.h"
#include "cudamatrix/cu-vector.h"

using namespace kaldi;

namespace {
// a generator of our own: the walk must not depend on either code base's rand() usage
struct Lcg {
  unsigned long long s;
  explicit Lcg(unsigned long long seed) : s(seed * 2862933555777941757ULL + 3037000493ULL) {}
  unsigned next() {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return static_cast<unsigned>(s >> 33);
  }
  int below(int n) { return static_cast<int>(next() % static_cast<unsigned>(n)); }
  float unit() { return (next() & 0xffffff) / 16777216.0f; }        // [0, 1)
  float sym() { return 2.0f * unit() - 1.0f; }                       // [-1, 1)
};

FILE *g_out = NULL;
void DumpMat(const char *what, const CuMatrix<BaseFloat> &m) {
  Matrix<BaseFloat> h;
  m.CopyToMat(&h);
  const int r = h.NumRows(), c = h.NumCols();
  fprintf(g_out, "%s %d %d\n", what, r, c);
  for (int i = 0; i < r; ++i) fwrite(h.RowData(i), sizeof(BaseFloat), c, g_out);
  fputc('\n', g_out);
}
void DumpVec(const char *what, const CuVector<BaseFloat> &v) {
  Vector<BaseFloat> h;
  v.CopyToVec(&h);
  fprintf(g_out, "%s 1 %d\n", what, static_cast<int>(h.Dim()));
  fwrite(h.Data(), sizeof(BaseFloat), h.Dim(), g_out);
  fputc('\n', g_out);
}
// a text (what a stream operator or a text-mode Write produced) as a 1 x N record of byte values
void DumpBytes(const char *what, const std::string &text) {
  fprintf(g_out, "%s 1 %d\n", what, static_cast<int>(text.size()));
  for (size_t i = 0; i < text.size(); ++i) {
    const BaseFloat f = static_cast<unsigned char>(text[i]);
    fwrite(&f, sizeof(f), 1, g_out);
  }
  fputc('\n', g_out);
}
void FillMat(Lcg *g, CuMatrix<BaseFloat> *m, int r, int c, bool positive = false) {
  Matrix<BaseFloat> h(r, c);
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) h(i, j) = positive ? 0.25f + g->unit() : g->sym();
  m->CopyFromMat(h);
}
void FillVec(Lcg *g, CuVector<BaseFloat> *v, int n, bool positive = false) {
  Vector<BaseFloat> h(n);
  for (int i = 0; i < n; ++i) h(i) = positive ? 0.25f + g->unit() : g->sym();
  v->CopyFromVec(h);
}
}  // namespace

int main(int argc, char *argv[]) {
  if (argc == 2 && std::string(argv[1]) == "refused") {
    // a call both code bases refuse: a partial-row vector that reaches the last column
    // (cudamatrix/cu-matrix-inl.h:793-795); the process ends with the error message either way
    CuMatrix<BaseFloat> M(4, 8);
    CuVector<BaseFloat> part;
    part.Resize(3);
    M.AddVecToPartialRows(1.0f, 5, part, 1.0f);
    return 0;
  }
  if (argc != 4) {
    std::cerr << "usage: cumatrix_walk <seed> <steps> <dump-file>\n";
    return 2;
  }
  try {
    Lcg g(strtoull(argv[1], NULL, 10));
    const int steps = atoi(argv[2]);
    g_out = fopen(argv[3], "wb");
    if (!g_out) return 2;
#if HAVE_CUDA == 1
    CuDevice::Instantiate().SelectGpuId(-1);
#endif
    const int dims[] = {1, 3, 7, 16, 33, 64, 100, 257};
    for (int step = 0; step < steps; ++step) {
      const int R = dims[g.below(8)], C = dims[g.below(8)], K = dims[g.below(8)];
      CuMatrix<BaseFloat> A, B, M;
      CuVector<BaseFloat> vr, vc;  // vr: one element per row, vc: one per column
      FillMat(&g, &M, R, C);
      FillMat(&g, &A, R, C);
      FillVec(&g, &vr, R, true);
      FillVec(&g, &vc, C, true);
      const BaseFloat alpha = 0.5f + g.unit(), beta = (g.below(3) == 0) ? 0.0f : 0.25f + g.unit();
      const int op = g.below(44);
      char tag[64];
      snprintf(tag, sizeof(tag), "step%d_op%d", step, op);
      if (getenv("WALK_TRACE")) fprintf(stderr, "%s R=%d C=%d K=%d\n", tag, R, C, K);
      switch (op) {
        case 0: M.SetZero(); DumpMat(tag, M); break;
        case 1: M.Set(alpha); DumpMat(tag, M); break;
        case 2: {
          const int r = 1 + g.below(R), c = 1 + g.below(C);
          M.PartSet(alpha, g.below(R - r + 1), r, g.below(C - c + 1), c);
          DumpMat(tag, M);
          break;
        }
        case 3: M.Binarize(0.1f); DumpMat(tag, M); break;
        case 4: M.Add(alpha); DumpMat(tag, M); break;
        case 5: FillMat(&g, &M, R, C, true); M.ApplyLog(); DumpMat(tag, M); break;
        case 6: M.ApplyExp(); DumpMat(tag, M); break;
        case 7: {
          // CuMatrix::Power: the reference's CUDA kernel raises every element to the power, its CPU branch calls the
          // host Matrix::Power, which is the MATRIX power of a square matrix and asserts on any other
          // (cudamatrix/cu-matrix-inl.h:477-494).  This build follows the kernel; the reference-side build
          // (-DWALK_NO_MATRIX_POWER) dumps the input only and the comparer checks this build against input^1.5.
          FillMat(&g, &M, R, C, true);
          DumpMat(tag, M);
#ifndef WALK_NO_MATRIX_POWER
          M.Power(1.5f);
          DumpMat(tag, M);
#endif
          break;
        }
        case 8: M.Scale(alpha); DumpMat(tag, M); break;
        case 9: FillMat(&g, &M, R, C, true); M.InvertElements(); DumpMat(tag, M); break;
        case 10: M.MulElements(A); DumpMat(tag, M); break;
        case 11: FillMat(&g, &A, R, C, true); M.DivElements(A); DumpMat(tag, M); break;
        case 12: M.MulColsVec(vc); DumpMat(tag, M); break;
        case 13: M.MulRowsVec(vr); DumpMat(tag, M); break;
        case 14: M.DivColsVec(vc); DumpMat(tag, M); break;
        case 15: M.DivRowsVec(vr); DumpMat(tag, M); break;
        case 16: M.AddMat(alpha, A, beta); DumpMat(tag, M); break;
        case 17: M.AddVecToCols(alpha, vr, beta); DumpMat(tag, M); break;
        case 18: M.AddVecToRows(alpha, vc, beta); DumpMat(tag, M); break;
        case 19: {
          if (C < 2) break;  // the vector may not reach the last column (cudamatrix/cu-matrix-inl.h:793)
          const int n = 1 + g.below(C - 1), off = g.below(C - n);
          CuVector<BaseFloat> part;
          FillVec(&g, &part, n);
          M.AddVecToPartialRows(alpha, off, part, beta);
          DumpMat(tag, M);
          break;
        }
        case 20: {  // all four transpose combinations of AddMatMat
          const bool ta = g.below(2), tb = g.below(2);
          CuMatrix<BaseFloat> X, Y;
          FillMat(&g, &X, ta ? K : R, ta ? R : K);
          FillMat(&g, &Y, tb ? C : K, tb ? K : C);
          M.AddMatMat(alpha, X, ta ? kTrans : kNoTrans, Y, tb ? kTrans : kNoTrans, beta);
          DumpMat(tag, M);
          break;
        }
        case 21: M.LogAddExpMat(A); DumpMat(tag, M); break;
        case 22: M.ApplyRowL2UpperBound(0.4f * static_cast<BaseFloat>(C > 4 ? 2 : 1)); DumpMat(tag, M); break;
        case 23: {  // CopyRowsFromMat: a run of rows between matrices of the same width
          CuMatrix<BaseFloat> src;
          const int RS = dims[g.below(8)];
          FillMat(&g, &src, RS, C);
          const int r = 1 + g.below(R < RS ? R : RS);
          M.CopyRowsFromMat(r, src, g.below(RS - r + 1), g.below(R - r + 1));
          DumpMat(tag, M);
          break;
        }
        case 24: {  // Resize keeps nothing and zeroes; then CopyFromMat of a device matrix
          M.Resize(C, R);
          DumpMat(tag, M);
          CuMatrix<BaseFloat> copy;
          copy.CopyFromMat(A);
          DumpMat(tag, copy);
          break;
        }
        case 25: vc.AddRowSumMat(alpha, M, beta); DumpVec(tag, vc); break;
        case 26: vr.AddColSumMat(alpha, M, beta); DumpVec(tag, vr); break;
        case 27: {
          CuVector<BaseFloat> w;
          FillVec(&g, &w, C);
          vc.AddVec(alpha, w, beta);
          DumpVec(tag, vc);
          vc.Scale(alpha); vc.Add(beta); DumpVec(tag, vc);
          vc.ApplyFloor(0.9f); DumpVec(tag, vc);
          vc.ApplyTruncate(0.95f, 1.4f); DumpVec(tag, vc);
          vc.MulElements(w); DumpVec(tag, vc);
          FillVec(&g, &w, C, true);
          vc.DivElements(w); DumpVec(tag, vc);
          vc.Round(); DumpVec(tag, vc);
          break;
        }
        case 28: {
          CuMatrix<BaseFloat> Y(R, C), E(R, C);  // the cu:: functions write into matrices their caller has sized
          cu::Sigmoid(M, &Y); DumpMat(tag, Y);
          cu::DiffSigmoid(A, Y, &E); DumpMat(tag, E);
          break;
        }
        case 29: {
          CuMatrix<BaseFloat> Y(R, C);
          cu::Softmax(M, &Y); DumpMat(tag, Y);
          CuStlVector<int32> ids;
          cu::FindRowMaxId(Y, &ids);
          std::vector<int32> h;
          ids.CopyToVec(&h);
          fprintf(g_out, "%s_ids 1 %d\n", tag, static_cast<int>(h.size()));
          for (size_t i = 0; i < h.size(); ++i) {
            const BaseFloat f = h[i];
            fwrite(&f, sizeof(f), 1, g_out);
          }
          fputc('\n', g_out);
          break;
        }
        case 30: {  // DiffXent on posteriors
          CuMatrix<BaseFloat> Y(R, C);
          cu::Softmax(M, &Y);
          std::vector<int32> t(R);
          for (int i = 0; i < R; ++i) t[i] = g.below(C);
          CuStlVector<int32> tgt;
          tgt.CopyFromVec(t);
          CuVector<BaseFloat> lp;
          cu::DiffXent(tgt, &Y, &lp);
          DumpMat(tag, Y);
          DumpVec(tag, lp);
          break;
        }
        case 31: {  // the frame shuffle
          std::vector<int32> idx(R);
          for (int i = 0; i < R; ++i) idx[i] = g.below(R);
          CuStlVector<int32> di;
          di.CopyFromVec(idx);
          CuMatrix<BaseFloat> Y(R, C);
          cu::Randomize(M, di, &Y);
          DumpMat(tag, Y);
          break;
        }
        case 32: {  // splice and column selection
          const int n = 1 + g.below(5);
          std::vector<int32> off(n), sel(1 + g.below(2 * C));
          for (int i = 0; i < n; ++i) off[i] = g.below(9) - 4;
          for (size_t i = 0; i < sel.size(); ++i) sel[i] = g.below(C);
          CuStlVector<int32> doff, dsel;
          doff.CopyFromVec(off);
          dsel.CopyFromVec(sel);
          CuMatrix<BaseFloat> Y(R, C * n), Z(R, static_cast<int>(sel.size()));
          cu::Expand(M, doff, &Y);
          cu::Copy(M, dsel, &Z);
          DumpMat(tag, Y);
          DumpMat(tag, Z);
          break;
        }
        case 33: {  // the L1 step
          FillMat(&g, &A, R, C);
          M.Scale(0.01f);
          cu::RegularizeL1(&M, &A, 0.02f, 0.1f);
          DumpMat(tag, M);
          DumpMat(tag, A);
          break;
        }
        case 34: {  // a smaller matrix added into a window of a larger one
          const int r = 1 + g.below(R), c = 1 + g.below(C);
          CuMatrix<BaseFloat> small;
          FillMat(&g, &small, r, c);
          M.PartAddMat(alpha, small, g.below(R - r + 1), g.below(C - c + 1), beta);
          DumpMat(tag, M);
          break;
        }
        case 35: {  // a window of A copied out, then a whole matrix copied into a window of M
          const int r = 1 + g.below(R), c = 1 + g.below(C);
          CuMatrix<BaseFloat> sub;
          sub.CopyFromMat(A, g.below(R - r + 1), r, g.below(C - c + 1), c);
          DumpMat(tag, sub);
          sub.CopyToMat(&M, g.below(R - r + 1), g.below(C - c + 1));
          DumpMat(tag, M);
          break;
        }
        case 36: {
          CuMatrix<BaseFloat> Y(R, C), E(R, C);
          cu::Relu(M, &Y); DumpMat(tag, Y);
          cu::DiffRelu(A, Y, &E); DumpMat(tag, E);
          break;
        }
        case 37: {
          CuMatrix<BaseFloat> Y(R, C), E(R, C);
          cu::SoftRelu(M, &Y); DumpMat(tag, Y);
          cu::DiffSoftRelu(A, M, &E); DumpMat(tag, E);
          break;
        }
        case 38: {
          vc.Power(2.0f); DumpVec(tag, vc);
          vc.ApplyLog(); DumpVec(tag, vc);
          vc.ApplyExp(); DumpVec(tag, vc);
          vc.InvertElements(); DumpVec(tag, vc);
          vr.Set(alpha); DumpVec(tag, vr);
          vr.Resize(C); DumpVec(tag, vr);
          CuVector<BaseFloat> copy;
          copy.CopyFromVec(vc); DumpVec(tag, copy);
          vr.SetZero(); DumpVec(tag, vr);
          break;
        }
        case 39: {  // a destination too small for the window is resized to end at the window (and so zeroed)
          CuMatrix<BaseFloat> dst;
          const int r0 = g.below(R + 3), c0 = g.below(C + 3);
          FillMat(&g, &dst, 1 + g.below(R + 2), 1 + g.below(C + 2));
          A.CopyToMat(&dst, r0, c0);
          DumpMat(tag, dst);
          break;
        }
        case 40: {  // the integer vector
          const int n = 1 + g.below(300);
          CuStlVector<int32> v(n);
          std::vector<int32> h;
          v.CopyToVec(&h);  // zero after construction
          v.Set(7 - g.below(15));
          std::vector<int32> h2;
          v.CopyToVec(&h2);
          h.insert(h.end(), h2.begin(), h2.end());
          std::vector<int32> src(1 + g.below(300));
          for (size_t i = 0; i < src.size(); ++i) src[i] = g.below(100000) - 50000;
          v.CopyFromVec(src);  // resizes
          v.CopyToVec(&h2);
          h.insert(h.end(), h2.begin(), h2.end());
          h.push_back(static_cast<int32>(v.Dim()));
          std::ostringstream os;
          os << v;
          v.SetZero();
          v.CopyToVec(&h2);
          h.insert(h.end(), h2.begin(), h2.end());
          v.Resize(5);
          v.CopyToVec(&h2);
          h.insert(h.end(), h2.begin(), h2.end());
          fprintf(g_out, "%s 1 %d\n", tag, static_cast<int>(h.size()));
          for (size_t i = 0; i < h.size(); ++i) {
            const BaseFloat f = h[i];
            fwrite(&f, sizeof(f), 1, g_out);
          }
          fputc('\n', g_out);
          DumpBytes(tag, os.str());
          break;
        }
        case 41: {  // a matrix through text and binary streams; whole numbers, so that the text is the same text
          Matrix<BaseFloat> h(R, C);
          for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j) h(i, j) = g.below(2001) - 1000;
          M.CopyFromMat(h);
          std::ostringstream text, bin, shown;
          M.Write(text, false);
          M.Write(bin, true);
          shown << M;
          DumpBytes(tag, text.str());
          DumpBytes(tag, bin.str());
          DumpBytes(tag, shown.str());
          std::istringstream from_text(text.str()), from_bin(bin.str());
          CuMatrix<BaseFloat> X(3, 3), Y;
          X.Read(from_text, false);
          Y.Read(from_bin, true);
          DumpMat(tag, X);
          DumpMat(tag, Y);
          break;
        }
        case 42: {  // and a vector
          Vector<BaseFloat> h(C);
          for (int j = 0; j < C; ++j) h(j) = g.below(2001) - 1000;
          vc.CopyFromVec(h);
          std::ostringstream text, bin, shown;
          vc.Write(text, false);
          vc.Write(bin, true);
          shown << vc;
          DumpBytes(tag, text.str());
          DumpBytes(tag, bin.str());
          DumpBytes(tag, shown.str());
          std::istringstream from_text(text.str()), from_bin(bin.str());
          CuVector<BaseFloat> x, y;
          x.Resize(2);
          x.Read(from_text, false);
          y.Read(from_bin, true);
          DumpVec(tag, x);
          DumpVec(tag, y);
          break;
        }
        case 43: {  // what the objects say about their memory, giving it back, and using them again
          std::ostringstream os;
          os << M.NumRows() << " " << M.NumCols() << " " << M.RowSizeInBytes() << " " << (M.Stride() >= M.NumCols()) << " "
             << (M.SizeInBytes() == static_cast<size_t>(M.NumRows()) * M.StrideSizeInBytes()) << " "
             << (M.StrideSizeInBytes() == M.Stride() * static_cast<MatrixIndexT>(sizeof(BaseFloat))) << " "
             << (M.RowData(R - 1) - M.Data() == static_cast<ptrdiff_t>(R - 1) * M.Stride());
          CuStlVector<int32> iv(K);
          M.Destroy();
          vc.Destroy();
          iv.Destroy();
          os << " | " << M.NumRows() << " " << M.NumCols() << " " << M.Stride() << " " << vc.Dim() << " " << iv.Dim();
          M.Destroy();  // (twice is harmless)
          M.Resize(K, R);
          vc.Resize(K);
          iv.Resize(R);
          std::vector<int32> h;
          iv.CopyToVec(&h);
          os << " | " << M.NumRows() << " " << M.NumCols() << " " << vc.Dim() << " " << h.size() << " " << h[R - 1];
          DumpBytes(tag, os.str());
          DumpMat(tag, M);
          DumpVec(tag, vc);
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
