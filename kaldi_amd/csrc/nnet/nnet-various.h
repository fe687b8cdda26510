// <expand> (frame splicing) and <copy> (column selection), the two data-rearranging components
// used inside feature transforms.  Public interface of the reference's nnet/nnet-various.h:35-135;
// on disk both keep their integer lists as a Vector<double>, <copy> with 1-based indices.
#ifndef KALDI_AMD_NNET_NNET_VARIOUS_H_
#define KALDI_AMD_NNET_NNET_VARIOUS_H_
#include <math.h>

#include <vector>

#include "cudamatrix/cu-math.h"
#include "nnet/nnet-component.h"

namespace kaldi {

class Expand : public Component {
 public:
  Expand(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet) {}
  ComponentType GetType() const { return kExpand; }
  void ReadData(std::istream &is, bool binary) {
    Vector<double> v;
    v.Read(is, binary);
    std::vector<int32> offsets(v.Dim());
    for (int32 i = 0; i < v.Dim(); ++i) offsets[i] = static_cast<int32>(round(v(i)));
    if (static_cast<MatrixIndexT>(offsets.size()) * input_dim_ != output_dim_)
      KALDI_ERR << "<expand>: " << offsets.size() << " offsets x input dim " << input_dim_ << " != output dim " << output_dim_;
    frame_offsets_.CopyFromVec(offsets);
  }
  void WriteData(std::ostream &os, bool binary) const {
    std::vector<int32> offsets;
    frame_offsets_.CopyToVec(&offsets);
    Vector<double> v(offsets.size());
    for (size_t i = 0; i < offsets.size(); ++i) v(i) = offsets[i];
    v.Write(os, binary);
  }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) { cu::Expand(in, frame_offsets_, out); }
  void BackpropagateFnc(const CuMatrix<BaseFloat> &, CuMatrix<BaseFloat> *) { KALDI_ERR << __func__ << "Not implemented!"; }

 protected:
  CuStlVector<int32> frame_offsets_;
};

class Copy : public Component {
 public:
  Copy(MatrixIndexT dim_in, MatrixIndexT dim_out, Nnet *nnet) : Component(dim_in, dim_out, nnet) {}
  ComponentType GetType() const { return kCopy; }
  void ReadData(std::istream &is, bool binary) {
    Vector<double> v;
    v.Read(is, binary);
    std::vector<int32> idx(v.Dim());
    for (int32 i = 0; i < v.Dim(); ++i) idx[i] = static_cast<int32>(round(v(i) - 1.0));
    if (static_cast<MatrixIndexT>(idx.size()) != output_dim_)
      KALDI_ERR << "<copy>: " << idx.size() << " indices, output dim " << output_dim_;
    copy_from_indices_.CopyFromVec(idx);
  }
  void WriteData(std::ostream &os, bool binary) const {
    std::vector<int32> idx;
    copy_from_indices_.CopyToVec(&idx);
    Vector<double> v(idx.size());
    for (size_t i = 0; i < idx.size(); ++i) v(i) = idx[i] + 1.0;
    v.Write(os, binary);
  }
  void PropagateFnc(const CuMatrix<BaseFloat> &in, CuMatrix<BaseFloat> *out) { cu::Copy(in, copy_from_indices_, out); }
  void BackpropagateFnc(const CuMatrix<BaseFloat> &, CuMatrix<BaseFloat> *) { KALDI_ERR << __func__ << "Not implemented!"; }

 protected:
  CuStlVector<int32> copy_from_indices_;
};

}  // namespace kaldi
#endif
