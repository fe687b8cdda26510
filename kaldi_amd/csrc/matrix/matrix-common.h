#ifndef KALDI_AMD_MATRIX_MATRIX_COMMON_H_
#define KALDI_AMD_MATRIX_MATRIX_COMMON_H_
#include <algorithm>
#include <istream>
#include <vector>

#include "base/kaldi-common.h"

namespace kaldi {
// numeric values of CblasNoTrans / CblasTrans, as in the reference (matrix/matrix-common.h)
typedef enum { kTrans = 112, kNoTrans = 111 } MatrixTransposeType;
typedef enum { kSetZero, kUndefined, kCopyData } MatrixResizeType;

template <typename Real> class Vector;
template <typename Real> class Matrix;

/// The payload of a binary matrix or vector: n floating-point numbers of in_bytes bytes each (4 or 8) from the stream
/// into *out, converted where the widths differ.  The count comes from a header that a damaged file can set to
/// anything, so beyond 16 M elements the buffer grows (doubling) as the bytes arrive: a stream that ends early is an
/// error before the memory is.  False, with *out empty, when the stream ends early.
template <typename Real>
bool ReadBinaryFloats(std::istream &is, size_t n, size_t in_bytes, std::vector<Real> *out) {
  const size_t at_once = static_cast<size_t>(1) << 24;
  if (n == 0) out->clear();
  std::vector<char> raw;
  size_t done = 0;
  // (no clear() in front: a buffer that is read into again and again is only filled with zeros where it grows)
  while (done < n) {
    const size_t want = std::min(n - done, std::max(done, at_once));
    out->resize(done + want);
    if (in_bytes == sizeof(Real)) {
      is.read(reinterpret_cast<char *>(&(*out)[done]), sizeof(Real) * want);
    } else {
      raw.resize(in_bytes * want);
      is.read(raw.data(), raw.size());
      if (!is.fail()) {
        if (in_bytes == 4) {
          const float *f = reinterpret_cast<const float *>(raw.data());
          for (size_t i = 0; i < want; ++i) (*out)[done + i] = static_cast<Real>(f[i]);
        } else {
          const double *d = reinterpret_cast<const double *>(raw.data());
          for (size_t i = 0; i < want; ++i) (*out)[done + i] = static_cast<Real>(d[i]);
        }
      }
    }
    if (is.fail()) {
      out->clear();
      return false;
    }
    done += want;
  }
  return true;
}
}  // namespace kaldi
#endif
