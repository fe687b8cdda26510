#ifndef KALDI_AMD_MATRIX_MATRIX_COMMON_H_
#define KALDI_AMD_MATRIX_MATRIX_COMMON_H_
#include "base/kaldi-common.h"

namespace kaldi {
// numeric values of CblasNoTrans / CblasTrans, as in the reference (matrix/matrix-common.h)
typedef enum { kTrans = 112, kNoTrans = 111 } MatrixTransposeType;
typedef enum { kSetZero, kUndefined, kCopyData } MatrixResizeType;

template <typename Real> class Vector;
template <typename Real> class Matrix;
}  // namespace kaldi
#endif
