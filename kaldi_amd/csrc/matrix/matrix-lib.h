#ifndef KALDI_AMD_MATRIX_MATRIX_LIB_H_
#define KALDI_AMD_MATRIX_MATRIX_LIB_H_
#include "matrix/kaldi-matrix.h"
#include "matrix/kaldi-vector.h"
#endif
