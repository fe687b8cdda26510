// Glue between the C++ classes and the C ABI of libkaldi_hip.so.
#ifndef KALDI_AMD_CUDAMATRIX_CU_COMMON_H_
#define KALDI_AMD_CUDAMATRIX_CU_COMMON_H_
#include "base/kaldi-common.h"
#include "kaldi_hip.h"
#include "matrix/matrix-common.h"

namespace kaldi {
/// Turns a non-zero status of a device-layer call into a KALDI_ERR (an exception), the role
/// cuSafeCall played around CUDA calls (reference cudamatrix/cu-common.h:35-42).
inline void CuCheck(int status, const char *what) {
  if (status != KH_OK) KALDI_ERR << "GPU error in " << what << ": " << hipK_error_string(status) << " (" << status << ")";
}
}  // namespace kaldi
#define CU_SAFE_CALL(call) ::kaldi::CuCheck((call), #call)
#endif
