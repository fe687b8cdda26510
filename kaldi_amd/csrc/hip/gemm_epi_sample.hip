// Kernels of the fp32 MFMA GEMM for one epilogue (see gemm_kernel.h).
#define KHG_DEFINE_KERNELS
#include "gemm_kernel.h"

namespace khg {
template void launch_layout<7>(hipStream_t, const GemmParams &, int, bool, bool, bool);
template void launch_reduce<7>(hipStream_t, const GemmParams &, const float *, int);  // (the entry point never splits K)
}  // namespace khg
