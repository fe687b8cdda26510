// Kernels of the fp32 MFMA GEMM for one epilogue (see gemm_kernel.h): the chained forward launch.
#define KHG_DEFINE_KERNELS
#define KHG_DEFINE_CHAIN
#include "gemm_kernel.h"
