// Kernels of the fp32 MFMA GEMM for one epilogue (see gemm_kernel.h): the grouped gradient launch.
#define KHG_DEFINE_KERNELS
#define KHG_DEFINE_GRAD_GROUP
#include "gemm_kernel.h"
