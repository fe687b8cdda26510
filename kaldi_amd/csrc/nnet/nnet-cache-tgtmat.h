// CacheTgtMat (feature cache with matrix targets, reference nnet/nnet-cache-tgtmat.h) lives in
// nnet-cache.h here; this header keeps the reference's include name working.
#ifndef KALDI_AMD_NNET_NNET_CACHE_TGTMAT_H_
#define KALDI_AMD_NNET_NNET_CACHE_TGTMAT_H_
#include "nnet/nnet-cache.h"
#endif
