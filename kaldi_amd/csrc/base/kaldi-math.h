// Random helpers with the reference's formulas (base/kaldi-math.h:88-100, kaldi-math.cc:36-61)
// over a reimplementation of the glibc rand() stream: the frame-shuffle permutation and the RNG
// seeds are drawn from these, so a run reproduces the reference's (unseeded) sequence.
#ifndef KALDI_AMD_BASE_KALDI_MATH_H_
#define KALDI_AMD_BASE_KALDI_MATH_H_
#include <math.h>
#include <stdlib.h>

#include <limits>

#include "base/kaldi-types.h"

#ifndef M_2PI
#define M_2PI 6.283185307179586476925286766559005
#endif
#define KALDI_ISNAN(x) std::isnan(x)
#define KALDI_ISINF(x) std::isinf(x)

namespace kaldi {

// The process-wide random stream.  The reference draws from libc rand() and never seeds it, so
// its sequence is glibc's for seed 1.  Rand() produces that same sequence from a private state
// (glibc's additive-feedback generator x[i] = x[i-3] + x[i-31], output x >> 1), so that other
// libraries in the process calling rand()/srand() cannot disturb it.
int Rand();
void SRand(unsigned int seed);

// uniform integer in [min_val, max_val] (modulus of one Rand() call; slightly biased)
int32 RandInt(int32 min_val, int32 max_val);

// uniform in (0,1): never exactly 0 or 1
inline float RandUniform() { return static_cast<float>((Rand() + 1.0) / (RAND_MAX + 2.0)); }

// Box-Muller, two uniform draws per sample, cosine branch
inline float RandGauss() {
  const double u = RandUniform();
  const double radius = sqrt(-2.0 * log(u));
  return static_cast<float>(radius * cos(2.0 * M_PI * RandUniform()));
}

}  // namespace kaldi
#endif
