#include "base/kaldi-math.h"

#include "base/kaldi-error.h"

namespace kaldi {

int32 RandInt(int32 min_val, int32 max_val) {
  KALDI_ASSERT(max_val >= min_val);
  if (max_val == min_val) return min_val;
  const int32 span = max_val + 1 - min_val;
  return min_val + static_cast<int32>(rand()) % span;
}

}  // namespace kaldi
