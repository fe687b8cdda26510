#include "base/kaldi-math.h"

#include "base/kaldi-error.h"

namespace kaldi {

namespace {
// glibc random_r, TYPE_3: 31 words of state, taps 3 and 31, 310 warm-up draws.
struct GlibcRand {
  int32 r[34];
  int32 ring[31];
  int pos;
  GlibcRand() { Seed(1); }
  void Seed(unsigned int seed) {
    if (seed == 0) seed = 1;
    r[0] = static_cast<int32>(seed);
    for (int i = 1; i < 31; ++i) {
      // 16807 * r[i-1] mod (2^31 - 1) without overflow (Schrage)
      const int32 hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
      int32 word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      r[i] = word;
    }
    for (int i = 0; i < 31; ++i) ring[i] = r[i];
    // x[31..33] are copies of x[0..2], the recurrence starts at i = 34:
    // ring[pos] is x[i-31], ring[(pos+28)%31] is x[i-3]
    pos = 3;
    for (int i = 0; i < 310; ++i) Next();
  }
  uint32 Next() {
    const uint32 v = static_cast<uint32>(ring[pos]) + static_cast<uint32>(ring[(pos + 28) % 31]);
    ring[pos] = static_cast<int32>(v);
    pos = (pos + 1) % 31;
    return v;
  }
};
GlibcRand g_rand;
}  // namespace

int Rand() { return static_cast<int>(g_rand.Next() >> 1); }
void SRand(unsigned int seed) { g_rand.Seed(seed); }

int32 RandInt(int32 min_val, int32 max_val) {
  KALDI_ASSERT(max_val >= min_val);
  if (max_val == min_val) return min_val;
  const int32 span = max_val + 1 - min_val;
  return min_val + static_cast<int32>(Rand()) % span;
}

}  // namespace kaldi
