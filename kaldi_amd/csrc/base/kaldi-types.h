// Basic typedefs of the host layer.  Interface of base/kaldi-types.h (BaseFloat = float,
// reference base/kaldi-types.h:27).
#ifndef KALDI_AMD_BASE_KALDI_TYPES_H_
#define KALDI_AMD_BASE_KALDI_TYPES_H_
#include <stdint.h>

namespace kaldi {
typedef float BaseFloat;
typedef int8_t int8;
typedef int16_t int16;
typedef int32_t int32;
typedef int64_t int64;
typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef float float32;
typedef double double64;
typedef int32 MatrixIndexT;
}  // namespace kaldi

#define KALDI_DISALLOW_COPY_AND_ASSIGN(type) \
  type(const type &) = delete;               \
  type &operator=(const type &) = delete
#endif
