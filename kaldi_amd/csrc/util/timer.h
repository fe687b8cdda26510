#ifndef KALDI_AMD_UTIL_TIMER_H_
#define KALDI_AMD_UTIL_TIMER_H_
#include "base/timer.h"
#endif
