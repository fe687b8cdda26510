#ifndef KALDI_AMD_BASE_KALDI_COMMON_H_
#define KALDI_AMD_BASE_KALDI_COMMON_H_
#include <assert.h>
#include <string.h>

#include <iostream>
#include <string>
#include <vector>

#include "base/io-funcs.h"
#include "base/kaldi-error.h"
#include "base/kaldi-math.h"
#include "base/kaldi-types.h"
#include "base/timer.h"
#endif
