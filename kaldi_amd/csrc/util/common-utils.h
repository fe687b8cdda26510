#ifndef KALDI_AMD_UTIL_COMMON_UTILS_H_
#define KALDI_AMD_UTIL_COMMON_UTILS_H_
#include "base/kaldi-common.h"
#include "util/kaldi-io.h"
#include "util/kaldi-table.h"
#include "util/parse-options.h"
#include "util/table-types.h"
#endif
