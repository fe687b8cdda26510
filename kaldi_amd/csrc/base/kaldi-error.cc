#include "base/kaldi-error.h"

#include <stdlib.h>
#include <string.h>

#include <exception>
#include <iostream>

namespace kaldi {

int32 g_kaldi_verbose_level = 0;
static std::string g_program_name;

const char *GetProgramName() { return g_program_name.c_str(); }

void SetProgramName(const char *argv0) {
  const char *slash = strrchr(argv0, '/');
  g_program_name = std::string(slash ? slash + 1 : argv0) + ":";
}

static const char *ShortFileName(const char *file) {
  const char *slash = strrchr(file, '/');
  return slash ? slash + 1 : file;
}

LogMessage::LogMessage(Kind kind, const char *func, const char *file, int32 line, int32 vlevel) : kind_(kind) {
  switch (kind) {
    case kLog: ss_ << "LOG ("; break;
    case kWarning: ss_ << "WARNING ("; break;
    case kError: ss_ << "ERROR ("; break;
    case kVlog: ss_ << "VLOG[" << vlevel << "] ("; break;
  }
  ss_ << GetProgramName() << func << "():" << ShortFileName(file) << ':' << line << ") ";
}

LogMessage::~LogMessage() noexcept(false) {
  std::cerr << ss_.str() << '\n';
  if (kind_ == kError && !std::uncaught_exception()) throw std::runtime_error(ss_.str());
}

void AssertFailure(const char *func, const char *file, int32 line, const char *cond) {
  std::cerr << "KALDI_ASSERT: at " << GetProgramName() << func << ':' << ShortFileName(file) << ':' << line
            << ", failed: " << cond << '\n';
  std::cerr.flush();
  abort();
}

}  // namespace kaldi
