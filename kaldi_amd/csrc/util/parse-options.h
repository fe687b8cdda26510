// Command-line option parser with the reference's command-line conventions
// (util/parse-options.{h,cc}): "--name=value" options before positional arguments, "--flag"
// alone means true for booleans, '_' and '-' are interchangeable and names are case-insensitive,
// "--config=file" (one --option=value per line, # comments), "--help", "--print-args=false",
// "--verbose=N".  Unknown options and malformed values print the usage and exit(1).
#ifndef KALDI_AMD_UTIL_PARSE_OPTIONS_H_
#define KALDI_AMD_UTIL_PARSE_OPTIONS_H_
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "base/kaldi-common.h"

namespace kaldi {

class ParseOptions {
 public:
  explicit ParseOptions(const char *usage);

  void Register(const std::string &name, bool *ptr, const std::string &doc) { Add(name, ptr, doc, false); }
  void Register(const std::string &name, int32 *ptr, const std::string &doc) { Add(name, ptr, doc, false); }
  void Register(const std::string &name, uint32 *ptr, const std::string &doc) { Add(name, ptr, doc, false); }
  void Register(const std::string &name, float *ptr, const std::string &doc) { Add(name, ptr, doc, false); }
  void Register(const std::string &name, double *ptr, const std::string &doc) { Add(name, ptr, doc, false); }
  void Register(const std::string &name, std::string *ptr, const std::string &doc) { Add(name, ptr, doc, false); }

  /// Parses argv; returns the index of the first positional argument.  exit(0) after --help,
  /// exit(1) on an invalid option.
  int Read(int argc, const char *const *argv);
  void PrintUsage(bool print_command_line = false);
  int NumArgs() const { return static_cast<int>(positional_.size()); }
  /// 1-based, like the reference
  std::string GetArg(int i) const;
  std::string GetOptArg(int i) const { return i <= NumArgs() ? GetArg(i) : std::string(); }

 private:
  struct Option {
    std::string name, doc;
    bool standard;
    std::function<bool(const std::string &)> set;  // false = malformed value
  };
  void Add(const std::string &name, bool *p, const std::string &doc, bool standard);
  void Add(const std::string &name, int32 *p, const std::string &doc, bool standard);
  void Add(const std::string &name, uint32 *p, const std::string &doc, bool standard);
  void Add(const std::string &name, float *p, const std::string &doc, bool standard);
  void Add(const std::string &name, double *p, const std::string &doc, bool standard);
  void Add(const std::string &name, std::string *p, const std::string &doc, bool standard);
  void Insert(const std::string &name, const std::string &doc, bool standard, const std::string &def,
              const char *type, std::function<bool(const std::string &)> set);
  bool Apply(const std::string &arg, const std::string &where);
  void ReadConfigFile(const std::string &filename);
  static std::string Normalize(const std::string &name);

  std::string usage_;
  std::map<std::string, Option> options_;  // keyed by normalized name
  std::vector<std::string> positional_;
  std::vector<std::string> argv_;
  bool print_args_, help_;
  std::string config_;
};

}  // namespace kaldi
#endif
