#include "util/parse-options.h"

#include <ctype.h>
#include <stdlib.h>

#include <algorithm>
#include <fstream>
#include <iomanip>
#include <sstream>

namespace kaldi {

static std::string TrimCopy(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// quotes an argument for display when it contains characters the shell would interpret
static std::string ShellQuote(const std::string &s) {
  static const char *plain = "[]~#^_-+=:.,/";
  bool ok = !s.empty();
  for (char c : s) ok = ok && (isalnum(static_cast<unsigned char>(c)) || strchr(plain, c));
  if (ok) return s;
  std::string out = "'";
  for (char c : s) {
    if (c == '\'') out += "'\\''";
    else out += c;
  }
  return out + "'";
}

ParseOptions::ParseOptions(const char *usage) : usage_(usage), print_args_(true), help_(false) {
  Add("config", &config_, "Configuration file with options", true);
  Add("print-args", &print_args_, "Print the command line arguments (to stderr)", true);
  Add("help", &help_, "Print out usage message", true);
  Add("verbose", &g_kaldi_verbose_level, "Verbose level (higher->more warnings)", true);
}

std::string ParseOptions::Normalize(const std::string &name) {
  std::string out;
  for (char c : name) out += (c == '_') ? '-' : static_cast<char>(tolower(static_cast<unsigned char>(c)));
  return out;
}

void ParseOptions::Insert(const std::string &name, const std::string &doc, bool standard, const std::string &def,
                          const char *type, std::function<bool(const std::string &)> set) {
  Option o;
  o.name = name;
  o.doc = doc + " (" + type + ", default = " + def + ")";
  o.standard = standard;
  o.set = set;
  options_[Normalize(name)] = o;
}

void ParseOptions::Add(const std::string &name, bool *p, const std::string &doc, bool standard) {
  Insert(name, doc, standard, *p ? "true" : "false", "bool", [p](const std::string &v) {
    std::string s = v;
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    if (s == "" || s == "true" || s == "t" || s == "1") *p = true;
    else if (s == "false" || s == "f" || s == "0") *p = false;
    else return false;
    return true;
  });
}
template <class T, class Conv>
static std::function<bool(const std::string &)> NumSetter(T *p, Conv conv) {
  return [p, conv](const std::string &v) {
    char *end = NULL;
    auto x = conv(v.c_str(), &end);
    if (end == v.c_str()) return false;
    *p = static_cast<T>(x);
    return true;
  };
}
void ParseOptions::Add(const std::string &name, int32 *p, const std::string &doc, bool standard) {
  Insert(name, doc, standard, std::to_string(*p), "int",
         NumSetter(p, [](const char *s, char **e) { return strtol(s, e, 0); }));
}
void ParseOptions::Add(const std::string &name, uint32 *p, const std::string &doc, bool standard) {
  Insert(name, doc, standard, std::to_string(*p), "uint",
         NumSetter(p, [](const char *s, char **e) { return strtoul(s, e, 0); }));
}
void ParseOptions::Add(const std::string &name, float *p, const std::string &doc, bool standard) {
  std::ostringstream d;
  d << *p;
  Insert(name, doc, standard, d.str(), "float", NumSetter(p, [](const char *s, char **e) { return strtod(s, e); }));
}
void ParseOptions::Add(const std::string &name, double *p, const std::string &doc, bool standard) {
  std::ostringstream d;
  d << *p;
  Insert(name, doc, standard, d.str(), "double", NumSetter(p, [](const char *s, char **e) { return strtod(s, e); }));
}
void ParseOptions::Add(const std::string &name, std::string *p, const std::string &doc, bool standard) {
  Insert(name, doc, standard, "\"" + *p + "\"", "string", [p](const std::string &v) {
    *p = v;
    return true;
  });
}

bool ParseOptions::Apply(const std::string &arg, const std::string &where) {
  // arg starts with "--"
  const size_t eq = arg.find('=');
  const std::string key = Normalize(arg.substr(2, eq == std::string::npos ? std::string::npos : eq - 2));
  const std::string value = eq == std::string::npos ? std::string() : TrimCopy(arg.substr(eq + 1));
  std::map<std::string, Option>::iterator it = options_.find(key);
  if (key.empty() || it == options_.end()) {
    std::cerr << "Invalid option " << arg << where << '\n';
    return false;
  }
  if (!it->second.set(value)) {
    std::cerr << "Invalid value for option " << arg << where << '\n';
    return false;
  }
  return true;
}

void ParseOptions::ReadConfigFile(const std::string &filename) {
  std::ifstream is(filename.c_str());
  if (!is.good()) {
    std::cerr << "Cannot open config file " << filename << '\n';
    exit(1);
  }
  std::string line;
  while (std::getline(is, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    line = TrimCopy(line);
    if (line.empty()) continue;
    if (line.compare(0, 2, "--") != 0 || !Apply(line, " in config file " + filename)) {
      if (line.compare(0, 2, "--") != 0) std::cerr << "Invalid option " << line << " in config file " << filename << '\n';
      PrintUsage(true);
      exit(1);
    }
  }
}

int ParseOptions::Read(int argc, const char *const *argv) {
  argv_.assign(argv, argv + argc);
  if (argc > 0) SetProgramName(argv[0]);
  // config files first, so that the command line overrides them
  int i;
  // (this pass looks at every argument, also behind the first positional one: util/parse-options.cc:277-290)
  for (i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a.compare(0, 2, "--") != 0) continue;
    const std::string key = Normalize(a.substr(2, a.find('=') == std::string::npos ? std::string::npos : a.find('=') - 2));
    if (key == "config") ReadConfigFile(TrimCopy(a.substr(a.find('=') == std::string::npos ? a.size() : a.find('=') + 1)));
    if (key == "help") {
      PrintUsage();
      exit(0);
    }
  }
  for (i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a.compare(0, 2, "--") != 0) break;
    if (!Apply(a, "")) {
      PrintUsage(true);
      exit(1);
    }
  }
  const int first_positional = i;
  for (; i < argc; ++i) positional_.push_back(argv[i]);
  if (print_args_) {
    std::ostringstream line;
    for (int j = 0; j < argc; ++j) line << ShellQuote(argv[j]) << " ";
    std::cerr << line.str() << '\n' << std::flush;
  }
  return first_positional;
}

std::string ParseOptions::GetArg(int i) const {
  if (i < 1 || i > NumArgs()) KALDI_ERR << "ParseOptions::GetArg, invalid index " << i;
  return positional_[i - 1];
}

void ParseOptions::PrintUsage(bool print_command_line) {
  std::cerr << '\n' << usage_ << '\n';
  bool header = false;
  for (const auto &kv : options_) {
    if (kv.second.standard) continue;
    if (!header) std::cerr << "Options:" << '\n';
    header = true;
    std::cerr << "  --" << std::setw(25) << std::left << kv.second.name << " : " << kv.second.doc << '\n';
  }
  if (header) std::cerr << '\n';
  std::cerr << "Standard options:" << '\n';
  for (const auto &kv : options_)
    if (kv.second.standard)
      std::cerr << "  --" << std::setw(25) << std::left << kv.second.name << " : " << kv.second.doc << '\n';
  std::cerr << '\n';
  if (print_command_line) {
    std::ostringstream line;
    line << "Command line was: ";
    for (const std::string &a : argv_) line << ShellQuote(a) << " ";
    std::cerr << line.str() << '\n' << std::flush;
  }
}

}  // namespace kaldi
