#include "util/kaldi-table.h"

#include <ctype.h>

namespace kaldi {

// util/text-utils.cc:86-97 of the reference: non-empty, and no ASCII character that is not printable or is a space;
// bytes above 127 pass (accented letters), except 255 (a form of space).
bool IsToken(const std::string &s) {
  if (s.empty()) return false;
  for (char ch : s) {
    const unsigned char c = static_cast<unsigned char>(ch);
    if (c == 255) return false;
    if (c < 128 && (!isprint(c) || isspace(c))) return false;
  }
  return true;
}

static std::string Strip(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

static std::vector<std::string> SplitCommas(const std::string &s) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    size_t b = s.find(',', a);
    out.push_back(Strip(s.substr(a, b == std::string::npos ? std::string::npos : b - a)));
    if (b == std::string::npos) break;
    a = b + 1;
  }
  return out;
}

bool ReadScriptFile(const std::string &rxfilename, bool print_warnings,
                    std::vector<std::pair<std::string, std::string> > *script_out) {
  Input in;
  if (!in.OpenTextMode(rxfilename)) {
    if (print_warnings) KALDI_WARN << "Error opening script file: " << PrintableRxfilename(rxfilename);
    return false;
  }
  std::string line;
  int lineno = 0;
  while (std::getline(in.Stream(), line)) {
    ++lineno;
    line = Strip(line);
    const size_t sp = line.find_first_of(" \t");
    if (line.empty() || sp == std::string::npos) {
      if (print_warnings)
        KALDI_WARN << "Invalid " << lineno << "'th line in script file " << PrintableRxfilename(rxfilename) << ":\"" << line
                   << '"';
      return false;
    }
    script_out->push_back(std::make_pair(line.substr(0, sp), Strip(line.substr(sp + 1))));
  }
  return true;
}

WspecifierType ClassifyWspecifier(const std::string &wspecifier, std::string *archive_wxfilename,
                                  std::string *script_wxfilename, WspecifierOptions *opts) {
  const size_t colon = wspecifier.find(':');
  if (colon == std::string::npos) return kNoWspecifier;
  const std::string rest = wspecifier.substr(colon + 1);
  if (!rest.empty() && isspace(static_cast<unsigned char>(rest[rest.size() - 1]))) return kNoWspecifier;
  WspecifierOptions o;
  bool ark = false, scp = false;
  std::vector<std::string> parts = SplitCommas(wspecifier.substr(0, colon));
  for (size_t i = 0; i < parts.size(); ++i) {
    const std::string &p = parts[i];
    if (p == "b") o.binary = true;
    else if (p == "t") o.binary = false;
    else if (p == "f") o.flush = true;
    else if (p == "nf") o.flush = false;
    else if (p == "ark") {
      if (ark || scp) return kNoWspecifier;  // only "ark,scp" order is allowed
      ark = true;
    } else if (p == "scp") {
      if (scp) return kNoWspecifier;
      scp = true;
    } else return kNoWspecifier;
  }
  if (opts) *opts = o;
  if (ark && scp) {
    const size_t comma = rest.find(',');
    if (comma == std::string::npos) return kNoWspecifier;
    if (archive_wxfilename) *archive_wxfilename = Strip(rest.substr(0, comma));
    if (script_wxfilename) *script_wxfilename = Strip(rest.substr(comma + 1));
    return kBothWspecifier;
  }
  if (ark) {
    if (archive_wxfilename) *archive_wxfilename = rest;
    return kArchiveWspecifier;
  }
  if (scp) {
    if (script_wxfilename) *script_wxfilename = rest;
    return kScriptWspecifier;
  }
  return kNoWspecifier;
}

RspecifierType ClassifyRspecifier(const std::string &rspecifier, std::string *rxfilename, RspecifierOptions *opts) {
  const size_t colon = rspecifier.find(':');
  if (colon == std::string::npos) return kNoRspecifier;
  const std::string rest = rspecifier.substr(colon + 1);
  if (!rest.empty() && isspace(static_cast<unsigned char>(rest[rest.size() - 1]))) return kNoRspecifier;
  RspecifierOptions o;
  RspecifierType type = kNoRspecifier;
  std::vector<std::string> parts = SplitCommas(rspecifier.substr(0, colon));
  for (size_t i = 0; i < parts.size(); ++i) {
    const std::string &p = parts[i];
    if (p == "b" || p == "t") continue;  // accepted so that wspecifier-style strings can be reused
    else if (p == "o") o.once = true;
    else if (p == "no") o.once = false;
    else if (p == "p") o.permissive = true;
    else if (p == "np") o.permissive = false;
    else if (p == "s") o.sorted = true;
    else if (p == "ns") o.sorted = false;
    else if (p == "cs") o.called_sorted = true;
    else if (p == "ncs") o.called_sorted = false;
    else if (p == "ark") {
      if (type != kNoRspecifier) return kNoRspecifier;
      type = kArchiveRspecifier;
    } else if (p == "scp") {
      if (type != kNoRspecifier) return kNoRspecifier;
      type = kScriptRspecifier;
    } else return kNoRspecifier;
  }
  if (type != kNoRspecifier) {
    if (rxfilename) *rxfilename = rest;
    if (opts) *opts = o;
  }
  return type;
}

}  // namespace kaldi
