#include "base/io-funcs.h"

#include <ctype.h>
#include <string.h>

namespace kaldi {

static void CheckToken(const char *token) {
  KALDI_ASSERT(token != NULL && *token != '\0');
  for (const char *p = token; *p; ++p) KALDI_ASSERT(!::isspace(static_cast<unsigned char>(*p)));
}

void WriteToken(std::ostream &os, bool /*binary*/, const char *token) {
  CheckToken(token);
  os << token << " ";
  if (os.fail()) throw std::runtime_error("Write failure in WriteToken.");
}
void WriteToken(std::ostream &os, bool binary, const std::string &token) { WriteToken(os, binary, token.c_str()); }

void ReadToken(std::istream &is, bool binary, std::string *token) {
  KALDI_ASSERT(token != NULL);
  if (!binary) is >> std::ws;
  is >> *token;
  if (is.fail()) KALDI_ERR << "ReadToken, failed to read token at file position " << is.tellg();
  if (!isspace(is.peek()))
    KALDI_ERR << "ReadToken, expected space after token, saw instead " << static_cast<char>(is.peek())
              << ", at file position " << is.tellg();
  is.get();
}

int Peek(std::istream &is, bool binary) {
  if (!binary) is >> std::ws;
  return is.peek();
}

int PeekToken(std::istream &is, bool binary) {
  if (!binary) is >> std::ws;
  if (is.peek() != '<') return is.peek();
  is.get();
  const int ans = is.peek();
  if (!is.unget()) KALDI_WARN << "Error ungetting '<' in PeekToken";
  return ans;
}

void ExpectToken(std::istream &is, bool binary, const char *token) {
  CheckToken(token);
  const std::streampos start = is.tellg();
  if (!binary) is >> std::ws;
  std::string got;
  is >> got;
  is.get();
  if (is.fail()) KALDI_ERR << "Failed to read token [started at file position " << start << "], expected " << token;
  if (got != token) KALDI_ERR << "Expected token \"" << token << "\", got instead \"" << got << "\".";
}
void ExpectToken(std::istream &is, bool binary, const std::string &token) { ExpectToken(is, binary, token.c_str()); }

void WriteBasicType(std::ostream &os, bool binary, bool b) {
  os << (b ? "T" : "F");
  if (!binary) os << " ";
  if (os.fail()) KALDI_ERR << "Write failure in WriteBasicType<bool>";
}

void ReadBasicType(std::istream &is, bool binary, bool *b) {
  if (!binary) is >> std::ws;
  const int c = is.peek();
  if (c == 'T') *b = true;
  else if (c == 'F') *b = false;
  else KALDI_ERR << "Read failure in ReadBasicType<bool>, file position is " << is.tellg();
  is.get();
}

template <class F>
static void WriteFloat(std::ostream &os, bool binary, F f) {
  if (binary) {
    os.put(static_cast<char>(sizeof(F)));
    os.write(reinterpret_cast<const char *>(&f), sizeof(F));
  } else {
    os << f << " ";
  }
}
void WriteBasicType(std::ostream &os, bool binary, float f) { WriteFloat(os, binary, f); }
void WriteBasicType(std::ostream &os, bool binary, double f) { WriteFloat(os, binary, f); }

// either width on disk is accepted and converted
template <class F>
static void ReadFloat(std::istream &is, bool binary, F *f) {
  if (binary) {
    const int c = is.get();
    if (c == 4) {
      float v;
      is.read(reinterpret_cast<char *>(&v), 4);
      *f = static_cast<F>(v);
    } else if (c == 8) {
      double v;
      is.read(reinterpret_cast<char *>(&v), 8);
      *f = static_cast<F>(v);
    } else {
      KALDI_ERR << "ReadBasicType: expected float, saw " << c << ", at file position " << is.tellg();
    }
  } else {
    is >> *f;
  }
  if (is.fail()) KALDI_ERR << "ReadBasicType: failed to read, at file position " << is.tellg();
}
void ReadBasicType(std::istream &is, bool binary, float *f) { ReadFloat(is, binary, f); }
void ReadBasicType(std::istream &is, bool binary, double *f) { ReadFloat(is, binary, f); }

}  // namespace kaldi
