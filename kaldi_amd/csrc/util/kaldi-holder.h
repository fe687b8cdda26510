// Holders: how one table value is laid out inside an archive entry (reference
// util/kaldi-holder-inl.h:37-98 for Kaldi objects, :189-286 for vectors of basic types).
//   Kaldi object : ["\0B"] + object.Write(binary)          (text: the object's text form)
//   basic vector : binary "\0B" int32 n, then n basic values; text: "v0 v1 ... \n"
#ifndef KALDI_AMD_UTIL_KALDI_HOLDER_H_
#define KALDI_AMD_UTIL_KALDI_HOLDER_H_
#include <algorithm>
#include <functional>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

#include "base/kaldi-common.h"
#include "matrix/kaldi-matrix.h"

namespace kaldi {

template <class KaldiType>
class KaldiObjectHolder {
 public:
  typedef KaldiType T;
  static bool Write(std::ostream &os, bool binary, const T &t) {
    InitKaldiOutputStream(os, binary);
    try {
      t.Write(os, binary);
      return os.good();
    } catch (const std::exception &e) {
      KALDI_WARN << "Exception caught writing Table object: " << e.what();
      return false;
    }
  }
  bool Read(std::istream &is) {
    bool binary;
    if (!InitKaldiInputStream(is, &binary)) {
      KALDI_WARN << "Reading Table object, failed reading binary header";
      return false;
    }
    try {
      t_.Read(is, binary);
      return true;
    } catch (const std::exception &) {
      KALDI_WARN << "Exception caught reading Table object";
      return false;
    }
  }
  /// Read, but only if keep(rows) says so (multi-rank tools: every rank walks the same table and keeps its own
  /// utterances).  For a binary matrix the decision is taken on the header and a payload nobody wants is skipped
  /// unread -- a seek on a file, a discard on a pipe; anything else is read and then dropped.  *kept tells whether
  /// Value() now holds the entry, *payload_read how many payload bytes were brought in from the stream.
  bool ReadIf(std::istream &is, const std::function<bool(int32)> &keep, bool *kept, size_t *payload_read) {
    bool binary;
    if (!InitKaldiInputStream(is, &binary)) {
      KALDI_WARN << "Reading Table object, failed reading binary header";
      return false;
    }
    try {
      if constexpr (std::is_same<T, Matrix<float> >::value || std::is_same<T, Matrix<double> >::value) {
        if (binary) {
          std::string tok;
          int32 r, c;
          size_t elem;
          T::ReadBinaryHeader(is, &tok, &r, &c, &elem);
          const std::streamoff bytes = static_cast<std::streamoff>(r) * c * elem;
          *kept = keep(r);
          if (*kept) {
            t_.ReadBinaryPayload(is, tok, r, c);
            *payload_read += bytes;
          } else if (bytes > 0) {
            // The skipped bytes must exist: a table cut off inside another rank's entry is as truncated for this
            // rank as for the one that reads it.  A seek past the end of a file succeeds, so the last skipped byte
            // is read; a discard that runs into the end of a pipe sets only eofbit, so its count is checked.
            const std::istream::pos_type here = is.tellg();
            if (here != std::istream::pos_type(-1)) {
              is.seekg(here + (bytes - 1));
              if (is.fail() || is.get() == std::istream::traits_type::eof()) {
                KALDI_WARN << "Reading Table object, truncated inside a skipped entry";
                return false;
              }
            } else {
              is.clear();
              is.ignore(bytes);  // not seekable: the bytes have to pass through anyway
              if (is.gcount() != bytes) {
                KALDI_WARN << "Reading Table object, truncated inside a skipped entry";
                return false;
              }
              *payload_read += bytes;
            }
            if (is.fail()) return false;
          }
          return true;
        }
      }
      const std::istream::pos_type before = is.tellg();
      t_.Read(is, binary);
      const std::istream::pos_type after = is.tellg();
      if (before != std::istream::pos_type(-1) && after != std::istream::pos_type(-1)) *payload_read += after - before;
      *kept = keep(RowsOf(t_));
      return true;
    } catch (const std::exception &) {
      KALDI_WARN << "Exception caught reading Table object";
      return false;
    }
  }
  const T &Value() const { return t_; }
  void Clear() { t_ = T(); }

 private:
  template <class U>
  static auto RowsOf(const U &u) -> decltype(u.NumRows(), int32()) { return u.NumRows(); }
  static int32 RowsOf(...) { return 0; }
  T t_;
};

template <class BasicType>
class BasicVectorHolder {
 public:
  typedef std::vector<BasicType> T;
  static bool Write(std::ostream &os, bool binary, const T &t) {
    InitKaldiOutputStream(os, binary);
    try {
      if (binary) WriteBasicType(os, binary, static_cast<int32>(t.size()));
      for (size_t i = 0; i < t.size(); ++i) WriteBasicType(os, binary, t[i]);
      if (!binary) os << '\n';
      return os.good();
    } catch (const std::exception &e) {
      KALDI_WARN << "Exception caught writing Table object (BasicVector): " << e.what();
      return false;
    }
  }
  bool Read(std::istream &is) {
    t_.clear();
    bool binary;
    if (!InitKaldiInputStream(is, &binary)) {
      KALDI_WARN << "Reading Table object [integer type], failed reading binary header";
      return false;
    }
    try {
      if (binary) {
        int32 n;
        ReadBasicType(is, true, &n);
        if (n < 0) return false;
        // (the count comes from the file: the vector grows as the elements arrive, so a damaged count is a read
        // error, not an allocation)
        t_.reserve(std::min<int32>(n, 1 << 20));
        for (int32 i = 0; i < n; ++i) {
          BasicType v;
          ReadBasicType(is, true, &v);
          t_.push_back(v);
        }
      } else {
        std::string line;
        std::getline(is, line);
        if (is.fail()) {
          KALDI_WARN << "BasicVectorHolder::Read, error reading line " << (is.eof() ? "[eof]" : "");
          return false;
        }
        std::istringstream ls(line);
        for (;;) {
          ls >> std::ws;
          if (ls.eof()) break;
          BasicType v;
          ReadBasicType(ls, false, &v);
          t_.push_back(v);
        }
      }
      return true;
    } catch (const std::exception &) {
      KALDI_WARN << "BasicVectorHolder::Read, read error or unexpected data";
      return false;
    }
  }
  /// The filtered read of the sequential reader (see KaldiObjectHolder::ReadIf): a vector is read whole and then
  /// kept or dropped by its length.
  bool ReadIf(std::istream &is, const std::function<bool(int32)> &keep, bool *kept, size_t *payload_read) {
    if (!Read(is)) return false;
    *kept = keep(static_cast<int32>(t_.size()));
    if (payload_read) *payload_read += t_.size() * sizeof(BasicType);
    if (!*kept) t_.clear();
    return true;
  }
  const T &Value() const { return t_; }
  void Clear() { t_.clear(); }

 private:
  T t_;
};

}  // namespace kaldi
#endif
