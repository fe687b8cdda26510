// Reads a (possibly damaged) table with the readers the tools use and ends with status 0 (read to the end), 1 (a
// reported error) -- never with a memory error: tests/fuzz/table_mutation.py builds this file and the host sources it
// needs with the address and undefined-behaviour sanitizers and feeds it archives with flipped, cut and inserted bytes.
//
// usage: table_mutation feats|ali|ali_random|vec <rspecifier>
#include <iostream>
#include <string>

#include "base/kaldi-common.h"
#include "util/common-utils.h"

using namespace kaldi;

int main(int argc, char *argv[]) {
  if (argc != 3) return 2;
  const std::string mode = argv[1], rspec = argv[2];
  try {
    long long total = 0;
    if (mode == "feats") {
      SequentialBaseFloatMatrixReader r(rspec);
      for (; !r.Done(); r.Next()) {
        const Matrix<BaseFloat> &m = r.Value();
        total += static_cast<long long>(m.NumRows()) * m.NumCols() + r.Key().size();
        if (m.NumRows() > 0 && m.NumCols() > 0) total += m(m.NumRows() - 1, m.NumCols() - 1) > 0;
      }
    } else if (mode == "vec") {
      SequentialBaseFloatVectorReader r(rspec);
      for (; !r.Done(); r.Next()) total += r.Value().Dim();
    } else if (mode == "ali") {
      SequentialInt32VectorReader r(rspec);
      for (; !r.Done(); r.Next()) total += r.Value().size();
    } else if (mode == "ali_random") {
      RandomAccessInt32VectorReader r(rspec);
      const char *keys[] = {"utt3", "utt0", "nothere", "utt5", "utt1", "utt1"};
      for (int i = 0; i < 6; ++i)
        if (r.HasKey(keys[i])) total += r.Value(keys[i]).size();
    } else {
      return 2;
    }
    std::cout << total << "\n";
    return 0;
  } catch (const std::exception &e) {
    return 1;
  }
}
