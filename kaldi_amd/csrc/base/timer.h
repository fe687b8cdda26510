// Wall-clock timer (interface of util/timer.h:52-72).
#ifndef KALDI_AMD_BASE_TIMER_H_
#define KALDI_AMD_BASE_TIMER_H_
#include <chrono>

namespace kaldi {
class Timer {
 public:
  Timer() { Reset(); }
  void Reset() { start_ = std::chrono::steady_clock::now(); }
  double Elapsed() const {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - start_).count();
  }

 private:
  std::chrono::steady_clock::time_point start_;
};
}  // namespace kaldi
#endif
