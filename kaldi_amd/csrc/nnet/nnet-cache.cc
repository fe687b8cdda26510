#include "nnet/nnet-cache.h"

#include <algorithm>

namespace kaldi {

void Cache::Init(int32 cachesize, int32 bunchsize) {
  KALDI_ASSERT(cachesize > 0);
  KALDI_ASSERT(bunchsize > 0);
  KALDI_ASSERT(cachesize >= bunchsize);
  if ((cachesize % bunchsize) != 0) KALDI_ERR << "Non divisible cachesize by bunchsize";
  cachesize_ = cachesize;
  bunchsize_ = bunchsize;
  state_ = EMPTY;
  filling_pos_ = 0;
  emptying_pos_ = 0;
  randomized_ = false;
  targets_on_device_ = false;
}

void Cache::AddData(const CuMatrix<BaseFloat> &features, const std::vector<int32> &targets) {
  if (state_ == FULL) KALDI_ERR << "Cannot add data, cache already full";
  KALDI_ASSERT(features.NumRows() == static_cast<MatrixIndexT>(targets.size()));

  if (features_.NumRows() != static_cast<MatrixIndexT>(cachesize_)) {  // first use
    features_.Resize(cachesize_, features.NumCols());
    targets_.resize(cachesize_);
  }
  if (features.NumRows() > static_cast<MatrixIndexT>(cachesize_ / 2))
    KALDI_WARN << "Too long segment and small feature cache!"
               << " cachesize: " << cachesize_ << " segmentsize: " << features.NumRows();

  if (state_ == EMPTY) {
    state_ = FILLING;
    filling_pos_ = 0;
    targets_on_device_ = false;
    // rows that did not fit last time come first
    int32 leftover = leftover_rows_;
    if (leftover > static_cast<int32>(cachesize_)) {
      KALDI_WARN << "Too small feature cache: " << cachesize_ << ", truncating: " << leftover - cachesize_
                 << " frames from previous segment leftover";
      leftover = cachesize_;
    }
    if (leftover > 0) {
      features_.CopyRowsFromMat(leftover, features_leftover_, 0, 0);
      std::copy(targets_leftover_.begin(), targets_leftover_.begin() + leftover, targets_.begin());
      // The buffer keeps its memory (the next overflow is written behind this copy, on the same stream): freeing
      // it would wait for the whole device, i.e. for every training step still queued from the previous cache,
      // and the fill of this cache would no longer run beside them.
      leftover_rows_ = 0;
      targets_leftover_.clear();
      filling_pos_ += leftover;
    }
  }
  KALDI_ASSERT(state_ == FILLING);

  const int32 space = cachesize_ - filling_pos_;
  const int32 length = features.NumRows();
  const int32 fill = std::min(space, length);
  const int32 rest = length - fill;
  KALDI_ASSERT(space > 0);

  features_.CopyRowsFromMat(fill, features, 0, filling_pos_);
  std::copy(targets.begin(), targets.begin() + fill, targets_.begin() + filling_pos_);
  if (rest > 0) {
    features_leftover_.Resize(rest, features_.NumCols());
    features_leftover_.CopyRowsFromMat(rest, features, fill, 0);
    leftover_rows_ = rest;
    targets_leftover_.assign(targets.begin() + fill, targets.end());
  }
  filling_pos_ += fill;
  if (filling_pos_ == cachesize_) state_ = FULL;
}

void Cache::Randomize() {
  KALDI_ASSERT(state_ == FULL || state_ == FILLING);
  // The permutation scheme of std::random_shuffle(first, last, rng) with rng(n) = RandInt(0, n-1):
  // element i is exchanged with a uniformly chosen element of [0, i].
  std::vector<int32> mask(filling_pos_);
  for (size_t i = 0; i < filling_pos_; ++i) mask[i] = i;
  for (size_t i = 1; i < filling_pos_; ++i) {
    const size_t j = RandInt(0, i);
    if (i != j) std::swap(mask[i], mask[j]);
  }
  Randomize(mask);
}

void Cache::Randomize(const std::vector<int32> &randmask) {
  KALDI_ASSERT(state_ == FULL || state_ == FILLING);
  KALDI_ASSERT(randmask.size() == filling_pos_);
  features_random_.Resize(cachesize_, features_.NumCols());
  targets_random_.resize(cachesize_);
  randmask_ = randmask;
  for (size_t i = 0; i < filling_pos_; ++i) {
    KALDI_ASSERT(randmask_[i] >= 0 && static_cast<size_t>(randmask_[i]) < filling_pos_);
    targets_random_[i] = targets_[randmask_[i]];
  }
  randmask_device_.CopyFromVec(randmask_);
  cu::Randomize(features_, randmask_device_, &features_random_);
  randomized_ = true;
  targets_on_device_ = false;
}

void Cache::StartEmptying() {
  if (state_ == EMPTY) KALDI_ERR << "GetBunch on empty cache!!!";
  if (state_ == FULL || state_ == FILLING) {  // FILLING: the last, partially filled cache
    state_ = EMPTYING;
    emptying_pos_ = 0;
  }
  KALDI_ASSERT(state_ == EMPTYING);
}

void Cache::Advance() {
  emptying_pos_ += bunchsize_;
  // stop when fewer than bunchsize_ rows remain
  if (emptying_pos_ + bunchsize_ > filling_pos_) state_ = EMPTY;
}

void Cache::GetBunch(CuMatrix<BaseFloat> *features, std::vector<int32> *targets) {
  StartEmptying();
  features->Resize(bunchsize_, features_.NumCols());
  targets->resize(bunchsize_);
  const CuMatrix<BaseFloat> &src = randomized_ ? features_random_ : features_;
  const std::vector<int32> &tsrc = randomized_ ? targets_random_ : targets_;
  features->CopyRowsFromMat(bunchsize_, src, emptying_pos_, 0);
  std::copy(tsrc.begin() + emptying_pos_, tsrc.begin() + emptying_pos_ + bunchsize_, targets->begin());
  Advance();
}

void Cache::GetBunch(CuMatrix<BaseFloat> *features, CuStlVector<int32> *targets) {
  StartEmptying();
  if (!targets_on_device_) {  // once per cache fill
    targets_device_.CopyFromVec(randomized_ ? targets_random_ : targets_);
    targets_on_device_ = true;
  }
  features->Resize(bunchsize_, features_.NumCols());
  const CuMatrix<BaseFloat> &src = randomized_ ? features_random_ : features_;
  features->CopyRowsFromMat(bunchsize_, src, emptying_pos_, 0);
  targets->CopyFromCuVec(targets_device_, emptying_pos_, bunchsize_);
  Advance();
}

void Cache::GetBunchView(CuMatrix<BaseFloat> *features, CuStlVector<int32> *targets) {
  StartEmptying();
  if (!targets_on_device_) {  // once per cache fill
    targets_device_.CopyFromVec(randomized_ ? targets_random_ : targets_);
    targets_on_device_ = true;
  }
  const CuMatrix<BaseFloat> &src = randomized_ ? features_random_ : features_;
  features->AttachRows(src, emptying_pos_, bunchsize_);
  targets->AttachRange(targets_device_, emptying_pos_, bunchsize_);
  Advance();
}

// ---- CacheTgtMat -------------------------------------------------------------------------------

void CacheTgtMat::Init(int32 cachesize, int32 bunchsize) {
  KALDI_ASSERT(cachesize > 0 && bunchsize > 0 && cachesize >= bunchsize);
  if ((cachesize % bunchsize) != 0) KALDI_ERR << "Non divisible cachesize by bunchsize";
  cachesize_ = cachesize;
  bunchsize_ = bunchsize;
  state_ = EMPTY;
  filling_pos_ = 0;
  emptying_pos_ = 0;
  randomized_ = false;
}

void CacheTgtMat::AddData(const CuMatrix<BaseFloat> &features, const CuMatrix<BaseFloat> &targets) {
  if (state_ == FULL) KALDI_ERR << "Cannot add data, cache already full";
  KALDI_ASSERT(features.NumRows() == targets.NumRows());
  if (features_.NumRows() != static_cast<MatrixIndexT>(cachesize_)) {  // first use
    features_.Resize(cachesize_, features.NumCols());
    targets_.Resize(cachesize_, targets.NumCols());
  }
  if (features.NumRows() > static_cast<MatrixIndexT>(cachesize_ / 2))
    KALDI_WARN << "Too long segment and small feature cache!"
               << " cachesize: " << cachesize_ << " segmentsize: " << features.NumRows();
  if (state_ == EMPTY) {
    state_ = FILLING;
    filling_pos_ = 0;
    int32 leftover = leftover_rows_;
    if (leftover > static_cast<int32>(cachesize_)) {
      KALDI_WARN << "Too small feature cache: " << cachesize_ << ", truncating: " << leftover - cachesize_
                 << " frames from previous segment leftover";
      leftover = cachesize_;
    }
    if (leftover > 0) {
      features_.CopyRowsFromMat(leftover, features_leftover_, 0, 0);
      targets_.CopyRowsFromMat(leftover, targets_leftover_, 0, 0);
      leftover_rows_ = 0;  // (both buffers keep their memory: see Cache::AddData)
      filling_pos_ += leftover;
    }
  }
  KALDI_ASSERT(state_ == FILLING);
  const int32 space = cachesize_ - filling_pos_;
  const int32 length = features.NumRows();
  const int32 fill = std::min(space, length);
  const int32 rest = length - fill;
  KALDI_ASSERT(space > 0);
  features_.CopyRowsFromMat(fill, features, 0, filling_pos_);
  targets_.CopyRowsFromMat(fill, targets, 0, filling_pos_);
  if (rest > 0) {
    features_leftover_.Resize(rest, features_.NumCols());
    features_leftover_.CopyRowsFromMat(rest, features, fill, 0);
    targets_leftover_.Resize(rest, targets_.NumCols());
    targets_leftover_.CopyRowsFromMat(rest, targets, fill, 0);
    leftover_rows_ = rest;
  }
  filling_pos_ += fill;
  if (filling_pos_ == cachesize_) state_ = FULL;
}

void CacheTgtMat::Randomize() {
  KALDI_ASSERT(state_ == FULL || state_ == FILLING);
  std::vector<int32> mask(filling_pos_);
  for (size_t i = 0; i < filling_pos_; ++i) mask[i] = i;
  for (size_t i = 1; i < filling_pos_; ++i) {  // std::random_shuffle's scheme, as in Cache::Randomize
    const size_t j = RandInt(0, i);
    if (i != j) std::swap(mask[i], mask[j]);
  }
  Randomize(mask);
}

void CacheTgtMat::Randomize(const std::vector<int32> &randmask) {
  KALDI_ASSERT(state_ == FULL || state_ == FILLING);
  KALDI_ASSERT(randmask.size() == filling_pos_);
  features_random_.Resize(cachesize_, features_.NumCols());
  targets_random_.Resize(cachesize_, targets_.NumCols());
  randmask_ = randmask;
  randmask_device_.CopyFromVec(randmask_);
  cu::Randomize(features_, randmask_device_, &features_random_);
  cu::Randomize(targets_, randmask_device_, &targets_random_);
  randomized_ = true;
}

void CacheTgtMat::GetBunch(CuMatrix<BaseFloat> *features, CuMatrix<BaseFloat> *targets) {
  if (state_ == EMPTY) KALDI_ERR << "GetBunch on empty cache!!!";
  if (state_ == FULL || state_ == FILLING) {
    state_ = EMPTYING;
    emptying_pos_ = 0;
  }
  KALDI_ASSERT(state_ == EMPTYING);
  features->Resize(bunchsize_, features_.NumCols());
  targets->Resize(bunchsize_, targets_.NumCols());
  features->CopyRowsFromMat(bunchsize_, randomized_ ? features_random_ : features_, emptying_pos_, 0);
  targets->CopyRowsFromMat(bunchsize_, randomized_ ? targets_random_ : targets_, emptying_pos_, 0);
  emptying_pos_ += bunchsize_;
  if (emptying_pos_ + bunchsize_ > filling_pos_) state_ = EMPTY;
}

}  // namespace kaldi
