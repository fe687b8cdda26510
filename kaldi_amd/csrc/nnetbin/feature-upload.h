// Uploads of per-utterance feature matrices that run beside the training kernels.
//
// The reference copies every utterance into one device matrix on the default stream
// (nnetbin/nnet-train-xent-hardlab-frmshuff.cc:141, cu-matrix-inl.h:173-196), which here would make
// the host wait for all the training queued before it.  Instead the rows go, on a stream of their
// own, into one of two device banks the size of a frame cache; the caller gets a window onto them
// and queues its consumers (feature transform, Cache::AddData) on the main stream as before.  A
// bank is overwritten only after the consumers of its previous contents have run (an event the
// host waits for when it comes back to that bank, one bank of uploads later).
#ifndef KALDI_AMD_NNETBIN_FEATURE_UPLOAD_H_
#define KALDI_AMD_NNETBIN_FEATURE_UPLOAD_H_
#include <algorithm>

#include "cudamatrix/cu-matrix.h"

namespace kaldi {

class FeatureUploader {
 public:
  explicit FeatureUploader(int32 bank_rows) : bank_rows_(bank_rows), bank_(0), pos_(0) {
    consumed_[0] = consumed_[1] = NULL;
  }
  ~FeatureUploader() {
    for (int i = 0; i < 2; ++i)
      if (consumed_[i]) hipK_event_destroy(consumed_[i]);
  }
  /// Copies `mat` to the device and returns a window onto the copy.  The window is valid until the
  /// next call; work that reads it must be queued (on the main stream) before that call.
  const CuMatrix<BaseFloat> &Upload(const Matrix<BaseFloat> &mat) {
    CuDevice &dev = CuDevice::Instantiate();
    const int32 rows = mat.NumRows();
    if (rows == 0 || mat.NumCols() == 0) {
      view_.Resize(0, 0);
      return view_;
    }
    CuMatrix<BaseFloat> &cur = banks_[bank_];
    const bool fits = cur.NumCols() == mat.NumCols() && pos_ + rows <= cur.NumRows();
    if (!fits) {
      // everything queued on the main stream so far is what reads the bank we leave
      if (cur.NumRows() > 0) MarkConsumed(bank_);
      bank_ ^= 1;
      pos_ = 0;
      CuMatrix<BaseFloat> &nxt = banks_[bank_];
      if (nxt.NumCols() != mat.NumCols() || nxt.NumRows() < rows) {
        if (nxt.NumRows() > 0) dev.Synchronize();  // its consumers, before the memory goes away
        nxt.Resize(std::max(rows, bank_rows_), mat.NumCols());
        dev.Synchronize();  // the zero fill was queued on the main stream
      } else if (consumed_[bank_]) {
        // The host waits, not the copy stream: it happens once per bank, and the copy stream stays
        // free of waits on other streams (nothing it carries can then be held up by training work).
        CU_SAFE_CALL(hipK_event_synchronize(consumed_[bank_]));
      }
    }
    {
      CuStreamScope on_copy_stream(dev.CopyStream());
      banks_[bank_].CopyRowsFromMat(rows, mat, 0, pos_);  // returns when the rows are on the device
    }
    view_.AttachRows(banks_[bank_], pos_, rows);
    pos_ += rows;
    return view_;
  }

 private:
  void MarkConsumed(int b) {
    CuDevice &dev = CuDevice::Instantiate();
    if (consumed_[b] == NULL) CU_SAFE_CALL(hipK_event_create(&consumed_[b]));
    CU_SAFE_CALL(hipK_event_record(consumed_[b], dev.MainStream()));
  }
  int32 bank_rows_;
  int bank_;
  int32 pos_;
  CuMatrix<BaseFloat> banks_[2], view_;
  khEvent_t consumed_[2];
};

}  // namespace kaldi
#endif
