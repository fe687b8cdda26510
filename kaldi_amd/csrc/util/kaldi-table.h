// Keyed tables: sequential reader, random-access reader and writer over archives ("ark:") and
// script files ("scp:"), with the reference's specifier syntax (util/kaldi-table.h:98-203):
//   rspecifier  [opts,]ark:rxfilename | [opts,]scp:rxfilename      opts: o s cs p (+ n-forms), b t ignored
//   wspecifier  [opts,]ark:wxfilename | scp:wxfilename | ark,scp:archive,script     opts: b t f nf
// Archive layout: for each entry  "<key> " + holder payload.  Script layout: "<key> <xfilename>\n".
#ifndef KALDI_AMD_UTIL_KALDI_TABLE_H_
#define KALDI_AMD_UTIL_KALDI_TABLE_H_
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "util/kaldi-holder.h"
#include "util/kaldi-io.h"

namespace kaldi {

enum WspecifierType { kNoWspecifier, kArchiveWspecifier, kScriptWspecifier, kBothWspecifier };
struct WspecifierOptions {
  bool binary, flush;
  WspecifierOptions() : binary(true), flush(false) {}
};
WspecifierType ClassifyWspecifier(const std::string &wspecifier, std::string *archive_wxfilename,
                                  std::string *script_wxfilename, WspecifierOptions *opts);

enum RspecifierType { kNoRspecifier, kArchiveRspecifier, kScriptRspecifier };
struct RspecifierOptions {
  bool once, sorted, called_sorted, permissive;
  RspecifierOptions() : once(false), sorted(false), called_sorted(false), permissive(false) {}
};
RspecifierType ClassifyRspecifier(const std::string &rspecifier, std::string *rxfilename, RspecifierOptions *opts);

/// reads "key xfilename" lines; false on a malformed line
bool ReadScriptFile(const std::string &rxfilename, bool print_warnings,
                    std::vector<std::pair<std::string, std::string> > *script_out);
bool IsToken(const std::string &s);  // a valid table key (util/text-utils.cc:86 of the reference)

// ---- sequential reader ---------------------------------------------------------------------
template <class Holder>
class SequentialTableReader {
 public:
  typedef typename Holder::T T;
  SequentialTableReader() : type_(kNoRspecifier), done_(true), pos_(0) {}
  explicit SequentialTableReader(const std::string &rspecifier) : type_(kNoRspecifier), done_(true), pos_(0) {
    if (!Open(rspecifier)) KALDI_ERR << "Error constructing TableReader: rspecifier is " << rspecifier;
  }
  /// Entries are presented only where keep(key, rows) is true (rows = NumRows() of the value); set before Open.
  /// keep is called once per entry, in table order.  Holders that can (binary matrices) decide on the entry's
  /// header and skip the payload of the others unread: a rank of a multi-GPU run then reads 1/N of the table.
  typedef std::function<bool(const std::string &, int32)> Filter;
  SequentialTableReader(const std::string &rspecifier, const Filter &keep)
      : type_(kNoRspecifier), done_(true), pos_(0), keep_(keep) {
    if (!Open(rspecifier)) KALDI_ERR << "Error constructing TableReader: rspecifier is " << rspecifier;
  }
  /// bytes of value payload brought in from the table so far (with a filter: what skipping saved shows here)
  size_t PayloadBytesRead() const { return payload_read_; }
  bool Open(const std::string &rspecifier) {
    rspecifier_ = rspecifier;
    std::string rx;
    type_ = ClassifyRspecifier(rspecifier, &rx, &opts_);
    if (type_ == kArchiveRspecifier) {
      if (!input_.OpenTextMode(rx)) {
        KALDI_WARN << "TableReader: failed to open stream " << PrintableRxfilename(rx);
        return false;
      }
    } else if (type_ == kScriptRspecifier) {
      if (!ReadScriptFile(rx, true, &script_)) return false;
      pos_ = 0;
    } else {
      KALDI_WARN << "Invalid rspecifier " << rspecifier;
      return false;
    }
    done_ = false;
    Advance();
    return true;
  }
  bool IsOpen() const { return type_ != kNoRspecifier; }
  bool Done() const { return done_; }
  std::string Key() const {
    if (done_) KALDI_ERR << "TableReader: Key() called when Done()";
    return key_;
  }
  const T &Value() {
    if (done_) KALDI_ERR << "TableReader: Value() called when Done()";
    return holder_.Value();
  }
  void Next() {
    if (done_) KALDI_ERR << "TableReader: Next() called when Done()";
    Advance();
  }
  bool Close() {
    input_.Close();
    type_ = kNoRspecifier;
    done_ = true;
    return true;
  }

 private:
  void Advance() {
    if (type_ == kArchiveRspecifier) {
      std::istream &is = input_.Stream();
      is.clear();
      is >> key_;  // skips the whitespace left after the previous entry
      if (is.eof() || is.fail()) {
        done_ = true;
        return;
      }
      const int c = is.peek();
      if (c != ' ' && c != '\t' && c != '\n')
        KALDI_ERR << "TableReader: invalid archive file format: expected space after key " << key_ << ", got character "
                  << static_cast<char>(c) << ", reading " << rspecifier_;
      if (c != '\n') is.get();  // a bare newline belongs to an empty text-mode value
      if (keep_) {
        for (;;) {
          bool kept = false;
          const std::string key = key_;
          if (!holder_.ReadIf(is, [&](int32 rows) { return keep_(key, rows); }, &kept, &payload_read_)) {
            if (opts_.permissive) {
              KALDI_WARN << "TableReader: error reading object for key " << key_ << "; ignoring the rest of the archive";
              done_ = true;
              return;
            }
            KALDI_ERR << "TableReader: failed to read object from archive, key " << key_ << ", reading " << rspecifier_;
          }
          if (kept) return;
          is.clear();
          is >> key_;
          if (is.eof() || is.fail()) {
            done_ = true;
            return;
          }
          const int c2 = is.peek();
          if (c2 != ' ' && c2 != '\t' && c2 != '\n')
            KALDI_ERR << "TableReader: invalid archive file format: expected space after key " << key_ << ", reading "
                      << rspecifier_;
          if (c2 != '\n') is.get();
        }
      }
      if (!holder_.Read(is)) {
        if (opts_.permissive) {
          KALDI_WARN << "TableReader: error reading object for key " << key_ << "; ignoring the rest of the archive";
          done_ = true;
          return;
        }
        KALDI_ERR << "TableReader: failed to read object from archive, key " << key_ << ", reading " << rspecifier_;
      }
    } else {
      while (pos_ < script_.size()) {
        const std::pair<std::string, std::string> &e = script_[pos_++];
        key_ = e.first;
        Input in;
        if (keep_) {
          bool kept = false;
          const std::string key = key_;
          if (in.OpenTextMode(e.second) &&
              holder_.ReadIf(in.Stream(), [&](int32 rows) { return keep_(key, rows); }, &kept, &payload_read_)) {
            if (kept) return;
            continue;
          }
        } else if (in.OpenTextMode(e.second) && holder_.Read(in.Stream())) {
          return;
        }
        if (!opts_.permissive) KALDI_ERR << "TableReader: failed to load object from " << PrintableRxfilename(e.second);
        KALDI_WARN << "TableReader: skipping key " << key_ << ", cannot read " << PrintableRxfilename(e.second);
      }
      done_ = true;
    }
  }
  std::string rspecifier_, key_;
  RspecifierType type_;
  RspecifierOptions opts_;
  Input input_;
  Holder holder_;
  bool done_;
  std::vector<std::pair<std::string, std::string> > script_;
  size_t pos_;
  Filter keep_;
  size_t payload_read_ = 0;
};

// ---- sequential reader with a read-ahead thread ---------------------------------------------
// Same interface as SequentialTableReader; a background thread reads and parses entries ahead of
// the consumer into a bounded queue, so that file / pipe input overlaps the consumer's work (the
// training tools spend their time waiting for the GPU, not for the table: "feature wait",
// nnetbin/nnet-train-xent-hardlab-frmshuff.cc:153-155).  Errors raised while reading are handed to
// the consumer at the entry where they occurred.  Not in the reference.
template <class Holder>
class PrefetchingTableReader {
 public:
  typedef typename Holder::T T;
  explicit PrefetchingTableReader(const std::string &rspecifier, size_t max_ahead = 64)
      : reader_(rspecifier), max_ahead_(max_ahead), stop_(false), finished_(false), have_(false) {
    thread_ = std::thread(&PrefetchingTableReader::Run, this);
    Pop();
  }
  /// with an entry filter (see SequentialTableReader); keep runs on the read-ahead thread
  PrefetchingTableReader(const std::string &rspecifier, const typename SequentialTableReader<Holder>::Filter &keep,
                         size_t max_ahead = 64)
      : reader_(rspecifier, keep), max_ahead_(max_ahead), stop_(false), finished_(false), have_(false) {
    thread_ = std::thread(&PrefetchingTableReader::Run, this);
    Pop();
  }
  /// payload bytes read so far (only settled once Done())
  size_t PayloadBytesRead() const { return reader_.PayloadBytesRead(); }
  ~PrefetchingTableReader() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
    }
    cv_space_.notify_all();
    if (thread_.joinable()) thread_.join();
  }
  bool Done() const { return !have_; }
  std::string Key() const {
    if (!have_) KALDI_ERR << "TableReader: Key() called when Done()";
    return cur_->key;
  }
  const T &Value() {
    if (!have_) KALDI_ERR << "TableReader: Value() called when Done()";
    return cur_->value;
  }
  void Next() {
    if (!have_) KALDI_ERR << "TableReader: Next() called when Done()";
    Pop();
  }

 private:
  struct Item {
    std::string key;
    T value;
  };
  void Run() {
    try {
      for (; !reader_.Done(); reader_.Next()) {
        std::unique_ptr<Item> it(new Item);
        it->key = reader_.Key();
        it->value = reader_.Value();
        std::unique_lock<std::mutex> l(mu_);
        cv_space_.wait(l, [this] { return stop_ || queue_.size() < max_ahead_; });
        if (stop_) return;
        queue_.push_back(std::move(it));
        cv_item_.notify_one();
      }
    } catch (...) {
      std::lock_guard<std::mutex> l(mu_);
      error_ = std::current_exception();
    }
    std::lock_guard<std::mutex> l(mu_);
    finished_ = true;
    cv_item_.notify_all();
  }
  void Pop() {
    std::unique_lock<std::mutex> l(mu_);
    cv_item_.wait(l, [this] { return !queue_.empty() || finished_; });
    if (!queue_.empty()) {
      cur_ = std::move(queue_.front());
      queue_.pop_front();
      have_ = true;
      cv_space_.notify_one();
      return;
    }
    have_ = false;
    cur_.reset();
    if (error_) {
      std::exception_ptr e = error_;
      error_ = nullptr;
      std::rethrow_exception(e);
    }
  }
  SequentialTableReader<Holder> reader_;
  size_t max_ahead_;
  std::thread thread_;
  std::mutex mu_;
  std::condition_variable cv_item_, cv_space_;
  std::deque<std::unique_ptr<Item> > queue_;
  std::unique_ptr<Item> cur_;
  std::exception_ptr error_;
  bool stop_, finished_, have_;
};

// ---- random-access reader ------------------------------------------------------------------
// Archives are consumed lazily: entries are read until the requested key turns up and everything
// seen on the way is kept.  With the "s" (sorted) option a miss is decided as soon as a larger
// key has been read; with "o" (once) a value is dropped when the NEXT key is requested, so that
// the reference returned by Value() stays valid until then.
template <class Holder>
class RandomAccessTableReader {
 public:
  typedef typename Holder::T T;
  RandomAccessTableReader() : type_(kNoRspecifier), eof_(true) {}
  explicit RandomAccessTableReader(const std::string &rspecifier) : type_(kNoRspecifier), eof_(true) {
    if (!Open(rspecifier)) KALDI_ERR << "Error opening RandomAccessTableReader object (rspecifier is: " << rspecifier << ")";
  }
  bool Open(const std::string &rspecifier) {
    rspecifier_ = rspecifier;
    std::string rx;
    type_ = ClassifyRspecifier(rspecifier, &rx, &opts_);
    if (type_ == kArchiveRspecifier) {
      if (!input_.OpenTextMode(rx)) return false;
      eof_ = false;
    } else if (type_ == kScriptRspecifier) {
      std::vector<std::pair<std::string, std::string> > script;
      if (!ReadScriptFile(rx, true, &script)) return false;
      // A key that comes twice, or -- with the "s" option -- keys that are not in sorted order, refuse the table
      // (util/kaldi-table-inl.h:1229-1246).
      {
        std::vector<std::string> keys(script.size());
        for (size_t i = 0; i < script.size(); ++i) keys[i] = script[i].first;
        if (!opts_.sorted) std::sort(keys.begin(), keys.end());
        for (size_t i = 0; i + 1 < keys.size(); ++i) {
          if (keys[i].compare(keys[i + 1]) >= 0) {
            KALDI_WARN << "Script file " << PrintableRxfilename(rx)
                       << (keys[i] == keys[i + 1] ? " contains duplicate key: "
                                                  : " is not sorted (remove s, option or add ns, option): key is ")
                       << keys[i];
            return false;
          }
        }
      }
      for (size_t i = 0; i < script.size(); ++i) script_[script[i].first] = script[i].second;
      eof_ = true;
    } else {
      return false;
    }
    return true;
  }
  bool IsOpen() const { return type_ != kNoRspecifier; }
  bool HasKey(const std::string &key) {
    if (!IsToken(key)) KALDI_ERR << "RandomAccessTableReader::HasKey, invalid key \"" << key << '"';
    return Find(key) != NULL;
  }
  const T &Value(const std::string &key) {
    const T *t = Find(key);
    if (!t) KALDI_ERR << "RandomAccessTableReader: Value() called for key " << key << " which is not present, reading " << rspecifier_;
    return *t;
  }
  bool Close() {
    input_.Close();
    cache_.clear();
    type_ = kNoRspecifier;
    return true;
  }

 private:
  const T *Find(const std::string &key) {
    // "s" and "cs" together: the caller has promised to ask in sorted order (util/kaldi-table-inl.h:1639-1650)
    if (type_ == kArchiveRspecifier && opts_.sorted && opts_.called_sorted) {
      if (!last_requested_key_.empty() && key.compare(last_requested_key_) < 0)
        KALDI_ERR << "RandomAccessTableReader: you provided the \"cs\" option "
                  << "but are not calling with keys in sorted order: " << key << " < " << last_requested_key_;
      last_requested_key_ = key;
    }
    if (opts_.once && !last_key_.empty() && last_key_ != key) {
      cache_.erase(last_key_);
      last_key_.clear();
    }
    typename std::unordered_map<std::string, std::unique_ptr<T> >::iterator it = cache_.find(key);
    if (it != cache_.end()) {
      last_key_ = key;
      return it->second.get();
    }
    if (type_ == kScriptRspecifier) {
      std::unordered_map<std::string, std::string>::iterator s = script_.find(key);
      if (s == script_.end()) return NULL;
      Input in;
      Holder h;
      if (!in.OpenTextMode(s->second) || !h.Read(in.Stream())) {
        if (opts_.permissive) return NULL;
        KALDI_ERR << "RandomAccessTableReader: failed to load object from " << PrintableRxfilename(s->second);
      }
      cache_[key].reset(new T(h.Value()));
      last_key_ = key;
      return cache_[key].get();
    }
    if (opts_.sorted && !newest_key_.empty() && key < newest_key_) return NULL;  // would have been seen already
    while (!eof_) {
      std::istream &is = input_.Stream();
      std::string k;
      is.clear();
      is >> k;
      if (is.eof() || is.fail()) {
        eof_ = true;
        break;
      }
      const int c = is.peek();
      if (c != ' ' && c != '\t' && c != '\n') KALDI_ERR << "RandomAccessTableReader: invalid archive file format after key " << k;
      if (c != '\n') is.get();
      Holder h;
      if (!h.Read(is)) {
        if (opts_.permissive) {
          eof_ = true;
          break;
        }
        KALDI_ERR << "RandomAccessTableReader: failed to read object with key " << k << " from " << rspecifier_;
      }
      if (opts_.sorted) {
        // the archive was promised to be sorted (util/kaldi-table-inl.h:1820-1831, 1684-1688)
        if (!newest_key_.empty() && k.compare(newest_key_) <= 0)
          KALDI_ERR << "RandomAccessTableReader: you provide the sorted (s) option but keys in archive "
                    << PrintableRxfilename(rspecifier_) << " are not in sorted order: " << newest_key_ << " is followed by " << k;
      } else if (cache_.find(k) != cache_.end()) {
        // (util/kaldi-table-inl.h:1992-1996)
        KALDI_ERR << "Error in RandomAccessTableReader: duplicate key " << k << " in archive " << rspecifier_;
      }
      cache_[k].reset(new T(h.Value()));
      newest_key_ = k;
      if (k == key) {
        last_key_ = key;
        return cache_[k].get();
      }
      if (opts_.sorted && k > key) return NULL;
    }
    return NULL;
  }
  std::string rspecifier_, last_key_, newest_key_, last_requested_key_;
  RspecifierType type_;
  RspecifierOptions opts_;
  Input input_;
  bool eof_;
  std::unordered_map<std::string, std::unique_ptr<T> > cache_;
  std::unordered_map<std::string, std::string> script_;
};

// ---- writer ----------------------------------------------------------------------------------
template <class Holder>
class TableWriter {
 public:
  typedef typename Holder::T T;
  TableWriter() : type_(kNoWspecifier) {}
  explicit TableWriter(const std::string &wspecifier) : type_(kNoWspecifier) {
    if (!Open(wspecifier)) KALDI_ERR << "TableWriter: failed to write to " << wspecifier;
  }
  ~TableWriter() {
    if (IsOpen() && !Close()) KALDI_WARN << "Error closing TableWriter [in destructor].";
  }
  bool Open(const std::string &wspecifier) {
    wspecifier_ = wspecifier;
    type_ = ClassifyWspecifier(wspecifier, &archive_name_, &script_name_, &opts_);
    if (type_ == kNoWspecifier) {
      KALDI_WARN << "Invalid wspecifier " << wspecifier;
      return false;
    }
    if (type_ == kArchiveWspecifier || type_ == kBothWspecifier) {
      if (!archive_.Open(archive_name_, opts_.binary, false)) return false;
      if (type_ == kBothWspecifier && ClassifyWxfilename(archive_name_) != kFileOutput) {
        KALDI_WARN << "TableWriter: ark,scp output needs a regular archive file, got " << archive_name_;
        return false;
      }
    }
    if (type_ == kBothWspecifier && !script_.Open(script_name_, false, false)) return false;
    if (type_ == kScriptWspecifier) {
      std::vector<std::pair<std::string, std::string> > entries;
      if (!ReadScriptFile(script_name_, true, &entries)) return false;
      for (size_t i = 0; i < entries.size(); ++i) script_map_[entries[i].first] = entries[i].second;
    }
    return true;
  }
  bool IsOpen() const { return type_ != kNoWspecifier; }
  void Write(const std::string &key, const T &value) {
    WriteWith(key, [&value](std::ostream &os, bool binary) { return Holder::Write(os, binary, value); });
  }
  /// The entry for `key` with its value written by `payload(stream, binary)` (false = failure): for a caller
  /// whose value is not held in a T, e.g. matrix rows in a download buffer.  It must write what Holder::Write
  /// would have written for that value.
  typedef std::function<bool(std::ostream &, bool)> PayloadWriter;
  void WriteWith(const std::string &key, const PayloadWriter &payload) {
    if (!IsToken(key)) KALDI_ERR << "TableWriter: using invalid key " << key;
    bool ok = true;
    if (type_ == kScriptWspecifier) {
      std::unordered_map<std::string, std::string>::iterator it = script_map_.find(key);
      if (it == script_map_.end()) KALDI_ERR << "TableWriter: key " << key << " not present in script file " << script_name_;
      Output out;
      ok = out.Open(it->second, opts_.binary, false) && payload(out.Stream(), opts_.binary) && out.Close();
    } else {
      std::ostream &os = archive_.Stream();
      os << key << ' ';
      if (type_ == kBothWspecifier) {
        os.flush();
        script_.Stream() << key << ' ' << archive_name_ << ':' << static_cast<long long>(os.tellp()) << '\n';
      }
      ok = payload(os, opts_.binary);
      if (opts_.flush) {
        os.flush();
        if (type_ == kBothWspecifier) script_.Stream().flush();
      }
    }
    if (!ok) KALDI_ERR << "TableWriter: write failure to " << wspecifier_;
  }
  void Flush() {
    if (archive_.IsOpen()) archive_.Stream().flush();
    if (script_.IsOpen()) script_.Stream().flush();
  }
  bool Close() {
    bool ok = true;
    if (archive_.IsOpen()) ok = archive_.Close() && ok;
    if (script_.IsOpen()) ok = script_.Close() && ok;
    type_ = kNoWspecifier;
    return ok;
  }

 private:
  std::string wspecifier_, archive_name_, script_name_;
  WspecifierType type_;
  WspecifierOptions opts_;
  Output archive_, script_;
  std::unordered_map<std::string, std::string> script_map_;
  KALDI_DISALLOW_COPY_AND_ASSIGN(TableWriter);
};

// ---- writer with a write-behind thread ------------------------------------------------------
// The caller fills one of a few reusable value buffers (Acquire), hands it over with its key
// (Submit) and goes on; a background thread serialises the values into the table in submission
// order and returns the buffers.  nnet-forward spends most of its time formatting and writing
// posteriors (nnetbin/nnet-forward.cc:151-160): this takes that off the thread that drives the
// GPU.  An error raised while writing surfaces at the next Acquire / Close.  Not in the reference.
template <class Holder>
class AsyncTableWriter {
 public:
  typedef typename Holder::T T;
  typedef std::function<void(const std::string &, const T &)> Check;
  /// `before_write` (optional) runs on the writer thread for every value before it is written, e.g. a
  /// validity check that throws
  explicit AsyncTableWriter(const std::string &wspecifier, size_t num_buffers = 3, Check before_write = Check())
      : writer_(wspecifier), buffers_(num_buffers), before_write_(before_write), stop_(false) {
    for (size_t i = 0; i < buffers_.size(); ++i) free_.push_back(&buffers_[i]);
    thread_ = std::thread(&AsyncTableWriter::Run, this);
  }
  ~AsyncTableWriter() {
    try {
      Close();
    } catch (...) {
    }
  }
  /// a buffer nobody else is using (waits for the writer thread if all are queued)
  T *Acquire() {
    std::unique_lock<std::mutex> l(mu_);
    cv_free_.wait(l, [this] { return !free_.empty() || error_; });
    Rethrow();
    T *b = free_.back();
    free_.pop_back();
    return b;
  }
  void Submit(const std::string &key, T *value) {
    std::lock_guard<std::mutex> l(mu_);
    queue_.push_back(std::make_pair(key, value));
    cv_work_.notify_one();
  }
  /// writes everything submitted so far, stops the thread and closes the table
  bool Close() {
    if (thread_.joinable()) {
      {
        std::lock_guard<std::mutex> l(mu_);
        stop_ = true;
      }
      cv_work_.notify_all();
      thread_.join();
    }
    {
      std::lock_guard<std::mutex> l(mu_);
      Rethrow();
    }
    return writer_.IsOpen() ? writer_.Close() : true;
  }

 private:
  void Rethrow() {
    if (error_) {
      std::exception_ptr e = error_;
      error_ = nullptr;
      std::rethrow_exception(e);
    }
  }
  void Run() {
    for (;;) {
      std::pair<std::string, T *> item;
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_work_.wait(l, [this] { return !queue_.empty() || stop_; });
        if (queue_.empty()) return;
        item = queue_.front();
        queue_.pop_front();
      }
      try {
        if (before_write_) before_write_(item.first, *item.second);
        writer_.Write(item.first, *item.second);
      } catch (...) {
        std::lock_guard<std::mutex> l(mu_);
        if (!error_) error_ = std::current_exception();
      }
      std::lock_guard<std::mutex> l(mu_);
      free_.push_back(item.second);
      cv_free_.notify_one();
    }
  }
  TableWriter<Holder> writer_;
  std::vector<T> buffers_;
  Check before_write_;
  std::vector<T *> free_;
  std::deque<std::pair<std::string, T *> > queue_;
  std::thread thread_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_free_;
  std::exception_ptr error_;
  bool stop_;
};

}  // namespace kaldi
#endif
