#include "util/kaldi-io.h"

#include <ctype.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include <ext/stdio_filebuf.h>

namespace kaldi {

static bool AllDigits(const std::string &s) {
  if (s.empty()) return false;
  for (char c : s)
    if (!isdigit(static_cast<unsigned char>(c))) return false;
  return true;
}

OutputType ClassifyWxfilename(const std::string &name) {
  if (name.empty() || name == "-") return kStandardOutput;
  if (name[0] == '|') return kPipeOutput;
  if (isspace(static_cast<unsigned char>(name[0])) || isspace(static_cast<unsigned char>(name[name.size() - 1])))
    return kNoOutput;
  if (name[name.size() - 1] == '|') return kNoOutput;  // an input pipe is not writable
  const size_t colon = name.rfind(':');
  if (colon != std::string::npos && AllDigits(name.substr(colon + 1))) return kNoOutput;  // offsets are read-only
  return kFileOutput;
}

InputType ClassifyRxfilename(const std::string &name) {
  if (name.empty() || name == "-") return kStandardInput;
  if (name[0] == '|') return kNoInput;  // an output pipe is not readable
  if (isspace(static_cast<unsigned char>(name[0])) || isspace(static_cast<unsigned char>(name[name.size() - 1])))
    return kNoInput;
  if (name[name.size() - 1] == '|') return kPipeInput;
  const size_t colon = name.rfind(':');
  if (colon != std::string::npos && AllDigits(name.substr(colon + 1))) return kOffsetFileInput;
  return kFileInput;
}

std::string PrintableRxfilename(const std::string &n) { return (n.empty() || n == "-") ? "standard input" : "\"" + n + "\""; }
std::string PrintableWxfilename(const std::string &n) { return (n.empty() || n == "-") ? "standard output" : "\"" + n + "\""; }

// ---- Output ----------------------------------------------------------------------------------
Output::Output(const std::string &wxfilename, bool binary, bool write_header) {
  if (!Open(wxfilename, binary, write_header))
    KALDI_ERR << "Error opening output stream " << PrintableWxfilename(wxfilename);
}

Output::~Output() {
  if (IsOpen() && !Close()) KALDI_WARN << "Error closing output file " << PrintableWxfilename(name_);
}

bool Output::Open(const std::string &wxfilename, bool binary, bool write_header) {
  if (IsOpen() && !Close()) KALDI_ERR << "Output::Open(), failed to close output stream: " << PrintableWxfilename(name_);
  name_ = wxfilename;
  switch (ClassifyWxfilename(wxfilename)) {
    case kStandardOutput:
      os_ = &std::cout;
      break;
    case kFileOutput:
      file_.reset(new std::ofstream(wxfilename.c_str(), std::ios_base::out | std::ios_base::binary));
      if (!file_->is_open()) {
        file_.reset();
        return false;
      }
      os_ = file_.get();
      break;
    case kPipeOutput: {
      const std::string cmd = wxfilename.substr(1);
      pipe_ = popen(cmd.c_str(), "w");
      if (!pipe_) {
        KALDI_WARN << "Failed opening pipe for writing, command is: " << cmd << ", errno is " << strerror(errno);
        return false;
      }
      pipe_buf_.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_, std::ios_base::out | std::ios_base::binary));
      pipe_stream_.reset(new std::ostream(pipe_buf_.get()));
      os_ = pipe_stream_.get();
      break;
    }
    default:
      KALDI_WARN << "Invalid output filename format " << PrintableWxfilename(wxfilename);
      return false;
  }
  if (write_header) {
    InitKaldiOutputStream(*os_, binary);
    if (!os_->good()) {
      Close();
      return false;
    }
  }
  return true;
}

std::ostream &Output::Stream() {
  if (!os_) KALDI_ERR << "Output::Stream() called but not open.";
  return *os_;
}

bool Output::Close() {
  if (!os_) return false;
  os_->flush();
  bool ok = !os_->fail();
  if (file_) {
    file_->close();
    ok = ok && !file_->fail();
    file_.reset();
  }
  if (pipe_) {
    pipe_stream_.reset();
    pipe_buf_.reset();
    const int status = pclose(pipe_);
    pipe_ = nullptr;
    if (status != 0) {
      KALDI_WARN << "Pipe " << name_ << " had nonzero return status " << status;
      ok = false;
    }
  }
  os_ = nullptr;
  return ok;
}

// ---- Input -----------------------------------------------------------------------------------
Input::Input(const std::string &rxfilename, bool *contents_binary) {
  if (!Open(rxfilename, contents_binary)) KALDI_ERR << "Error opening input stream " << PrintableRxfilename(rxfilename);
}

Input::~Input() { Close(); }

bool Input::Open(const std::string &rxfilename, bool *contents_binary) {
  return OpenInternal(rxfilename, true, contents_binary);
}
bool Input::OpenTextMode(const std::string &rxfilename) { return OpenInternal(rxfilename, false, NULL); }

bool Input::OpenInternal(const std::string &rxfilename, bool look_for_header, bool *contents_binary) {
  Close();
  name_ = rxfilename;
  switch (ClassifyRxfilename(rxfilename)) {
    case kStandardInput:
      is_ = &std::cin;
      break;
    case kFileInput:
    case kOffsetFileInput: {
      std::string path = rxfilename;
      long long offset = 0;
      if (ClassifyRxfilename(rxfilename) == kOffsetFileInput) {
        const size_t colon = rxfilename.rfind(':');
        path = rxfilename.substr(0, colon);
        offset = atoll(rxfilename.c_str() + colon + 1);
      }
      file_.reset(new std::ifstream(path.c_str(), std::ios_base::in | std::ios_base::binary));
      if (!file_->is_open()) {
        file_.reset();
        return false;
      }
      if (offset > 0) {
        file_->seekg(offset, std::ios_base::beg);
        if (file_->fail()) {
          file_.reset();
          return false;
        }
      }
      is_ = file_.get();
      break;
    }
    case kPipeInput: {
      const std::string cmd = rxfilename.substr(0, rxfilename.size() - 1);
      pipe_ = popen(cmd.c_str(), "r");
      if (!pipe_) {
        KALDI_WARN << "Failed opening pipe for reading, command is: " << cmd << ", errno is " << strerror(errno);
        return false;
      }
      pipe_buf_.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_, std::ios_base::in | std::ios_base::binary));
      pipe_stream_.reset(new std::istream(pipe_buf_.get()));
      is_ = pipe_stream_.get();
      break;
    }
    default:
      KALDI_WARN << "Invalid input filename format " << PrintableRxfilename(rxfilename);
      return false;
  }
  if (look_for_header) {
    bool binary = false;
    if (!InitKaldiInputStream(*is_, &binary)) {
      Close();
      return false;
    }
    if (contents_binary) *contents_binary = binary;
  }
  return true;
}

std::istream &Input::Stream() {
  if (!is_) KALDI_ERR << "Input::Stream(), not open.";
  return *is_;
}

void Input::Close() {
  if (file_) file_.reset();
  if (pipe_) {
    pipe_stream_.reset();
    pipe_buf_.reset();
    pclose(pipe_);
    pipe_ = nullptr;
  }
  is_ = nullptr;
}

}  // namespace kaldi
