#include "cudamatrix/cu-device.h"

#include <stdlib.h>

#include <algorithm>
#include <iostream>
#include <sstream>
#include <vector>

namespace kaldi {

void CuDevice::SelectGpuId(int32 gpu_id) {
  if (initialised_ && (gpu_id < 0 || gpu_id == active_gpu_id_)) return;
  requested_gpu_id_ = gpu_id;
  int n = 0;
  CU_SAFE_CALL(hipK_device_count(&n));
  if (n <= 0) KALDI_ERR << "No GPU visible: this build has no CPU fallback (an MI355X / gfx950 device is required)";
  CU_SAFE_CALL(hipK_init(gpu_id));
  int dev = -1;
  CU_SAFE_CALL(hipK_active_device(&dev));
  active_gpu_id_ = dev;
  CU_SAFE_CALL(hipK_default_stream(&main_stream_));
  stream_ = main_stream_;
  side_stream_ = NULL;
  aux_stream_ = NULL;
  copy_stream_ = NULL;
  initialised_ = true;
  if (verbose_) {
    char arch[64];
    int cus = 0, wave = 0;
    CU_SAFE_CALL(hipK_device_props(&cus, &wave, arch, sizeof(arch)));
    KALDI_LOG << "Selected device: " << active_gpu_id_ << " (" << (gpu_id < 0 ? "automatically" : "manually") << ") "
              << arch << ", " << cus << " CUs, " << GetFreeMemory();
  }
}

std::string CuDevice::GetFreeMemory() {
  size_t free_b = 0, total_b = 0;
  CU_SAFE_CALL(hipK_mem_info(&free_b, &total_b));
  std::ostringstream os;
  os << "free: " << free_b / (1024 * 1024) << "M, total: " << total_b / (1024 * 1024) << "M, ratio: "
     << (total_b ? static_cast<double>(free_b) / total_b : 0.0);
  return os.str();
}

void CuDevice::PrintProfile() {
  if (!verbose_ || profile_map_.empty()) return;
  std::vector<std::pair<double, std::string> > rows;
  for (std::map<std::string, double>::const_iterator it = profile_map_.begin(); it != profile_map_.end(); ++it)
    rows.push_back(std::make_pair(it->second, it->first));
  std::sort(rows.begin(), rows.end());
  std::ostringstream os;
  os << "-----\n[cudevice profile]\n";
  for (size_t i = 0; i < rows.size(); ++i) os << rows[i].second << "\t" << rows[i].first << "s\n";
  os << "-----";
  KALDI_LOG << os.str();
}

}  // namespace kaldi
