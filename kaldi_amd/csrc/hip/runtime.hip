// Device selection, memory and copies: what the reference took from the CUDA runtime and
// cu-device.cc (src/cudamatrix/cu-device.cc:56-210, cu-matrix-inl.h:96-310).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace kh {

Context &ctx() {
  static Context c;
  return c;
}

static int workspace_of(hipStream_t st, Workspace **out) {
  Context &c = ctx();
  Workspace *w = nullptr;
  for (int i = 0; i < c.num_workspaces; ++i)
    if (c.workspaces[i].stream == st) w = &c.workspaces[i];
  if (!w) {
    if (c.num_workspaces == Context::kMaxWorkspaces) {
      snprintf(c.last_msg, sizeof(c.last_msg), "more than %d streams use library scratch memory", Context::kMaxWorkspaces);
      return KH_ERR_TOO_LARGE;
    }
    w = &c.workspaces[c.num_workspaces++];
    w->stream = st;
  }
  *out = w;
  return KH_OK;
}

int stream_ticket(hipStream_t st, unsigned **ticket) {
  Workspace *w;
  int rc = workspace_of(st, &w);
  if (rc != KH_OK) return rc;
  if (!w->ticket) {
    KH_HIP(hipMalloc((void **)&w->ticket, 256));
    // Zeroed ON the stream whose kernels use it.  This used to be a hipMemset, the library's only operation on
    // the null stream, and it had two effects.  A non-blocking stream has no ordering with the null stream and
    // the fill is not guaranteed to have finished when hipMemset returns: the first loss kernels of a process
    // could see the allocation's old contents, no workgroup was "last", and the sums of those bunches were
    // dropped (5 of 40 short tool runs).  And with priority streams on 4 hardware queues, the copy stream created
    // before the update streams, a training tool never came back from synchronising its main stream at the second
    // cache fill (3 of 3 runs with the null-stream fill, 0 of 5 without, same settings otherwise).  The library now
    // issues nothing on the null stream.
    KH_HIP(hipMemsetAsync(w->ticket, 0, 256, st));
  }
  *ticket = w->ticket;
  return KH_OK;
}

int ensure_workspace(hipStream_t st, size_t bytes, void **ptr) {
  Workspace *w;
  int rc = workspace_of(st, &w);
  if (rc != KH_OK) return rc;
  if (bytes > w->bytes) {
    // Allocation churn is what this avoids: round up generously, reallocate rarely.
    size_t want = bytes + bytes / 4;
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    KH_HIP(hipDeviceSynchronize());
    if (w->ptr) KH_HIP(hipFree(w->ptr));
    w->ptr = nullptr;
    w->bytes = 0;
    KH_HIP(hipMalloc(&w->ptr, want));
    w->bytes = want;
  }
  *ptr = w->ptr;
  return KH_OK;
}

static int free_workspaces() {
  Context &c = ctx();
  for (int i = 0; i < c.num_workspaces; ++i) {
    if (c.workspaces[i].ptr) KH_HIP(hipFree(c.workspaces[i].ptr));
    if (c.workspaces[i].ticket) KH_HIP(hipFree(c.workspaces[i].ticket));
    c.workspaces[i] = Workspace();
  }
  c.num_workspaces = 0;
  return KH_OK;
}

}  // namespace kh

using kh::ctx;

// KALDI_HIP_SLOW_CALLS=<ms>: report (stderr) every blocking runtime call of this file that takes longer -- allocation,
// free, synchronisation, blocking copies.  A diagnostic for stalls that are not the kernels' (a shared GPU, a
// driver-side wait); off by default.
namespace {
struct SlowCall {
  const char *what;
  timespec t0;
  static double limit_ms() {
    static const double v = getenv("KALDI_HIP_SLOW_CALLS") ? atof(getenv("KALDI_HIP_SLOW_CALLS")) : 0.0;
    return v;
  }
  explicit SlowCall(const char *w) : what(w) {
    if (limit_ms() > 0.0) clock_gettime(CLOCK_MONOTONIC, &t0);
  }
  ~SlowCall() {
    if (limit_ms() <= 0.0) return;
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double ms = 1e3 * (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_nsec - t0.tv_nsec);
    if (ms > limit_ms()) fprintf(stderr, "[kaldi_hip] %s took %.1f ms\n", what, ms);
  }
};
// KALDI_HIP_MEM_REPORT=1: at process exit, one line (stderr) with the device memory the callers of hipK_malloc /
// hipK_malloc_pitch held at most, what was still held at exit, and the number of allocations made.  A diagnostic for
// soak runs (does the footprint grow with the number of utterances?); off by default, and then nothing is recorded.
struct MemLedger {
  std::mutex mu;
  std::unordered_map<void *, size_t> live;
  size_t now = 0, peak = 0, calls = 0;
  bool on = getenv("KALDI_HIP_MEM_REPORT") != nullptr;
  void add(void *p, size_t bytes) {
    if (!on) return;
    std::lock_guard<std::mutex> l(mu);
    live[p] = bytes;
    now += bytes;
    peak = std::max(peak, now);
    ++calls;
  }
  void drop(void *p) {
    if (!on) return;
    std::lock_guard<std::mutex> l(mu);
    std::unordered_map<void *, size_t>::iterator it = live.find(p);
    if (it == live.end()) return;
    now -= it->second;
    live.erase(it);
  }
  ~MemLedger() {
    if (on)
      fprintf(stderr, "[kaldi_hip] device memory held through hipK_malloc*: peak %.1f MiB, at exit %.1f MiB in %zu blocks, %zu allocations\n",
              peak / 1048576.0, now / 1048576.0, live.size(), calls);
  }
};
MemLedger &ledger() {
  static MemLedger l;
  return l;
}
}  // namespace


KH_EXPORT int hipK_device_count(int *count) {
  KH_REQUIRE(count != nullptr);
  KH_HIP(hipGetDeviceCount(count));
  return KH_OK;
}

KH_EXPORT int hipK_init(int device_id) {
  kh::Context &c = ctx();
  if (c.initialised && (device_id < 0 || device_id == c.device)) return KH_OK;
  if (c.initialised) {
    int st = hipK_shutdown();
    if (st != KH_OK) return st;
  }
  int n = 0;
  KH_HIP(hipGetDeviceCount(&n));
  if (n <= 0) return (int)hipErrorNoDevice;
  int chosen = device_id;
  if (device_id < 0) {
    // the GPU with the most free memory, like CuDevice's automatic selection
    size_t best_free = 0;
    chosen = 0;
    for (int i = 0; i < n; ++i) {
      size_t f = 0, t = 0;
      if (hipSetDevice(i) != hipSuccess) continue;
      if (hipMemGetInfo(&f, &t) != hipSuccess) continue;
      if (f > best_free) {
        best_free = f;
        chosen = i;
      }
    }
  } else if (device_id >= n) {
    snprintf(c.last_msg, sizeof(c.last_msg), "hipK_init: device %d requested, %d visible", device_id, n);
    return KH_ERR_INVALID_ARG;
  }
  KH_HIP(hipSetDevice(chosen));
  hipDeviceProp_t prop;
  KH_HIP(hipGetDeviceProperties(&prop, chosen));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(c.last_msg, sizeof(c.last_msg), "hipK_init: device %d is %s; this library is built for gfx950 only",
             chosen, prop.gcnArchName);
    return KH_ERR_INVALID_ARG;
  }
  {
    // the library's default stream carries the chain of dependent kernels: give it the highest
    // priority, so that its (short) kernels are not queued behind work of the other streams
    int lo = 0, hi = 0;
    KH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char *e = getenv("KALDI_HIP_STREAM_PRIO");
    if (e && atoi(e) == 0) KH_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    else KH_HIP(hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, hi));
  }
  {
    int st = kh::static_scratch_setup(chosen);
    if (st != KH_OK) return st;
  }
  c.device = chosen;
  c.num_cu = prop.multiProcessorCount;
  c.initialised = true;
  return KH_OK;
}

namespace {
int staging_free();
}

KH_EXPORT int hipK_shutdown(void) {
  kh::Context &c = ctx();
  if (!c.initialised) return KH_OK;
  KH_HIP(hipDeviceSynchronize());
  int st = kh::free_workspaces();
  if (st == KH_OK) st = staging_free();
  if (st != KH_OK) return st;
  if (c.stream) KH_HIP(hipStreamDestroy(c.stream));
  c.stream = nullptr;
  c.initialised = false;
  c.device = -1;
  return KH_OK;
}

KH_EXPORT int hipK_active_device(int *device_id) {
  KH_REQUIRE(device_id != nullptr);
  *device_id = ctx().initialised ? ctx().device : -1;
  return KH_OK;
}

KH_EXPORT int hipK_mem_info(size_t *free_bytes, size_t *total_bytes) {
  KH_CHECK_INIT();
  KH_REQUIRE(free_bytes && total_bytes);
  KH_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return KH_OK;
}

KH_EXPORT int hipK_default_stream(khStream_t *stream) {
  KH_CHECK_INIT();
  KH_REQUIRE(stream != nullptr);
  *stream = (khStream_t)ctx().stream;
  return KH_OK;
}

KH_EXPORT int hipK_stream_create(khStream_t *stream) {
  KH_CHECK_INIT();
  KH_REQUIRE(stream != nullptr);
  hipStream_t s;
  int lo = 0, hi = 0;
  KH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  const char *e = getenv("KALDI_HIP_STREAM_PRIO");
  if (e && atoi(e) == 0) KH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  else KH_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));  // lowest: see hipK_init
  *stream = (khStream_t)s;
  return KH_OK;
}

KH_EXPORT int hipK_stream_destroy(khStream_t stream) {
  KH_REQUIRE(stream != nullptr);
  KH_HIP(hipStreamSynchronize((hipStream_t)stream));
  kh::Context &c = ctx();
  for (int i = 0; i < c.num_workspaces; ++i)
    if (c.workspaces[i].stream == (hipStream_t)stream) {
      if (c.workspaces[i].ptr) KH_HIP(hipFree(c.workspaces[i].ptr));
      if (c.workspaces[i].ticket) KH_HIP(hipFree(c.workspaces[i].ticket));
      c.workspaces[i] = c.workspaces[--c.num_workspaces];
      c.workspaces[c.num_workspaces] = kh::Workspace();
      break;
    }
  KH_HIP(hipStreamDestroy((hipStream_t)stream));
  return KH_OK;
}

KH_EXPORT int hipK_stream_synchronize(khStream_t stream) {
  KH_CHECK_INIT();
  SlowCall sc("hipStreamSynchronize");
  KH_HIP(hipStreamSynchronize(kh::pick(stream)));
  return KH_OK;
}

KH_EXPORT int hipK_stream_wait_event(khStream_t stream, khEvent_t ev) {
  KH_CHECK_INIT();
  KH_REQUIRE(ev != nullptr);
  KH_HIP(hipStreamWaitEvent(kh::pick(stream), (hipEvent_t)ev, 0));
  return KH_OK;
}

KH_EXPORT int hipK_device_synchronize(void) {
  KH_CHECK_INIT();
  SlowCall sc("hipDeviceSynchronize");
  KH_HIP(hipDeviceSynchronize());
  return KH_OK;
}

KH_EXPORT const char *hipK_error_string(int status) {
  switch (status) {
    case KH_OK: return "success";
    case KH_ERR_INVALID_ARG: return ctx().last_msg[0] ? ctx().last_msg : "invalid argument";
    case KH_ERR_NOT_INITIALISED: return "library not initialised";
    case KH_ERR_COMM: return ctx().last_msg[0] ? ctx().last_msg : "RCCL error";
    case KH_ERR_TOO_LARGE: return ctx().last_msg[0] ? ctx().last_msg : "size beyond the supported maximum";
    default: return hipGetErrorString((hipError_t)status);
  }
}

KH_EXPORT int hipK_device_props(int *num_cu, int *wave_size, char *arch_name, size_t arch_name_len) {
  KH_CHECK_INIT();
  hipDeviceProp_t prop;
  KH_HIP(hipGetDeviceProperties(&prop, ctx().device));
  if (num_cu) *num_cu = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch_name && arch_name_len) {
    strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return KH_OK;
}

KH_EXPORT int hipK_malloc_pitch(void **ptr, int32_t *stride_elems, int32_t rows, int32_t cols, size_t elem_size) {
  KH_CHECK_INIT();
  KH_REQUIRE(ptr && stride_elems && rows >= 0 && cols >= 0 && (elem_size == 4 || elem_size == 8));
  // Rows start on 256-byte boundaries: whole rows are then made of full 128-byte lines and every
  // row base is 16-byte aligned for dwordx4 accesses.
  size_t pitch = ((size_t)cols * elem_size + 255) & ~(size_t)255;
  *stride_elems = (int32_t)(pitch / elem_size);
  *ptr = nullptr;
  if (rows == 0 || cols == 0) return KH_OK;
  SlowCall sc("hipMalloc");
  KH_HIP(hipMalloc(ptr, pitch * (size_t)rows));
  ledger().add(*ptr, pitch * (size_t)rows);
  return KH_OK;
}

KH_EXPORT int hipK_malloc(void **ptr, size_t bytes) {
  KH_CHECK_INIT();
  KH_REQUIRE(ptr != nullptr);
  *ptr = nullptr;
  if (bytes == 0) return KH_OK;
  SlowCall sc("hipMalloc");
  KH_HIP(hipMalloc(ptr, bytes));
  ledger().add(*ptr, bytes);
  return KH_OK;
}

KH_EXPORT int hipK_free(void *ptr) {
  if (!ptr) return KH_OK;
  SlowCall sc("hipFree");
  KH_HIP(hipFree(ptr));
  ledger().drop(ptr);
  return KH_OK;
}

KH_EXPORT int hipK_malloc_host(void **ptr, size_t bytes) {
  KH_CHECK_INIT();
  KH_REQUIRE(ptr != nullptr);
  *ptr = nullptr;
  if (bytes == 0) return KH_OK;
  SlowCall sc("hipHostMalloc");
  KH_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
  return KH_OK;
}

KH_EXPORT int hipK_free_host(void *ptr) {
  if (!ptr) return KH_OK;
  KH_HIP(hipHostFree(ptr));
  return KH_OK;
}

KH_EXPORT int hipK_memset_async(khStream_t s, void *ptr, int value, size_t bytes) {
  KH_CHECK_INIT();
  if (bytes == 0) return KH_OK;
  KH_REQUIRE(ptr != nullptr);
  KH_HIP(hipMemsetAsync(ptr, value, bytes, kh::pick(s)));
  return KH_OK;
}

static hipMemcpyKind to_kind(int kind) {
  return kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
}

// Host <-> device copies of pitched matrices go through two pinned staging buffers: the host side
// is ordinary pageable memory (a freshly read utterance, a std::vector), and a pageable 2-D copy
// ran at ~0.5 GB/s here.  The CPU packs / unpacks one chunk while the DMA engine moves the other.
namespace {
constexpr size_t kStageBytes = (size_t)16 << 20;
struct Staging {
  char *buf[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  std::mutex mu;  // one copy at a time owns the two buffers (host threads of the tools' pipelines may call in)
};
Staging &staging() {
  static Staging s;
  return s;
}
int staging_init() {
  Staging &g = staging();
  for (int i = 0; i < 2; ++i) {
    if (!g.buf[i]) KH_HIP(hipHostMalloc((void **)&g.buf[i], kStageBytes, hipHostMallocDefault));
    if (!g.done[i]) KH_HIP(hipEventCreateWithFlags(&g.done[i], hipEventDisableTiming));
  }
  return KH_OK;
}
int staging_free() {
  Staging &g = staging();
  for (int i = 0; i < 2; ++i) {
    if (g.buf[i]) KH_HIP(hipHostFree(g.buf[i]));
    if (g.done[i]) KH_HIP(hipEventDestroy(g.done[i]));
    g.buf[i] = nullptr;
    g.done[i] = nullptr;
  }
  return KH_OK;
}
// rows of `width` bytes, packed tightly in the staging buffers, chunk by chunk
int staged_copy_2d(hipStream_t st, char *dst, size_t dpitch, const char *src, size_t spitch, size_t width, size_t height,
                   bool to_device) {
  std::lock_guard<std::mutex> own(staging().mu);
  int rc = staging_init();
  if (rc != KH_OK) return rc;
  Staging &g = staging();
  const size_t rows_per = kStageBytes / width;
  if (rows_per == 0) {  // a single row larger than a staging buffer: plain copy
    KH_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
    KH_HIP(hipStreamSynchronize(st));
    return KH_OK;
  }
  if (to_device) {
    int b = 0;
    for (size_t r0 = 0; r0 < height; r0 += rows_per, b ^= 1) {
      const size_t n = std::min(rows_per, height - r0);
      KH_HIP(hipEventSynchronize(g.done[b]));  // the DMA that last read this buffer
      if (spitch == width) memcpy(g.buf[b], src + r0 * spitch, n * width);
      else
        for (size_t r = 0; r < n; ++r) memcpy(g.buf[b] + r * width, src + (r0 + r) * spitch, width);
      KH_HIP(hipMemcpy2DAsync(dst + r0 * dpitch, dpitch, g.buf[b], width, width, n, hipMemcpyHostToDevice, st));
      KH_HIP(hipEventRecord(g.done[b], st));
    }
    KH_HIP(hipStreamSynchronize(st));
  } else {
    // chunk i+1 is in flight while chunk i is unpacked
    size_t r_prev = 0, n_prev = 0;
    int b = 0;
    for (size_t r0 = 0; r0 < height || n_prev; r0 += rows_per, b ^= 1) {
      size_t n = 0;
      if (r0 < height) {
        n = std::min(rows_per, height - r0);
        KH_HIP(hipMemcpy2DAsync(g.buf[b], width, src + r0 * spitch, spitch, width, n, hipMemcpyDeviceToHost, st));
        KH_HIP(hipEventRecord(g.done[b], st));
      }
      if (n_prev) {
        KH_HIP(hipEventSynchronize(g.done[b ^ 1]));
        if (dpitch == width) memcpy(dst + r_prev * dpitch, g.buf[b ^ 1], n_prev * width);
        else
          for (size_t r = 0; r < n_prev; ++r) memcpy(dst + (r_prev + r) * dpitch, g.buf[b ^ 1] + r * width, width);
      }
      r_prev = r0;
      n_prev = n;
    }
  }
  return KH_OK;
}
}  // namespace

KH_EXPORT int hipK_memcpy_2d(khStream_t s, void *dst, size_t dpitch, const void *src, size_t spitch,
                             size_t width_bytes, size_t height, int kind) {
  KH_CHECK_INIT();
  if (width_bytes == 0 || height == 0) return KH_OK;
  KH_REQUIRE(dst && src && kind >= 0 && kind <= 2 && dpitch >= width_bytes && spitch >= width_bytes);
  hipStream_t st = kh::pick(s);
  if (kind != 2) {
    // Between pageable host memory and the device the runtime copies a pitched matrix row by row: 9.3 us per row,
    // whatever its length (a 366 x 40 utterance: 3.4 ms; measured, tools/h2d_probe.py).  Every copy of more than one
    // row therefore goes through the pinned staging buffers (24-40 us for the same utterance); one row, or rows that
    // lie back to back on both sides, are one plain copy.
    const bool one_block = height == 1 || (dpitch == width_bytes && spitch == width_bytes);
    if (!one_block || width_bytes * height >= 65536)
      return staged_copy_2d(st, (char *)dst, dpitch, (const char *)src, spitch, width_bytes, height, kind == 0);
    KH_HIP(hipMemcpyAsync(dst, src, width_bytes * height, to_kind(kind), st));
    KH_HIP(hipStreamSynchronize(st));
    return KH_OK;
  }
  KH_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, to_kind(kind), st));
  return KH_OK;
}

// The copy is only queued: `s` runs it in order with its other work and the host goes on.  The host side must be
// pinned memory (hipK_malloc_host); it may be read (D2H) or reused (H2D) once an event recorded on `s` after
// this call has completed.
KH_EXPORT int hipK_memcpy_2d_async(khStream_t s, void *dst, size_t dpitch, const void *src, size_t spitch,
                                   size_t width_bytes, size_t height, int kind) {
  KH_CHECK_INIT();
  if (width_bytes == 0 || height == 0) return KH_OK;
  KH_REQUIRE(dst && src && kind >= 0 && kind <= 2 && dpitch >= width_bytes && spitch >= width_bytes);
  hipStream_t st = kh::pick(s);
  if (dpitch == width_bytes && spitch == width_bytes)
    KH_HIP(hipMemcpyAsync(dst, src, width_bytes * height, to_kind(kind), st));  // one contiguous block
  else
    KH_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, to_kind(kind), st));
  return KH_OK;
}

KH_EXPORT int hipK_memcpy(khStream_t s, void *dst, const void *src, size_t bytes, int kind) {
  KH_CHECK_INIT();
  if (bytes == 0) return KH_OK;
  KH_REQUIRE(dst && src && kind >= 0 && kind <= 2);
  hipStream_t st = kh::pick(s);
  SlowCall sc("hipK_memcpy");
  KH_HIP(hipMemcpyAsync(dst, src, bytes, to_kind(kind), st));
  if (kind != 2) KH_HIP(hipStreamSynchronize(st));
  return KH_OK;
}

KH_EXPORT int hipK_event_create(khEvent_t *ev) {
  KH_CHECK_INIT();
  KH_REQUIRE(ev != nullptr);
  hipEvent_t e;
  KH_HIP(hipEventCreate(&e));
  *ev = (khEvent_t)e;
  return KH_OK;
}
KH_EXPORT int hipK_event_create_ordering(khEvent_t *ev) {
  KH_CHECK_INIT();
  KH_REQUIRE(ev != nullptr);
  hipEvent_t e;
  KH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *ev = (khEvent_t)e;
  return KH_OK;
}
KH_EXPORT int hipK_event_destroy(khEvent_t ev) {
  if (!ev) return KH_OK;
  KH_HIP(hipEventDestroy((hipEvent_t)ev));
  return KH_OK;
}
KH_EXPORT int hipK_event_record(khEvent_t ev, khStream_t s) {
  KH_CHECK_INIT();
  KH_REQUIRE(ev != nullptr);
  KH_HIP(hipEventRecord((hipEvent_t)ev, kh::pick(s)));
  return KH_OK;
}
KH_EXPORT int hipK_event_synchronize(khEvent_t ev) {
  KH_REQUIRE(ev != nullptr);
  KH_HIP(hipEventSynchronize((hipEvent_t)ev));
  return KH_OK;
}
KH_EXPORT int hipK_event_elapsed_ms(float *ms, khEvent_t start, khEvent_t stop) {
  KH_REQUIRE(ms && start && stop);
  KH_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return KH_OK;
}

KH_EXPORT int hipK_reserve_workspace(size_t bytes) {
  KH_CHECK_INIT();
  void *p;
  return kh::ensure_workspace(kh::ctx().stream, bytes, &p);
}
