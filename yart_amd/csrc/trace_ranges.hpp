// trace_ranges.hpp — roctx ranges around the stages of a render (host only).
//
// The reference times its render, waves and tiles with std::chrono and reports them through its callbacks
// (cpu/tile-renderer.hpp:111-113, 146-147, 169, 212-215); YartStats carries the same figures per stage from HIP events.
// For a profiler's timeline the stages are also marked as roctx ranges: `rocprofv3 --marker-trace --kernel-trace` then
// shows sampler tables / generate / bounce k { extend, shade, shadow, roulette, compact } / blend around the kernels
// they launch. The roctx library is loaded with dlopen on first use (as RCCL is, multi_device.inc): without it — or with
// YART_ROCTX=0 — a range costs one predictable branch. Launches are asynchronous, so a range normally covers the
// ENQUEUE of its stage; with YART_ROCTX_SYNC=1 a range ends with a synchronisation of the render stream and covers the
// stage's execution (a measurement mode: it serialises host and device, never set by default).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace yart_hip {

struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  void (*mark)(const char*) = nullptr;
  bool sync = false;
};
inline const RoctxApi& roctx() {
  static RoctxApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* on = std::getenv("YART_ROCTX");
    if (on && std::strcmp(on, "0") == 0) return;
    void* h = nullptr;
    const char* named = std::getenv("YART_ROCTX_LIB");
    if (named && *named) h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    // (under rocprofv3 the SDK's roctx is already in the process; RTLD_NOLOAD finds it without loading a second copy)
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    }
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so.4"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return;
    auto push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    auto mark = reinterpret_cast<void (*)(const char*)>(dlsym(h, "roctxMarkA"));
    if (!push || !pop) return;
    api.push = push; api.pop = pop; api.mark = mark;
    const char* s = std::getenv("YART_ROCTX_SYNC");
    api.sync = s && std::strcmp(s, "0") != 0;
  });
  return api;
}

// Scoped range. `stream`: synchronised at the end of the range in YART_ROCTX_SYNC mode (errors are left to the caller's next check).
class TraceRange {
 public:
  explicit TraceRange(const char* name, hipStream_t stream = nullptr) : stream_(stream) {
    const RoctxApi& r = roctx();
    if (r.push) { r.push(name); open_ = true; }
  }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
  void end() {
    if (!open_) return;
    const RoctxApi& r = roctx();
    if (r.sync) (void)hipStreamSynchronize(stream_);
    r.pop();
    open_ = false;
  }
  ~TraceRange() { end(); }

 private:
  hipStream_t stream_;
  bool open_ = false;
};

}  // namespace yart_hip
