// ranges.hpp -- roctx ranges around the host-side phases of the step drivers ("roctx ranges +
// rocprofv3 counters"), so that `rocprofv3 --kernel-trace --marker-trace` shows which step / RK stage / exchange a
// kernel belongs to. Off unless T8GPU_ROCTX=1: the library is then dlopen'ed (libroctx64 of the ROCm install or the one
// torch bundles -- whichever the process already has), so there is no link-time dependency and no cost when off.
#ifndef T8GPU_HIP_RANGES_HPP
#define T8GPU_HIP_RANGES_HPP

#include <dlfcn.h>

#include <cstdlib>

namespace t8gpu_hip {

struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)()             = nullptr;
  bool on                  = false;
};

inline RoctxApi load_roctx() {
  RoctxApi    api;
  const char* env = std::getenv("T8GPU_ROCTX");
  if (!env || env[0] != '1') return api;
  for (const char* name : {"libroctx64.so", "libroctx64.so.4", "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so"}) {
    void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    api.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    api.pop  = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (api.push && api.pop) {
      api.on = true;
      break;
    }
  }
  return api;
}

inline const RoctxApi& roctx() {
  static const RoctxApi api = load_roctx();
  return api;
}

/// RAII host range; a no-op unless T8GPU_ROCTX=1 and a roctx library could be loaded.
struct Range {
  explicit Range(const char* name) : live(roctx().on) {
    if (live) roctx().push(name);
  }
  ~Range() {
    if (live) roctx().pop();
  }
  Range(const Range&)            = delete;
  Range& operator=(const Range&) = delete;
  bool live;
};

}  // namespace t8gpu_hip

#endif  // T8GPU_HIP_RANGES_HPP
