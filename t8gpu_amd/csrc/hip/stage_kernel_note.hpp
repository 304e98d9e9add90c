// stage_kernel_note.hpp -- which kernel did the last fused-stage call launch for most of its work? bench.py asks
// (t8gpu_hip_last_stage_kernel) so that PMC figures taken from a committed profile are reported only for the kernel that
// was profiled. Host-side bookkeeping only, kept per host thread (kernels_compat.hip).
#ifndef T8GPU_HIP_STAGE_KERNEL_NOTE_HPP
#define T8GPU_HIP_STAGE_KERNEL_NOTE_HPP

#include <cctype>
#include <cstdio>
#include <cstring>

namespace t8gpu_hip {

struct StageKernelNote {
  char      name[192];
  long long weight;
};
StageKernelNote& stage_kernel_note();   // (kernels_compat.hip)

inline void stage_kernel_note_reset() { stage_kernel_note().weight = -1; }

// pattern: the kernel as rocprofv3 prints it, with the identifiers T, K, S for float type, flux kind and stage, e.g.
// "k_plain_stage<T, K, S>"; weight: the work units (tiles, blocks) of this launch -- the heaviest launch of a call stays.
inline void note_stage_kernel(long long weight, const char* pattern, int tsize, int kind, int stage) {
  StageKernelNote& n = stage_kernel_note();
  if (weight <= n.weight) return;
  n.weight   = weight;
  size_t o   = 0;
  auto   put = [&](const char* s) {
    for (; *s && o + 1 < sizeof(n.name); s++) n.name[o++] = *s;
  };
  for (const char* p = pattern; *p;) {
    const bool ident_before = p > pattern && (std::isalnum(static_cast<unsigned char>(p[-1])) || p[-1] == '_');
    const bool ident_after  = std::isalnum(static_cast<unsigned char>(p[1])) || p[1] == '_';
    char       num[16];
    if (!ident_before && !ident_after && (*p == 'T' || *p == 'K' || *p == 'S')) {
      if (*p == 'T') {
        put(tsize == 8 ? "double" : "float");
      } else {
        std::snprintf(num, sizeof(num), "%d", *p == 'K' ? kind : stage);
        put(num);
      }
      p++;
    } else {
      const char c[2] = {*p, 0};
      put(c);
      p++;
    }
  }
  n.name[o] = 0;
}

}  // namespace t8gpu_hip

#endif  // T8GPU_HIP_STAGE_KERNEL_NOTE_HPP
