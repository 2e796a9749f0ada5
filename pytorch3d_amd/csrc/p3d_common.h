// p3d_common.h -- launch plumbing shared by the gfx950 translation units.
#pragma once

// (No ablation switches in the sources: the probe and ablation builds whose records are under profiles/ were made from
// temporary patches into a separate library, P3D_LIB_PATH -- the scripts that read them say so.)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p3d_amd.h"

#define P3D_API extern "C" __attribute__((visibility("default")))

namespace p3d {

constexpr int kWave = 64;  // CDNA wavefront

// ---- per-kernel timing hooks (profile.cpp) ---------------------------------------------
bool profile_enabled();
void profile_begin(const char* name, hipStream_t s);
void profile_end(hipStream_t s);

struct LaunchScope {
  hipStream_t s;
  bool on;
  LaunchScope(const char* name, hipStream_t stream) : s(stream), on(profile_enabled()) {
    if (on) profile_begin(name, s);
  }
  ~LaunchScope() {
    if (on) profile_end(s);
  }
};

inline int launch_status() { return hipGetLastError() == hipSuccess ? P3D_OK : P3D_ERR_LAUNCH; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
  char* base;
  size_t cap;
  size_t off;
  Arena(void* p, size_t c) : base(static_cast<char*>(p)), cap(c), off(0) {}
  template <typename T>
  T* take(size_t n) {
    const size_t bytes = align_up(n * sizeof(T), 256);
    char* r = base ? base + off : nullptr;
    off += bytes;
    return reinterpret_cast<T*>(r);
  }
  bool ok() const { return off <= cap; }
};

#if defined(__HIPCC__)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` strictly below the calling lane
__device__ __forceinline__ int mask_rank(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
#endif

}  // namespace p3d
