// profile.cpp -- optional per-kernel timing with HIP events recorded on the launch stream.
// Off by default; bench.py switches it on to measure the dominant kernel's launch duration
// live (the number the roofline object is built from).
#include <mutex>
#include <string>
#include <vector>

#include "p3d_common.h"

namespace p3d {
namespace {
struct Pending {
  int entry;
  hipEvent_t a, b;
};
struct Entry {
  std::string name;
  int64_t launches;
  double total_ms;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Entry> g_entries;
std::vector<Pending> g_pending;
thread_local Pending g_cur;

int entry_of(const char* name) {
  for (size_t i = 0; i < g_entries.size(); ++i)
    if (g_entries[i].name == name) return (int)i;
  g_entries.push_back(Entry{name, 0, 0.0});
  return (int)g_entries.size() - 1;
}
}  // namespace

bool profile_enabled() { return g_on; }

void profile_begin(const char* name, hipStream_t s) {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_cur.entry = entry_of(name);
  }
  (void)hipEventCreate(&g_cur.a);
  (void)hipEventCreate(&g_cur.b);
  (void)hipEventRecord(g_cur.a, s);
}

void profile_end(hipStream_t s) {
  (void)hipEventRecord(g_cur.b, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending.push_back(g_cur);
}
}  // namespace p3d

using namespace p3d;

P3D_API void p3d_profile_enable(int enable) { g_on = enable != 0; }

P3D_API void p3d_profile_collect(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& p : g_pending) {
    (void)hipEventSynchronize(p.b);
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      g_entries[p.entry].launches += 1;
      g_entries[p.entry].total_ms += ms;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  g_pending.clear();
}

P3D_API int p3d_profile_num_entries(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_entries.size();
}

P3D_API const char* p3d_profile_entry(int i, int64_t* launches, double* total_ms) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (i < 0 || i >= (int)g_entries.size()) return nullptr;
  if (launches) *launches = g_entries[i].launches;
  if (total_ms) *total_ms = g_entries[i].total_ms;
  return g_entries[i].name.c_str();
}

P3D_API void p3d_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& e : g_entries) {
    e.launches = 0;
    e.total_ms = 0.0;
  }
}

P3D_API int p3d_abi_version(void) { return P3D_ABI_VERSION; }

P3D_API const char* p3d_error_string(int code) {
  switch (code) {
    case P3D_OK:
      return "ok";
    case P3D_ERR_INVALID_ARG:
      return "invalid argument";
    case P3D_ERR_K_TOO_LARGE:
      return "Must have points_per_pixel <= 150";
    case P3D_ERR_TOO_MANY_BINS:
      return "too many bins per side (bin_size too small); that's too many!";
    case P3D_ERR_WORKSPACE:
      return "workspace missing or smaller than p3d_*_workspace_bytes()";
    case P3D_ERR_LAUNCH:
      return "HIP kernel launch failed";
    case P3D_ERR_UNSUPPORTED:
      return "unsupported configuration";
    default:
      return "unknown error";
  }
}
