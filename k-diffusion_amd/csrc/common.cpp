// Error reporting + per-launch event timing shared by all entry points of libkdiff_hip.so.
#include "kd_common.h"
#include <map>
#include <mutex>

namespace kd {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static bool g_prof = false;
static std::vector<ProfRec> g_recs;

bool prof_on() { return g_prof; }

void prof_begin(const char* name, double flops, double bytes, hipStream_t s) {
  ProfRec r;
  r.name = name; r.flops = flops; r.bytes = bytes;
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
}

void prof_end(hipStream_t s) { (void)hipEventRecord(g_recs.back().e1, s); }

}  // namespace kd

using namespace kd;

// ---- library options (explicit switches instead of environment variables read inside the library) -------------------------
namespace kd {
static std::mutex g_opt_mu;
static std::map<std::string, int> g_opts;
int option(const char* name, int dflt) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  auto it = g_opts.find(name);
  return it == g_opts.end() ? dflt : it->second;
}
}  // namespace kd

extern "C" int kd_set_option(const char* name, int value) {
  if (!name) return fail(KD_EINVAL, "kd_set_option: null name");
  std::lock_guard<std::mutex> lk(g_opt_mu);
  g_opts[name] = value;
  return KD_OK;
}
extern "C" int kd_get_option(const char* name, int dflt) { return name ? option(name, dflt) : dflt; }

extern "C" int kd_version(void) { return 200; }
extern "C" const char* kd_last_error(void) { return err_buf(); }

extern "C" int kd_prof_enable(int on) { g_prof = on != 0; return KD_OK; }
extern "C" int kd_prof_count(void) { return (int)g_recs.size(); }
extern "C" int kd_prof_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes) {
  if (i < 0 || i >= (int)g_recs.size()) return fail(KD_EINVAL, "kd_prof_get: index %d out of range", i);
  ProfRec& r = g_recs[i];
  if (hipEventSynchronize(r.e1) != hipSuccess) return fail(KD_ELAUNCH, "kd_prof_get: event sync failed");
  float t = 0.f;
  if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return fail(KD_ELAUNCH, "kd_prof_get: elapsed failed");
  if (name && name_cap > 0) { strncpy(name, r.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (ms) *ms = t;
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  return KD_OK;
}
extern "C" int kd_prof_reset(void) {
  for (auto& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_recs.clear();
  return KD_OK;
}
