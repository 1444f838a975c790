// Error reporting + per-launch event timing shared by all entry points of libkdiff_hip.so.
#include "kd_common.h"
#include <atomic>
#include <climits>
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

static std::atomic<bool> g_prof{false};
static std::mutex g_rec_mutex;            // launches may come from several host threads (one stream each): the record list is shared
static std::vector<ProfRec> g_recs;

bool prof_on() { return g_prof.load(std::memory_order_relaxed); }

static unsigned g_gen = 0;                // bumped by kd_prof_reset: a scope that began before a reset must not touch the records after it

ProfTicket prof_begin(const char* name, double flops, double bytes, hipStream_t s) {
  ProfRec r;
  r.name = name; r.flops = flops; r.bytes = bytes;
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, s);
  std::lock_guard<std::mutex> lock(g_rec_mutex);
  g_recs.push_back(r);
  return ProfTicket{(int)g_recs.size() - 1, g_gen};
}

void prof_end(ProfTicket t, hipStream_t s) {
  // The end event is recorded UNDER the lock: kd_prof_reset destroys the events under the same lock, so the handle cannot die between
  // the look-up and the record.  A reset between begin and end changes the generation: index t.idx then names another launch's
  // record (or none) and is left alone.
  std::lock_guard<std::mutex> lock(g_rec_mutex);
  if (t.gen == g_gen && t.idx >= 0 && t.idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[t.idx].e1, s);
}

}  // namespace kd

namespace kd {
int cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
}  // namespace kd

using namespace kd;

// ---- library options (explicit switches instead of environment variables read inside the library) -------------------------
// A fixed table of named integers; reads on the launch path are one relaxed atomic load (option names are string literals in
// the callers: the index is resolved once per call site through a function-local static).
namespace kd {
namespace {
struct Opt { const char* name; std::atomic<int> value; std::atomic<bool> set; };
Opt g_opts[] = {{"skinny", {0}, {false}}, {"astat", {0}, {false}}, {"ksplit", {0}, {false}}, {"astat_max_k", {0}, {false}}, {"astat_waves", {0}, {false}},
                {"astat_storewait", {0}, {false}}, {"gemm_debug", {0}, {false}}, {"bf16_fast", {0}, {false}}, {"wstat", {0}, {false}},
                {"wstat_waves", {0}, {false}}, {"wstat_max_slices", {0}, {false}}, {"wstat_prefetch", {0}, {false}}, {"astat_bf16", {0}, {false}},
                {"astat_splits", {0}, {false}}, {"tiled_bm", {0}, {false}}, {"attn_global_qw", {0}, {false}},
                {"patch_fast", {0}, {false}}, {"ffn_fused", {0}, {false}}, {"ffn_fused_256", {0}, {false}}, {"astat_rows", {0}, {false}}, {"tiled_deep", {0}, {false}}, {"code_warm", {0}, {false}}, {"ffn_variant", {0}, {false}},
                {"x3", {0}, {false}}, {"x3_splits", {0}, {false}}, {"ffn_x3", {0}, {false}}, {"ffn_x3_half", {0}, {false}}, {"x3_res", {0}, {false}}, {"attn_x3", {0}, {false}}, {"x3_half", {0}, {false}}, {"x3r", {0}, {false}}, {"x3_unpatch", {0}, {false}}, {"x3r_lw", {0}, {false}}, {"x3r_split", {0}, {false}}, {"tiled_lw", {0}, {false}}, {"x3_min_rows", {0}, {false}}, {"x3r_min_rows", {0}, {false}}, {"ffn_x3_min_panels_256", {0}, {false}}, {"x3s_max_rows", {0}, {false}}, {"x3s_max_wgs", {0}, {false}}, {"x3s_scale_lds", {0}, {false}}, {"b16s_max_rows", {0}, {false}}, {"b16s_max_wgs", {0}, {false}}, {"ffn_bf16_min_rows", {0}, {false}}, {"x3s_trace", {0}, {false}}, {"attn_block_bf16", {0}, {false}}, {"proj_block_bf16", {0}, {false}}, {"mx8", {0}, {false}}, {"mx8_splits", {0}, {false}}, {"mx8_min_rows", {0}, {false}}};
constexpr int N_OPTS = sizeof(g_opts) / sizeof(g_opts[0]);
}  // namespace
int option_index(const char* name) {
  for (int i = 0; i < N_OPTS; ++i)
    if (!strcmp(g_opts[i].name, name)) return i;
  return -1;
}
int option_at(int idx, int dflt) {
  if (idx < 0) return dflt;
  return g_opts[idx].set.load(std::memory_order_relaxed) ? g_opts[idx].value.load(std::memory_order_relaxed) : dflt;
}
}  // namespace kd

extern "C" int kd_set_option(const char* name, int value) {
  if (!name) return fail(KD_EINVAL, "kd_set_option: null name");
  const int i = option_index(name);
  if (i < 0) return fail(KD_EINVAL, "kd_set_option: unknown option '%s'", name);
  if (value == INT_MIN) {                 // back to the built-in default of every call site
    g_opts[i].set.store(false, std::memory_order_relaxed);
    return KD_OK;
  }
  g_opts[i].value.store(value, std::memory_order_relaxed);
  g_opts[i].set.store(true, std::memory_order_relaxed);
  return KD_OK;
}
extern "C" int kd_get_option(const char* name, int dflt) { return name ? option_at(option_index(name), dflt) : dflt; }

extern "C" int kd_version(void) { return 200; }
extern "C" const char* kd_last_error(void) { return err_buf(); }

extern "C" int kd_prof_enable(int on) { g_prof.store(on != 0, std::memory_order_relaxed); return KD_OK; }
extern "C" int kd_prof_count(void) {
  std::lock_guard<std::mutex> lock(g_rec_mutex);
  return (int)g_recs.size();
}
extern "C" int kd_prof_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes) {
  ProfRec r;
  {
    std::lock_guard<std::mutex> lock(g_rec_mutex);
    if (i < 0 || i >= (int)g_recs.size()) return fail(KD_EINVAL, "kd_prof_get: index %d out of range", i);
    r = g_recs[i];
  }
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
  std::lock_guard<std::mutex> lock(g_rec_mutex);
  for (auto& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_recs.clear();
  ++g_gen;
  return KD_OK;
}

// ---- a launch list in one host call (include/kdiff_hip.h: kd_run_list) ------------------------------------------------------------------------
extern "C" int kd_run_list(const KdCall* calls, int n, void* stream, int* failed) {
  if (!calls || n < 0) return kd::fail(KD_EINVAL, "kd_run_list: null list");
  for (int k = 0; k < n; ++k) {
    const KdCall& c = calls[k];
    const int* i = c.i;
    int rc;
    switch (c.op) {
      case KD_OP_GEMM_F32: rc = kd_gemm_f32(static_cast<const KdGemm*>(c.p[0]), stream); break;
      case KD_OP_GEMM_BF16: rc = kd_gemm_bf16(static_cast<const KdGemm*>(c.p[0]), stream); break;
      case KD_OP_FFN_F32: rc = kd_ffn_f32(static_cast<const KdFfn*>(c.p[0]), stream); break;
      case KD_OP_FFN_BF16: rc = kd_ffn_bf16(static_cast<const KdFfn*>(c.p[0]), stream); break;
      case KD_OP_ATTN_GLOBAL_F32:
        rc = kd_attn_global_f32(static_cast<const float*>(c.p[0]), static_cast<float*>(const_cast<void*>(c.p[1])), i[0], i[1], i[2], i[3], static_cast<const float*>(c.p[2]),
                                static_cast<const float*>(c.p[3]), static_cast<const float*>(c.p[4]), c.f, i[4], stream);
        break;
      case KD_OP_ATTN_WINDOW_F32:
        rc = kd_attn_window_f32(static_cast<const float*>(c.p[0]), static_cast<float*>(const_cast<void*>(c.p[1])), i[0], i[1], i[2], i[3], i[4], i[5], i[6],
                                static_cast<const float*>(c.p[2]), static_cast<const float*>(c.p[3]), static_cast<const float*>(c.p[4]), c.f, i[7], stream);
        break;
      case KD_OP_ATTN_NA2D_F32:
        rc = kd_attn_na2d_f32(static_cast<const float*>(c.p[0]), static_cast<float*>(const_cast<void*>(c.p[1])), i[0], i[1], i[2], i[3], i[4], i[5],
                              static_cast<const float*>(c.p[2]), static_cast<const float*>(c.p[3]), static_cast<const float*>(c.p[4]), c.f, i[6], stream);
        break;
      case KD_OP_ATTN_GLOBAL_BF16: rc = kd_attn_global_bf16(c.p[0], const_cast<void*>(c.p[1]), i[0], i[1], i[2], stream); break;
      case KD_OP_ATTN_WINDOW_BF16: rc = kd_attn_window_bf16(c.p[0], const_cast<void*>(c.p[1]), i[0], i[1], i[2], i[3], i[4], i[5], stream); break;
      case KD_OP_ATTN_NA2D_BF16: rc = kd_attn_na2d_bf16(c.p[0], const_cast<void*>(c.p[1]), i[0], i[1], i[2], i[3], i[4], stream); break;
      case KD_OP_NORM_SPLIT_F32:
        rc = kd_norm_split_f32(static_cast<const float*>(c.p[0]), static_cast<const float*>(c.p[1]), i[0], i[1], const_cast<void*>(c.p[2]), const_cast<void*>(c.p[3]), i[2], i[3],
                               c.f, stream);
        break;
      case KD_OP_ATTN_BLOCK_BF16:
        rc = kd_attn_block_bf16(static_cast<const KdGemm*>(c.p[0]), stream);
        break;
      case KD_OP_PROJ_BLOCK_BF16: rc = kd_proj_block_bf16(static_cast<const KdGemm*>(c.p[0]), stream); break;
      case KD_OP_GEMM_MX8: rc = kd_gemm_mx8(static_cast<const KdGemm*>(c.p[0]), stream); break;
      default: rc = kd::fail(KD_EINVAL, "kd_run_list: entry %d names no entry point (op %d)", k, c.op);
    }
    if (rc != KD_OK) {
      if (failed) *failed = k;
      return rc;
    }
  }
  return KD_OK;
}
