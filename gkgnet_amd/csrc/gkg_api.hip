// gkg_api.hip — version / error-string part of the C ABI (include/gkg_hip.h).
#include <stdio.h>
#include <string.h>

#include "gkg_common.h"

static thread_local char g_err[256] = "";

int gkg_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int gkg_fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

extern "C" int gkg_version(void) { return GKG_ABI_VERSION; }

// Identifier of the graph capture `stream` is recording into (0: not capturing).  Lets host-side caches that must emit a
// refresh kernel once per captured step (the x6 weight planes) tell one capture from the next.
extern "C" unsigned long long gkg_stream_capture_id(void* stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return st == hipStreamCaptureStatusActive ? (id ? id : 1ull) : 0ull;
}

extern "C" const char* gkg_last_error_string(void) { return g_err; }

// ------------------------------------------------------------------------------------------ profiling
#include <mutex>
#include <vector>

namespace {
struct ProfRec { int kernel; hipEvent_t a, b; };
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_recs;          // recorded, not yet read
std::vector<hipEvent_t> g_free;       // event pool
double g_ms[GKG_PROF_NUM] = {0};
long g_cnt[GKG_PROF_NUM] = {0};
double g_work[GKG_PROF_NUM] = {0};
constexpr size_t kMaxPending = 1 << 16;

hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
void drain_locked() {
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_ms[r.kernel] += ms;
      g_cnt[r.kernel] += 1;
    }
    g_free.push_back(r.a);
    g_free.push_back(r.b);
  }
  g_recs.clear();
}
}  // namespace

GkgProfScope::GkgProfScope(int kernel_id, hipStream_t s, double work) : slot(-1), st(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_recs.size() >= kMaxPending) return;
  if (kernel_id >= 0 && kernel_id < GKG_PROF_NUM) g_work[kernel_id] += work;
  ProfRec r{kernel_id, get_event(), get_event()};
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}

void gkg_prof_add_work(int kernel_id, double work) {
  if (!g_prof_on || kernel_id < 0 || kernel_id >= GKG_PROF_NUM) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_work[kernel_id] += work;
}

GkgProfScope::~GkgProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if ((size_t)slot < g_recs.size()) (void)hipEventRecord(g_recs[slot].b, st);
}

extern "C" void gkg_prof_enable(int on) { g_prof_on = on != 0; }

extern "C" void gkg_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  drain_locked();
  for (int i = 0; i < GKG_PROF_NUM; ++i) { g_ms[i] = 0; g_cnt[i] = 0; g_work[i] = 0; }
}

extern "C" double gkg_prof_work(int kernel_id) {
  if (kernel_id < 0 || kernel_id >= GKG_PROF_NUM) return 0.0;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  return g_work[kernel_id];
}

extern "C" int gkg_prof_read(int kernel_id, double* total_ms, long* launches) {
  if (kernel_id < 0 || kernel_id >= GKG_PROF_NUM || !total_ms || !launches)
    return gkg_fail(GKG_ERR_SHAPE, "gkg_prof_read: bad kernel id / null output");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  drain_locked();
  *total_ms = g_ms[kernel_id];
  *launches = g_cnt[kernel_id];
  return 0;
}
