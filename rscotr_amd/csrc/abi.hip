// Library-level entries of the rscotr C ABI (version, error string).
#include "common.h"
#include "rscotr.h"
#include <mutex>
#include <string>
#include <vector>

namespace rscotr {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

// ---- launch-site profiling ------------------------------------------------------------------------------
namespace {
struct ProfRec {
  hipEvent_t e0, e1;
  double work;
  int kind;
  char name[112];
};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
int g_prof_used = 0;
int g_prof_every[PROF_KINDS] = {0, 0, 0, 0, 0};
long g_prof_seen[PROF_KINDS] = {0, 0, 0, 0, 0};
}  // namespace

ProfScope::ProfScope(int kind, double work, hipStream_t s, const char* fmt, ...) : slot(-1), stream(s) {
  if (g_prof_every[kind] <= 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_every[kind] <= 0 || g_prof_used >= (int)g_prof.size()) return;
  if (g_prof_seen[kind]++ % g_prof_every[kind]) return;
  slot = g_prof_used++;
  ProfRec& r = g_prof[slot];
  r.work = work;
  r.kind = kind;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(r.name, sizeof(r.name), fmt, ap);
  va_end(ap);
  // (plain records: inside a stream capture they would become dependency nodes without a timestamp, and
  // hipEventRecordWithFlags(hipEventRecordExternal) is rejected during capture on ROCm 7.2 — tried)
  (void)hipEventRecord(r.e0, stream);
}

ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].e1, stream);
}
}  // namespace rscotr

// Profiling control (bench.py): every_* = record one launch in n of that kind (0 = off); max_records event
// pairs are created here and destroyed by rscotr_prof_disable().  rscotr_prof_get() needs the stream idle.
extern "C" int rscotr_prof_enable(int every_gemm, int every_msda_fwd, int every_msda_bwd, int max_records) {
  std::lock_guard<std::mutex> lk(rscotr::g_prof_mu);
  if (max_records < 0) return rscotr::fail(RSCOTR_E_ARG, "rscotr_prof_enable: negative max_records");
  for (auto& r : rscotr::g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  rscotr::g_prof.assign(max_records, rscotr::ProfRec{});
  for (auto& r : rscotr::g_prof) {
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess)
      return rscotr::fail(RSCOTR_E_LAUNCH, "rscotr_prof_enable: hipEventCreate failed");
  }
  rscotr::g_prof_used = 0;
  const int any = (every_gemm > 0 || every_msda_fwd > 0 || every_msda_bwd > 0) ? 1 : 0;
  const int ev[rscotr::PROF_KINDS] = {every_gemm, every_msda_fwd, every_msda_bwd, any, any};
  for (int k = 0; k < rscotr::PROF_KINDS; ++k) { rscotr::g_prof_every[k] = ev[k]; rscotr::g_prof_seen[k] = 0; }
  return RSCOTR_OK;
}

// n EMPTY brackets (an event pair with nothing in between, kind PROF_HBM, name "rscotr::empty_bracket") on `stream`: what the
// bracketing itself adds to every sampled duration (the second record's packet is processed after the first one's: ~5 us on
// ROCm 7.2 / MI355X — a quarter of a 20 us launch).  bench.py subtracts their mean from every sample.
extern "C" int rscotr_prof_empty(int n, void* stream) {
  for (int i = 0; i < n; ++i) { rscotr::ProfScope prof(rscotr::PROF_HBM, 0.0, static_cast<hipStream_t>(stream), "rscotr::empty_bracket"); }
  return RSCOTR_OK;
}

extern "C" int rscotr_prof_pause(void) {  // stop recording, keep the records
  std::lock_guard<std::mutex> lk(rscotr::g_prof_mu);
  for (int k = 0; k < rscotr::PROF_KINDS; ++k) rscotr::g_prof_every[k] = 0;
  return rscotr::g_prof_used;
}

extern "C" int rscotr_prof_get(int i, int* kind, double* work, float* ms, char* name, int name_len) {
  std::lock_guard<std::mutex> lk(rscotr::g_prof_mu);
  if (i < 0 || i >= rscotr::g_prof_used) return rscotr::fail(RSCOTR_E_ARG, "rscotr_prof_get: index out of range");
  const rscotr::ProfRec& r = rscotr::g_prof[i];
  float t = 0.f;
  if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess)
    return rscotr::fail(RSCOTR_E_LAUNCH, "rscotr_prof_get: events not complete");
  *kind = r.kind; *work = r.work; *ms = t;
  if (name && name_len > 0) snprintf(name, name_len, "%s", r.name);
  return RSCOTR_OK;
}

extern "C" int rscotr_prof_disable(void) {
  std::lock_guard<std::mutex> lk(rscotr::g_prof_mu);
  for (int k = 0; k < rscotr::PROF_KINDS; ++k) rscotr::g_prof_every[k] = 0;
  for (auto& r : rscotr::g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  rscotr::g_prof.clear();
  rscotr::g_prof_used = 0;
  return RSCOTR_OK;
}

extern "C" int rscotr_version(void) { return RSCOTR_ABI_VERSION; }

extern "C" const char* rscotr_last_error(void) { return rscotr::err_buf(); }

// Number of devices visible to the library; <0 when the HIP runtime cannot be initialised.
extern "C" int rscotr_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return rscotr::fail(RSCOTR_E_ARCH, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return n;
}
