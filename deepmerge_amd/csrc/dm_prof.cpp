#include "dm_prof.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "deepmerge_hip.h"

namespace {
struct Rec { std::string name; hipEvent_t a, b; double flops, bytes; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
constexpr size_t kMaxRecs = 1u << 20;

hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
}  // namespace

DmProfScope::DmProfScope(const char *name, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on || g_recs.size() >= kMaxRecs) return;
  Rec r{name, take_event(), take_event(), flops, bytes};
  hipEventRecord(r.a, stream);
  slot = (int)g_recs.size();
  g_recs.push_back(r);
}
DmProfScope::~DmProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < (int)g_recs.size()) hipEventRecord(g_recs[slot].b, stream);
}

extern "C" int dm_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return DM_OK;
}

extern "C" int32_t dm_prof_collect(DmProfRow *rows, int32_t max_rows) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, DmProfRow> agg;
  for (auto &r : g_recs) {
    hipEventSynchronize(r.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    DmProfRow &row = agg[r.name];
    if (row.launches == 0) { std::memset(&row, 0, sizeof(row)); std::strncpy(row.name, r.name.c_str(), sizeof(row.name) - 1); }
    row.launches += 1;
    row.total_ms += ms;
    row.total_flops += r.flops;
    row.total_bytes += r.bytes;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  int32_t n = 0;
  for (auto &kv : agg) {
    if (n >= max_rows) break;
    rows[n++] = kv.second;
  }
  return n;
}
