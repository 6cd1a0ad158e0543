// Error bookkeeping + version for libsempyr_hip.so
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/sempyr.h"

static thread_local char g_err[512] = "";

void sempyr_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sempyr_version(void) { return SEMPYR_ABI_VERSION; }
extern "C" const char* sempyr_last_error_string(void) { return g_err; }

// ------------------------------------------------------------------------------------------------------------------
// HBM-bound kernels: opt-in timing (see HbmScope in common.h)
// ------------------------------------------------------------------------------------------------------------------
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"

namespace {
struct HbmProf {
    std::atomic<bool> on{false};
    std::mutex mu;
    std::vector<hipEvent_t> events;        // pairs
    struct Rec { const char* name; double bytes; };
    std::vector<Rec> recs;
    size_t cap = 0;
    std::string report;
} g_hbm;
}

HbmScope::HbmScope(const char* name, double bytes, hipStream_t stream) : slot(-1), st(stream) {
    if (!g_hbm.on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_hbm.mu);
    if (g_hbm.recs.size() >= g_hbm.cap) return;
    slot = (int)g_hbm.recs.size();
    g_hbm.recs.push_back({name, bytes});
    (void)hipEventRecord(g_hbm.events[2 * slot], st);
}
HbmScope::~HbmScope() {
    if (slot >= 0) (void)hipEventRecord(g_hbm.events[2 * slot + 1], st);
}

void sempyr_hbm_profile_begin(int max_launches) {
    std::lock_guard<std::mutex> lk(g_hbm.mu);
    while (g_hbm.events.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_hbm.events.push_back(e);
    }
    g_hbm.cap = g_hbm.events.size() / 2;
    g_hbm.recs.clear();
    g_hbm.on.store(true);
}

void sempyr_hbm_profile_end() {
    g_hbm.on.store(false);
    std::lock_guard<std::mutex> lk(g_hbm.mu);
    struct Agg { long long n = 0; double bytes = 0.0, ms = 0.0; };
    std::map<std::string, Agg> agg;
    for (size_t i = 0; i < g_hbm.recs.size(); ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_hbm.events[2 * i + 1]) != hipSuccess) continue;
        if (hipEventElapsedTime(&t, g_hbm.events[2 * i], g_hbm.events[2 * i + 1]) != hipSuccess) continue;
        Agg& a = agg[g_hbm.recs[i].name];
        a.n += 1; a.bytes += g_hbm.recs[i].bytes; a.ms += t;
    }
    g_hbm.report = "kernel,launches,algorithmic_bytes,ms\n";
    char line[256];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s,%lld,%.0f,%.5f\n", kv.first.c_str(), kv.second.n, kv.second.bytes, kv.second.ms);
        g_hbm.report += line;
    }
    g_hbm.recs.clear();
}

extern "C" int sempyr_profile_hbm_report(char* out, int64_t cap)
{
    std::lock_guard<std::mutex> lk(g_hbm.mu);
    if (!out || cap < 1) { sempyr_set_error("profile_hbm_report: no buffer"); return SEMPYR_ERR_INVALID; }
    if ((int64_t)g_hbm.report.size() + 1 > cap) { sempyr_set_error("profile_hbm_report: buffer too small (%zu needed)", g_hbm.report.size() + 1); return SEMPYR_ERR_INVALID; }
    memcpy(out, g_hbm.report.c_str(), g_hbm.report.size() + 1);
    return 0;
}
