// profile.hip -- profiling: HIP events on the engine's own streams (ProfScope, engine_internal.h) and the sa_profile_* entries.
#include "engine_internal.h"

namespace sa {

std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof;
bool g_prof_on = false;
hipEvent_t g_prof_epoch[PROF_MAX_DEV] = {};
int g_trace_scopes = 0;

int prof_id(const char* name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = 0; i < g_prof.size(); i++)
        if (g_prof[i].name == name) return (int)i;
    ProfEntry e;
    e.name = name;
    g_prof.push_back(e);
    return (int)g_prof.size() - 1;
}

void prof_flush(Slot* sl) {  // call after the slot's stream has been synchronised
    if (sl->prof_pending.empty()) { sl->events_used = 0; return; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : sl->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            g_prof[r.id].total_ms += ms;
            g_prof[r.id].launches += 1;
            float t0 = 0;
            if (sl->dev >= 0 && sl->dev < PROF_MAX_DEV && g_prof_epoch[sl->dev] &&
                hipEventElapsedTime(&t0, g_prof_epoch[sl->dev], r.e0) == hipSuccess) {
                if (g_prof[r.id].spans.size() < ((size_t)1 << 20)) g_prof[r.id].spans.push_back({sl->dev, t0, t0 + ms});  // (bounded: a
                // profile left enabled without a reset keeps its totals, sa_profile_busy_ms then covers the first 2^20 launches)
            }
            else (void)hipGetLastError();
        }
    }
    sl->prof_pending.clear();
    sl->events_used = 0;
}

}  // namespace sa

using namespace sa;

extern "C" {

void sa_profile_enable(int on) { g_prof_on = on != 0; }
void sa_profile_reset(void) {
    for (auto* dc : g_dev) {  // a fresh epoch per device: launch times are kept relative to it
        if (dc->dev < 0 || dc->dev >= PROF_MAX_DEV) continue;
        check_set_device(dc->dev, "profile reset");
        if (!g_prof_epoch[dc->dev] && hipEventCreate(&g_prof_epoch[dc->dev]) != hipSuccess) { g_prof_epoch[dc->dev] = nullptr; continue; }
        hipEventRecord(g_prof_epoch[dc->dev], dc->admin);
        hipEventSynchronize(g_prof_epoch[dc->dev]);
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) { e.total_ms = 0; e.launches = 0; e.spans.clear(); }
}
// Time during which AT LEAST ONE launch of scope `name` was running (union of its launches' [start, end] over all slots; summed
// over devices).  With several calls in flight a launch's own duration says how long it shared the GPU, not how fast it is.
double sa_profile_busy_ms(const char* name) {
    if (!name) return 0.0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        if (e.name != name) continue;
        std::vector<ProfSpan> v = e.spans;
        std::sort(v.begin(), v.end(), [](const ProfSpan& a, const ProfSpan& b) { return a.dev != b.dev ? a.dev < b.dev : a.t0 < b.t0; });
        double busy = 0;
        size_t i = 0;
        while (i < v.size()) {
            float lo = v[i].t0, hi = v[i].t1;
            size_t j = i + 1;
            while (j < v.size() && v[j].dev == v[i].dev && v[j].t0 <= hi) { hi = std::max(hi, v[j].t1); j++; }
            busy += hi - lo;
            i = j;
        }
        return busy;
    }
    return 0.0;
}
int sa_profile_num_entries(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_prof.size();
}
int sa_profile_get(int i, char* name_buf, size_t name_cap, double* total_ms, uint64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (i < 0 || i >= (int)g_prof.size()) return -1;
    if (name_buf && name_cap) {
        strncpy(name_buf, g_prof[i].name.c_str(), name_cap - 1);
        name_buf[name_cap - 1] = '\0';
    }
    if (total_ms) *total_ms = g_prof[i].total_ms;
    if (launches) *launches = g_prof[i].launches;
    return 0;
}

}  // extern "C"
