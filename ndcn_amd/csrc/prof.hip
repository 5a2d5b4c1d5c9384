#include "prof.h"

#include <mutex>
#include <vector>

#include "common.h"

namespace ndcn {

namespace {
struct Rec { int kind; hipEvent_t a, b; double bytes, flops; };
struct Prof {
    bool on = false;
    bool paused = false;
    bool overflow = false;
    std::vector<Rec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    std::mutex mu;
};
Prof g_prof;
constexpr size_t kMaxRecs = 1 << 16;
}  // namespace

ProfScope::ProfScope(int kind, hipStream_t s, double bytes, double flops) : slot(-1), st(s) {
    if (!g_prof.on || g_prof.paused) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.recs.size() >= kMaxRecs) { g_prof.overflow = true; return; }
    Rec r;
    r.kind = kind; r.bytes = bytes; r.flops = flops;
    if (!g_prof.pool.empty()) {
        r.a = g_prof.pool.back().first; r.b = g_prof.pool.back().second;
        g_prof.pool.pop_back();
    } else {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    }
    (void)hipEventRecord(r.a, st);
    g_prof.recs.push_back(r);
    slot = (int)g_prof.recs.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    (void)hipEventRecord(g_prof.recs[slot].b, st);
}

bool prof_pause(bool on) {       // returns the previous state (nested users restore it)
    const bool was = g_prof.paused;
    g_prof.paused = on;
    return was;
}

}  // namespace ndcn

using namespace ndcn;

extern "C" {

NDCN_API int ndcn_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    return NDCN_OK;
}

// Drains the recorded launches: per kind {count, total ms, total algorithmic bytes, total flops}.
// h_out has PROF_NKINDS * 4 doubles.  Synchronises on the recorded events.
NDCN_API int ndcn_prof_read(double *h_out, int n_kinds) {
    NDCN_CHECK_ARG(h_out && n_kinds >= PROF_NKINDS, "output too small");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int i = 0; i < n_kinds * 4; ++i) h_out[i] = 0.0;
    for (Rec &r : g_prof.recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            h_out[r.kind * 4 + 0] += 1.0;
            h_out[r.kind * 4 + 1] += (double)ms;
            h_out[r.kind * 4 + 2] += r.bytes;
            h_out[r.kind * 4 + 3] += r.flops;
        }
        g_prof.pool.emplace_back(r.a, r.b);
    }
    g_prof.recs.clear();
    const int rc = g_prof.overflow ? 1 : 0;
    g_prof.overflow = false;
    return rc;
}

NDCN_API int ndcn_prof_kinds(void) { return PROF_NKINDS; }

}  // extern "C"
