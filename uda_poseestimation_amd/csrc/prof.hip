// Optional per-launch timing of the MFMA kernels with HIP events recorded on the launch stream (used by bench.py for the
// roofline figure: algorithmic FLOPs of each launch / its measured duration).  Disabled by default: zero overhead.
#include <vector>
#include "common.h"

namespace {
struct Rec { hipEvent_t a, b; double flops; int kind; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
bool g_on = false;
hipEvent_t get_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}
}  // namespace

bool prof_enabled() { return g_on; }
// returns a token (index) or -1
int prof_before(hipStream_t s, int kind, double flops) {
    if (!g_on) return -1;
    Rec r;
    r.a = get_event(); r.b = get_event(); r.flops = flops; r.kind = kind;
    if (!r.a || !r.b) return -1;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void prof_after(hipStream_t s, int token) {
    if (token >= 0) (void)hipEventRecord(g_recs[token].b, s);
}
void prof_begin() { g_recs.clear(); g_pool_next = 0; g_on = true; }
// out[kind*3 + {0,1,2}] = launches, total ms, total flops for kind in 0..2 (fprop, dgrad, wgrad)
int prof_end(double* out) {
    g_on = false;
    for (int i = 0; i < 9; ++i) out[i] = 0.0;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        out[r.kind * 3 + 0] += 1.0;
        out[r.kind * 3 + 1] += (double)ms;
        out[r.kind * 3 + 2] += r.flops;
    }
    g_recs.clear();
    g_pool_next = 0;
    return UDAPOSE_OK;
}
