// Optional per-launch timing of the MFMA kernels with HIP events recorded on the launch stream (used by bench.py for the
// roofline figure: algorithmic FLOPs of each launch / its measured duration).  Disabled by default: zero overhead.
// The recorder is process-wide by design (autograd issues the backward launches from its own worker thread) and guarded by a
// mutex; it only ever adds event records around launches, it never changes which kernel runs.
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"

namespace {
struct Rec { hipEvent_t a, b; double flops; int kind; };
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
std::atomic<bool> g_on{false};
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
    if (!g_on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r;
    r.a = get_event(); r.b = get_event(); r.flops = flops; r.kind = kind;
    if (!r.a || !r.b) return -1;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void prof_after(hipStream_t s, int token) {
    if (token < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (token < (int)g_recs.size()) (void)hipEventRecord(g_recs[token].b, s);
}
void prof_begin() { std::lock_guard<std::mutex> lk(g_mu); g_recs.clear(); g_pool_next = 0; g_on = true; }
// out[kind*3 + {0,1,2}] = launches, total ms, total flops for kind in 0..2 (fprop, dgrad, wgrad)
int prof_end(double* out) {
    g_on = false;
    std::lock_guard<std::mutex> lk(g_mu);
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
