// Tap-plan construction (which input pixel and weight slab every filter tap of every sub-pixel class touches) and the
// three convolution entry points built on the igemm / wgrad kernels.
#include <map>
#include <mutex>
#include <tuple>
#include "conv_plan.h"

int prof_before(hipStream_t s, int kind, double flops);
void prof_after(hipStream_t s, int token);

static double alg_flops(const ConvGeom& g) {
    // algorithmic FLOPs of the convolution (true channel count: 8-channel inputs are 3-channel images padded)
    const double ci = g.Ci == 8 ? 3.0 : (double)g.Ci;
    const double opix = g.transposed ? (double)g.N * g.Hi * g.Wi : (double)g.N * g.Ho() * g.Wo();
    return 2.0 * opix * g.Co * ci * g.KH * g.KW;
}

namespace {

// per (device, geometry class) tap tables; guarded by g_mu (any thread, any device)
std::mutex g_mu;
std::map<std::tuple<int, int, int, int, int, int, int, int>, TapPlan*> g_plans;

// classes of a stride-s transposed mapping: output/in-grad pixel o = s*i + a receives tap k iff k == (a+p) mod s,
// from source pixel i + (a+p-k)/s.
void build_subpixel(TapPlan& tp, int KH, int KW, int KWp, int s, int p) {
    tp.nclass = s * s;
    for (int a = 0; a < s; ++a)
        for (int b = 0; b < s; ++b) {
            IgClass& c = tp.cls[a * s + b];
            c.tap_off = (int)tp.taps.size();
            c.oa = a;
            c.ob = b;
            for (int kh = 0; kh < KH; ++kh) {
                if (((a + p - kh) % s + s) % s) continue;
                for (int kw = 0; kw < KW; ++kw) {
                    if (((b + p - kw) % s + s) % s) continue;
                    tp.taps.push_back(IgTap{(a + p - kh) / s, (b + p - kw) / s, kh * KWp + kw, a * s + b});
                }
            }
            c.ntaps = (int)tp.taps.size() - c.tap_off;
        }
}
void build_direct(TapPlan& tp, int KH, int KW, int KWp, int p) {
    tp.nclass = 1;
    tp.cls[0] = IgClass{0, KH * KWp, 0, 0};
    for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KWp; ++kw) tp.taps.push_back(IgTap{kh - p, kw - p, kh * KWp + kw, 0});
}

}  // namespace

// every tap of every class is (0, 0) -> weight slab 0 (1x1 kernels, any stride): the kernels then skip the tap-table read
static int plan_is_tap0(const TapPlan& tp) {
    for (const IgTap& t : tp.taps) if (t.dy != 0 || t.dx != 0 || t.widx != 0) return 0;
    for (int c = 0; c < tp.nclass; ++c) if (tp.cls[c].ntaps > 1) return 0;
    return 1;
}

const TapPlan* get_tap_plan(const ConvGeom& g, int direction, hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_tuple(dev, g.KH, g.KW, g.stride, g.pad, g.transposed, g.KWp(), direction);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    // building a plan allocates and copies synchronously: never during a stream capture (prepare the geometry first)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return nullptr;
    TapPlan* tp = new TapPlan();
    const int KWp = g.KWp();
    const bool sub = (direction == 0) ? (g.transposed != 0) : (g.transposed == 0 && g.stride > 1);
    if (g.stride > 2 && sub && direction != 2) { delete tp; return nullptr; }
    if (direction == 2) {
        // grouped wgrad of a Ci == 8 conv: one tap per filter ROW (its KWp column taps x 8 channels are contiguous in x)
        tp->nclass = 1;
        tp->cls[0] = IgClass{0, g.KH, 0, 0};
        for (int kh = 0; kh < g.KH; ++kh) tp->taps.push_back(IgTap{kh - g.pad, -g.pad, kh, 0});
    } else if (sub) {
        build_subpixel(*tp, g.KH, g.KW, KWp, g.stride, g.pad);
    } else if (direction == 0 || g.transposed) {
        build_direct(*tp, g.KH, g.KW, KWp, g.pad);          // fprop of a conv, or dgrad of a transposed conv
    } else {
        tp->nclass = 1;                                      // dgrad of a stride-1 conv: mirrored offsets
        tp->cls[0] = IgClass{0, g.KH * g.KW, 0, 0};
        for (int kh = 0; kh < g.KH; ++kh)
            for (int kw = 0; kw < g.KW; ++kw) tp->taps.push_back(IgTap{g.pad - kh, g.pad - kw, kh * KWp + kw, 0});
    }
    IgTap* d = nullptr;
    const size_t bytes = (tp->taps.size() + 4) * sizeof(IgTap);
    if (hipMalloc((void**)&d, bytes) != hipSuccess) { delete tp; return nullptr; }
    std::vector<IgTap> padded(tp->taps);
    for (int i = 0; i < 4; ++i) padded.push_back(IgTap{1 << 20, 1 << 20, 0, 0});
    if (hipMemcpy(d, padded.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { delete tp; return nullptr; }
    tp->d_taps = d;
    g_plans[key] = tp;
    return tp;
}

int conv_prepare(const ConvGeom& g) {
    if (!get_tap_plan(g, 0)) return UDAPOSE_ERR_UNSUPPORTED;
    if (!g.reflect && !g.upsample && !g.smallc() && !get_tap_plan(g, 1)) return UDAPOSE_ERR_UNSUPPORTED;
    if (g.smallc() && !g.transposed && g.KWp() == 8 && !get_tap_plan(g, 2)) return UDAPOSE_ERR_UNSUPPORTED;
    return UDAPOSE_OK;
}

// geometry the 3x3 run-staged igemm form takes (fprop and data gradient alike: the gradient of such a conv is such a conv)
int conv_h3_ok(const ConvGeom& g) {
    const bool ok = g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && !g.reflect && !g.upsample && g.Ci % 64 == 0 &&
                    g.Co % 64 == 0 && g.Wi <= 64;
    return ok ? (g.Wi <= 16 ? 3 : (g.Wi <= 32 ? 2 : 1)) : 0;
}

int conv_stat_rows(const ConvGeom& g) {
    const int nclass = g.transposed ? g.stride * g.stride : 1;
    const int M = g.transposed ? g.N * g.Hi * g.Wi : g.N * g.Ho() * g.Wo();
    const TapPlan* tp = get_tap_plan(g, 0);
    const int K = (tp ? tp->cls[0].ntaps : g.KH * g.KW) * g.Ci;     // same K as conv_fprop passes
    return igemm_stat_rows(M, g.Co, nclass, igemm_pick_tile(M, g.Co, nclass, K, conv_h3_ok(g), g.policy()));
}

int conv_fprop(hipStream_t s, const ConvGeom& g, const elem_t* x, const elem_t* w_fwd, void* y, const ConvEpilogue& e) {
    if (patch_conv_ok(g, e)) {
        const int tok = prof_before(s, 0, alg_flops(g));
        const int rc = patch_conv_fprop(s, g, x, w_fwd, y, e);
        prof_after(s, tok);
        return rc;
    }
    const TapPlan* tp = get_tap_plan(g, 0, s);
    if (!tp) return UDAPOSE_ERR_NOT_PREPARED;
    if (g.transposed && (g.reflect || g.upsample)) return UDAPOSE_ERR_UNSUPPORTED;
    IgParams p{};
    p.x = x; p.w = w_fwd; p.y = y; p.res = e.res; p.bias = e.bias; p.scale = e.scale; p.stats = e.stats; p.taps = tp->d_taps;
    p.N = g.N; p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci;
    p.Ho = g.Ho(); p.Wo = g.Wo(); p.Co = g.Co;
    if (g.transposed) { p.Hg = g.Hi; p.Wg = g.Wi; p.s = 1; p.os = g.stride; }
    else { p.Hg = p.Ho; p.Wg = p.Wo; p.s = g.stride; p.os = 1; }
    p.M = g.N * p.Hg * p.Wg;
    p.wtaps = g.wtaps();
    p.flags = (g.reflect ? IG_FLAG_REFLECT : 0) | (g.upsample ? IG_FLAG_UPSAMPLE : 0) | (e.relu ? IG_FLAG_RELU : 0) |
              (e.out_f32 ? IG_FLAG_OUT_F32 : 0) | (g.smallc() ? IG_FLAG_SMALLC : 0) | ((e.f32 || e.split) ? IG_FLAG_F32 : 0) |
              (e.split ? IG_FLAG_SPLIT : 0);
    p.nclass = tp->nclass;
    for (int c = 0; c < tp->nclass; ++c) p.cls[c] = tp->cls[c];
    p.tap0 = plan_is_tap0(*tp);
    const int tok = prof_before(s, 0, alg_flops(g));
    const int rc = igemm_launch(p, igemm_pick_tile(p.M, p.Co, p.nclass, p.cls[0].ntaps * p.Ci, conv_h3_ok(g), g.policy()), s, g.policy());
    prof_after(s, tok);
    return rc;
}

static int dgrad_params(const ConvGeom& g, IgParams& p, hipStream_t s = nullptr) {
    if (g.reflect || g.upsample || g.smallc()) return UDAPOSE_ERR_UNSUPPORTED;
    const TapPlan* tp = get_tap_plan(g, 1, s);
    if (!tp) return UDAPOSE_ERR_NOT_PREPARED;
    p.taps = tp->d_taps;
    p.N = g.N; p.Hi = g.Ho(); p.Wi = g.Wo(); p.Ci = g.Co;     // igemm "input" is dy
    p.Ho = g.Hi; p.Wo = g.Wi; p.Co = g.Ci;                     // igemm "output" is dx
    if (g.transposed) { p.Hg = g.Hi; p.Wg = g.Wi; p.s = g.stride; p.os = 1; }
    else if (g.stride > 1) { p.Hg = (g.Hi + g.stride - 1) / g.stride; p.Wg = (g.Wi + g.stride - 1) / g.stride; p.s = 1; p.os = g.stride; }
    else { p.Hg = g.Hi; p.Wg = g.Wi; p.s = 1; p.os = 1; }
    p.M = g.N * p.Hg * p.Wg;
    p.wtaps = g.wtaps();
    p.nclass = tp->nclass;
    for (int c = 0; c < tp->nclass; ++c) p.cls[c] = tp->cls[c];
    p.tap0 = plan_is_tap0(*tp);
    return UDAPOSE_OK;
}

int conv_dgrad_stat_rows(const ConvGeom& g) {
    IgParams p{};
    const int rc = dgrad_params(g, p);
    if (rc != UDAPOSE_OK) return rc;
    return igemm_stat_rows(p.M, p.Co, p.nclass, igemm_pick_tile(p.M, p.Co, p.nclass, p.cls[0].ntaps * p.Ci, conv_h3_ok(g), g.policy()));
}

int conv_dgrad(hipStream_t s, const ConvGeom& g, const elem_t* dy, const elem_t* w_bwd, void* dx, const elem_t* res, int out_f32, DgradBnStat* bs) {
    IgParams p{};
    const int rc0 = dgrad_params(g, p, s);
    if (rc0 != UDAPOSE_OK) return rc0;
    p.x = dy; p.w = w_bwd; p.y = dx; p.res = res;
    p.flags = (out_f32 ? IG_FLAG_OUT_F32 : 0) | ((!g.transposed && g.stride == 1) ? IG_FLAG_MIRROR : 0);   // (stride-1 data gradient: mirrored taps)
    const int tile = igemm_pick_tile(p.M, p.Co, p.nclass, p.cls[0].ntaps * p.Ci, conv_h3_ok(g), g.policy());
    if (bs) {
        p.bs_y = bs->y; p.bs_z = bs->mask ? (const elem_t*)bs->mask : bs->z; p.bs_mean = bs->mean; p.bs_invstd = bs->invstd; p.bs_gamma = bs->gamma; p.bs_beta = bs->beta;
        if (bs->mask) p.flags |= IG_FLAG_BSMASK;
        p.stats = bs->slab;
        bs->rows = igemm_stat_rows(p.M, p.Co, p.nclass, tile);
        if (!bs->y || !bs->slab || !bs->mean || !bs->invstd || (!bs->z && !bs->mask && (!bs->gamma || !bs->beta))) return UDAPOSE_ERR_ARG;
    }
    const int tok = prof_before(s, 1, alg_flops(g));
    const int rc = igemm_launch(p, tile, s, g.policy());
    prof_after(s, tok);
    return rc;
}

int conv_wgrad_params(const ConvGeom& g, const elem_t* dy, const elem_t* x, float* dw, int rows_valid, WgParams* out, double* flops) {
    if (g.reflect || g.upsample) return UDAPOSE_ERR_UNSUPPORTED;
    const bool rowtap = rows_valid == -2;          // grouped Ci == 8 form (see wgrad_dma_body)
    if (rowtap && (!g.smallc() || g.transposed || g.KWp() != 8)) return UDAPOSE_ERR_UNSUPPORTED;
    const TapPlan* tp = get_tap_plan(g, rowtap ? 2 : 0);
    if (!tp) return UDAPOSE_ERR_NOT_PREPARED;
    WgParams p{};
    p.dy = dy; p.x = x; p.dw = dw; p.taps = tp->d_taps;
    p.kw = rowtap ? g.KW : 0;
    p.N = g.N; p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci;
    p.Ho = g.Ho(); p.Wo = g.Wo(); p.Co = g.Co;
    if (g.transposed) { p.Hg = g.Hi; p.Wg = g.Wi; p.s = 1; p.os = g.stride; }
    else { p.Hg = p.Ho; p.Wg = p.Wo; p.s = g.stride; p.os = 1; }
    p.M = g.N * p.Hg * p.Wg;
    p.wtaps = rowtap ? g.KH : g.wtaps();
    p.flags = (g.smallc() ? IG_FLAG_SMALLC : 0) | (g.transposed ? WG_FLAG_SWAP : 0);
    if (g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.transposed && !g.smallc() && !g.reflect && !g.upsample) p.flags |= WG_FLAG_ROW3_OK;
    p.nclass = tp->nclass;
    for (int c = 0; c < tp->nclass; ++c) p.cls[c] = tp->cls[c];
    p.total_taps = (int)tp->taps.size();
    const int Rdim = g.transposed ? g.Ci : g.Co;
    p.rows_valid = rows_valid < 0 ? Rdim : rows_valid;
    *out = p;   // (rows_valid == -2 selects the row-tap form of a Ci == 8 conv for the grouped launch)
    if (flops) *flops = alg_flops(g);
    return UDAPOSE_OK;
}

int conv_wgrad(hipStream_t s, const ConvGeom& g, const elem_t* dy, const elem_t* x, float* dw, int accumulate, int rows_valid) {
    WgParams p;
    double fl = 0.0;
    const int rc0 = conv_wgrad_params(g, dy, x, dw, rows_valid, &p, &fl);
    if (rc0 != UDAPOSE_OK) return rc0;
    const int Rdim = g.transposed ? g.Ci : g.Co, Cdim = g.transposed ? g.Co : g.Ci;
    const int tok = prof_before(s, 2, fl);
    const int rc = wgrad_launch(p, wgrad_pick_tile(Rdim, Cdim, g.smallc(), g.policy()), accumulate, s, g.policy());
    prof_after(s, tok);
    return rc;
}

int conv_prof_before(hipStream_t s, int kind, double flops) { return prof_before(s, kind, flops); }
void conv_prof_after(hipStream_t s, int token) { prof_after(s, token); }
