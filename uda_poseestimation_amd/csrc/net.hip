// PoseResNet executor: the whole forward / backward of the reference's pose network
// (lib/models/pose_resnet.py:59-91 over the torchvision ResNet-v1.5 trunk, lib/models/resnet.py:25-40) as one C++ plan
// that enqueues hand-written gfx950 kernels on a HIP stream.  No allocation, no synchronisation, no host round trip
// inside forward/backward (graph-capturable after one warm-up call that fills the tap-plan cache).
//
// Data layout in HBM: activations NHWC bf16 (pre-BN conv output y and post-BN/ReLU z both kept for backward), weights
// bf16 packed [Co][taps][Ci] (fprop) and [Ci][taps][Co] (dgrad) from the fp32 channels_last master copies, BN statistics
// fp32.  Parameters are addressed by index in torch `.parameters()` order, buffers in `.buffers()` order.
#include <algorithm>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "conv_plan.h"

int pw_nchw_f32_to_nhwc_bf16(hipStream_t, const float*, elem_t*, int, int, int, int);
int pw_nhwc_to_nchw_f32(hipStream_t, const void*, int, float*, int, int, int, int, const float*, const float*);
int pw_cast_f32_bf16(hipStream_t, const float*, elem_t*, size_t);
int pw_transpose_cast(hipStream_t, const float*, elem_t*, int, int, int);
int pw_pack_strided(hipStream_t, const float*, elem_t*, int, int, int, int, int, int, long, long, long, long);
int pw_unpack_strided(hipStream_t, const float*, float*, int, int, int, int, int, int, long, long, long, long, float);
int pw_bn_finalize(hipStream_t, const float*, int, int, double, const float*, const float*, float*, float*, long long*, float, float, float*, float*,
                   float*, float*, const float*);
int pw_bn_eval_coeff(hipStream_t, int, const float*, const float*, const float*, const float*, float, float*, float*);
int pw_bn_apply(hipStream_t, const elem_t*, const elem_t*, elem_t*, size_t, int, const float*, const float*, int, unsigned char*, int);
int pw_bn_bwd_rows(size_t);
int pw_bn_bwd(hipStream_t, const void*, int, const elem_t*, const elem_t*, elem_t*, elem_t*, size_t, int, const float*, const float*, const float*, int,
              float*, float*, float*, float*, float, const float*, int);
int pw_bn_bwd_pre(hipStream_t, const void*, int, const elem_t*, elem_t*, size_t, int, const float*, const float*, const float*, const float*, int, float*,
                  float*, float*, float, int, int);
int pw_maxpool3x3s2_fwd(hipStream_t, const elem_t*, elem_t*, unsigned char*, int, int, int, int);
int pw_maxpool3x3s2_bwd(hipStream_t, const elem_t*, const unsigned char*, elem_t*, int, int, int, int);
int pw_bn_relu_maxpool3x3s2(hipStream_t, const elem_t*, elem_t*, unsigned char*, int, int, int, int, const float*, const float*);
int pw_bn_bwd_pooled(hipStream_t, const elem_t*, const unsigned char*, int, int, const elem_t*, elem_t*, size_t, int, const float*, const float*,
                     const float*, float*, float*, float*, float*, float, const float*);
int pw_plane_sum(hipStream_t, const float*, float*, int, int, int, float);
int pw_bn_running_update(hipStream_t, const float*, int, float*, float*, long long*, float);
int pw_bn_running_update_multi(hipStream_t, const BnRunJob*, int, int, const void*, float);
int pw_bn_train_fused(hipStream_t, const elem_t*, const elem_t*, elem_t*, size_t, int, const float*, int, const float*, const float*, float*, float*,
                      long long*, float, float, float*, int, int, unsigned char*);
int pw_zero_multi(hipStream_t, const ZeroJob*, int, void*);
int pw_pack_multi(hipStream_t, const void*, const int*, const int*, int);
int pw_nchw_f32_to_nhwc_f32(hipStream_t, const float*, float*, int, int, int, int);
size_t opt_tail_job_bytes();
int opt_chunk();
void opt_tail_job_fill(void*, float*, const float*, float*, float*, float*, void*, void*, void*, void*, int, int, int, int, long long);
int opt_tail(hipStream_t, const void*, const int*, const int*, int, float, float, float, float, float, int, float, float*, float, float, int, long long, int);
int pw_transpose_f32(hipStream_t, const float*, float*, int, int, int);
int pw_pack_strided_f32(hipStream_t, const float*, float*, int, int, int, int, int, int, long, long, long, long);
int pw_bn_apply_f32(hipStream_t, const float*, const float*, float*, size_t, int, const float*, const float*, int);
int pw_maxpool3x3s2_fwd_f32(hipStream_t, const float*, float*, unsigned char*, int, int, int, int);
int pw_nchw_f32_to_nhwc_split(hipStream_t, const float*, void*, int, int, int, int);
int pw_f32_to_split(hipStream_t, const float*, void*, size_t);
int pw_transpose_split(hipStream_t, const float*, void*, int, int, int);
int pw_pack_strided_split(hipStream_t, const float*, void*, int, int, int, int, int, int, long, long, long, long);
int pw_bn_apply_split(hipStream_t, const float*, const void*, void*, size_t, int, const float*, const float*, int);
int pw_bn_train_fused_split(hipStream_t, const float*, const void*, void*, size_t, int, const float*, int, const float*, const float*, float*, float*,
                            long long*, float, float, float*, int, int);
int pw_maxpool3x3s2_fwd_split(hipStream_t, const void*, void*, unsigned char*, int, int, int, int);

namespace {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
// Policy::debug_sync: synchronise after every enqueued stage and report the source line of the first failure (set for the
// duration of a net_* call on the calling thread; never on by default)
static thread_local int t_dbg_sync = 0;
struct DbgSyncScope { int prev; explicit DbgSyncScope(int on) : prev(t_dbg_sync) { t_dbg_sync = on; } ~DbgSyncScope() { t_dbg_sync = prev; } };
#define CK(expr) do { int _e = (expr); \
    if (t_dbg_sync) { fprintf(stderr, "[udapose] net.hip:%d %s\n", __LINE__, #expr); fflush(stderr); \
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "[udapose] FAILED at net.hip:%d\n", __LINE__); return UDAPOSE_ERR_LAUNCH; } } \
    if (_e != UDAPOSE_OK) return _e; } while (0)

struct ConvL {
    ConvGeom g;
    int w_idx = -1;         // parameter index of the weight
    int bias_idx = -1;
    size_t wf_off = 0, wb_off = 0;   // bf16 packs inside wpack (bytes)
    size_t in_off = 0;      // activation arena offset of the input (bytes)
    size_t y_off = 0;       // pre-BN output
    size_t dy_off = 0;      // workspace offset of this layer's own dy buffer (backward)
};
struct BnL {
    int C = 0;
    int g_idx = -1, b_idx = -1;             // parameter indices
    int rm_idx = -1, rv_idx = -1, nbt_idx = -1;   // buffer indices
    size_t save_off = 0;    // fp32 [3][C] saved mean / invstd / unbiased var in the arena
    size_t z_off = 0;       // post-BN(-ReLU) output
    size_t mask_off = 0;    // bn3 of a block: ReLU bit mask of z (one byte per 8 channels), what the data gradients read instead of z
    size_t npix = 0;
};
struct Block {
    ConvL c1, c2, c3, cd;
    BnL b1, b2, b3, bd;
    bool has_ds = false;
    size_t in_off = 0, zd_off = 0;
    size_t npix_in = 0;
};

struct Net {
    int layers[4], K, N, H, W;
    int split_dz_idx = -1;  // pool buffer holding the gradient that net_backward part 1 hands to part 2 (a function of the plan)
    Policy policy;          // dispatch policy of this plan (udapose_net_set_policy); every ConvGeom below points at it
    int f32 = 0;            // 1: fp32 storage + exact fp32 MFMA (forward only: the reference's teacher / validate() precision)
                            // 2: f16x2 split storage (common.h), three fp16 MFMAs per K step: the FAST fp32-grade mode (forward only);
                            //    conv inputs (z, pooled map, image) and weight packs are split tensors, pre-BN conv outputs y are fp32
    int fwd_only = 0;       // forward-only plan (mode bit 9; the teacher's no-grad forwards, validate()): nothing is kept for a backward, so the
                            // pre-BN outputs y and the post-BN outputs z of all layers rotate through SIX scratch buffers (y | block input | block
                            // output | bn1 output | bn2 output | downsample output) instead of a bump-allocated arena - layer after layer rewrites the same
                            // few MB, which stay resident in the L2s / the Infinity Cache (round 5; the arena form streams 2.8 GB per forward to HBM)
    struct Scratch { size_t need[6] = {0, 0, 0, 0, 0, 0}, off[6] = {0, 0, 0, 0, 0, 0}; bool sizing = false; } sc;     // (forward-only plans: build())
    int deconv_bias = 0;    // Upsampling(bias=True) (pose_resnet.py:15,41,96 `deconv_with_bias`): every ConvTranspose2d has a bias parameter
    size_t es = 2;          // bytes per activation element
    int n_params = 0, n_buffers = 0;
    std::vector<long long> param_numel;
    // stem
    ConvL stem; BnL stem_bn;
    size_t x8_off = 0, pool_off = 0, poolidx_off = 0;
    int Hs = 0, Ws = 0, Hp = 0, Wp = 0;
    std::vector<Block> blocks;
    ConvL up[3]; BnL up_bn[3];
    ConvL head;
    size_t head_out_off = 0;          // fp32 NHWC [N,Ho,Wo,K]
    int fc_w_idx = -1, fc_b_idx = -1;
    size_t act_bytes = 0, wpack_bytes = 0, ws_bytes = 0;
    // workspace carve (bytes)
    size_t ws_slab = 0, ws_slabf = 0, ws_coef = 0, ws_gbuf[6] = {0, 0, 0, 0, 0, 0}, ws_dyhead = 0, ws_dwtmp = 0, ws_headbwd = 0;
    size_t ws_wgpart = 0, ws_wgpart_bytes = 0;      // partial tiles of the split weight-gradient reductions (sized by net_ws_bytes from the policy of that moment)
    size_t gbuf_bytes = 0;
    int Hout = 0, Wout = 0;
    // batched weight packing: device job tables, rebuilt when the parameter / pack pointers change
    struct PackTab { void* jobs = nullptr; int* blk_job = nullptr; int* blk_sub = nullptr; int nblocks = 0; const void* key0 = nullptr;
                     const void* key1 = nullptr; const void* keyw = nullptr; };
    PackTab pack_fwd, pack_all;
    // grouped weight-gradient launch of one backward pass: device tables, rebuilt when the buffers change
    struct WgGroup {
        float k_beta = -1.f; int k_stages = 0; int k_part = 0;     // part: 0 all layers, 1 upper (head..layer3), 2 lower (layer2..stem)
        std::vector<std::pair<int, ptrdiff_t>> rel;      // (parameter index, byte offset of its gradient from grads[0]) the table assumes
        std::vector<int> rel_cls;                        // tile class of the launch that computes rel[i]'s gradient
        // tile classes of the grouped launch: 0 = 128x128, 1 = 64x64 (and the filter-row form), 2 = 256x128 (128x64 per wave)
        WgParams* d_tab[WG_CLASSES] = {}; WgGroupBlk* d_blk[WG_CLASSES] = {}; int per_xcd[WG_CLASSES] = {};
        double flops[WG_CLASSES] = {};
        std::vector<std::pair<ptrdiff_t, size_t>> zero;  // dW ranges (offset from grads[0], bytes) cleared first (split reductions, when overwriting)
        ZeroJob* d_zero = nullptr; int n_zero = 0;       // the same ranges as a device job table, when all are 16-byte granular
        // Deterministic split reductions (round 6): a layer whose pixel range is split over several work-groups (layer1 / layer2, the last
        // deconvolution, head, stem) has every split store its PARTIAL tile into the pass's workspace (ws_wgpart); one launch then adds the
        // splits of all such layers in split order into the gradient tensors (pw_split_sum): no atomics, no clears, bit-reproducible
        SumJob* d_sum = nullptr; int* d_sum_blk = nullptr; int n_sum_blk = 0;
        unsigned long long last_use = 0;
    };
    std::deque<WgGroup> wg_groups;       // (stable addresses, never evicted: a captured hipGraph may reference any table built so far)
    // fused optimizer tail (Adam + EMA + weight packs of student and teacher in one sweep): device job table
    struct UpdTab { void* jobs = nullptr; int* blk_job = nullptr; int* blk_sub = nullptr; int nblocks = 0;
                    const void* k_ps = nullptr; const void* k_pt = nullptr; const void* k_g = nullptr; const void* k_m = nullptr;
                    const void* k_ws = nullptr; const void* k_wt = nullptr; };
    UpdTab upd;
    // batched deferred running-statistics update: device job table, rebuilt when the buffer pointers change
    BnRunJob* d_runjobs = nullptr; int n_runjobs = 0; const void* runjobs_key = nullptr;
    unsigned long long wg_tick = 0;
};
struct PackJobH { const float* src; elem_t* dst; int A, T, B, kind; long long n; };

size_t act_alloc(Net& n, size_t bytes) { size_t o = n.act_bytes; n.act_bytes = align_up(o + bytes); return o; }
size_t wp_alloc(Net& n, size_t bytes) { size_t o = n.wpack_bytes; n.wpack_bytes = align_up(o + bytes); return o; }

// forward-only plans: named scratch slots, each as large as its largest tenant (sized in a dry pass of the layout)
enum { SC_Y = 0, SC_A, SC_B, SC_MID, SC_MID2, SC_DS, SC_COUNT };
size_t sc_take(Net& n, int slot, size_t bytes) {
    if (n.sc.sizing) { n.sc.need[slot] = std::max(n.sc.need[slot], align_up(bytes)); return 0; }
    return n.sc.off[slot];
}
void add_conv(Net& n, ConvL& c, int Hi, int Wi, int Ci, int Co, int K, int stride, int pad, int transposed, size_t in_off, bool need_bwd_pack, int y_slot = -1) {
    c.g = ConvGeom{n.N, Hi, Wi, Ci, Co, K, K, stride, pad, transposed, 0, 0};
    c.w_idx = n.n_params++;
    n.param_numel.push_back((long long)Co * (Ci == 8 ? 3 : Ci) * K * K);
    c.in_off = in_off;
    const size_t welems = (size_t)Co * c.g.wtaps() * Ci;
    // fp32 mode reads plain-conv weights straight from the fp32 master ([Co][taps][Ci] is its physical layout)
    if (n.f32 != 1 || c.g.smallc() || transposed) c.wf_off = wp_alloc(n, welems * n.es);
    if (need_bwd_pack && !n.f32) c.wb_off = wp_alloc(n, welems * 2);
    const size_t ybytes = (size_t)n.N * c.g.Ho() * c.g.Wo() * Co * n.es;
    c.y_off = (n.fwd_only && y_slot >= 0) ? sc_take(n, y_slot, ybytes) : act_alloc(n, ybytes);
}
void add_bn(Net& n, BnL& b, int C, size_t npix, bool alloc_z = true, int z_slot = -1) {
    b.C = C;
    b.g_idx = n.n_params++; n.param_numel.push_back(C);
    b.b_idx = n.n_params++; n.param_numel.push_back(C);
    b.rm_idx = n.n_buffers++; b.rv_idx = n.n_buffers++; b.nbt_idx = n.n_buffers++;
    b.save_off = act_alloc(n, (size_t)3 * C * 4);
    b.npix = npix;
    if (alloc_z) b.z_off = (n.fwd_only && z_slot >= 0) ? sc_take(n, z_slot, npix * C * n.es) : act_alloc(n, npix * C * n.es);
}

Net* build(const int layers[4], int K, int N, int H, int W, int mode) {
    Net* np = new Net();
    Net& n = *np;
    const int f32 = mode & 0xff;          // (mode: low byte = precision 0 / 1 / 2, bit 8 = the deconvolutions carry a bias, bit 9 = forward-only plan)
    n.deconv_bias = (mode >> 8) & 1;
    n.f32 = f32 == 2 ? 2 : (f32 ? 1 : 0);
    n.es = f32 ? 4 : 2;
    for (int i = 0; i < 4; ++i) n.layers[i] = layers[i];
    n.K = K; n.N = N; n.H = H; n.W = W;
    n.fwd_only = (mode >> 9) & 1;
    // The arena layout.  Differentiable plans: every tensor its own range (the backward reads them all).  Forward-only plans: the
    // SAME walk with the y / z tensors mapped onto six scratch slots by liveness - one live pre-BN output at a time (SC_Y), the
    // block's input and output ping-pong between SC_A and SC_B (the input is the residual: live until bn3), bn1 / bn2 outputs
    // in SC_MID / SC_MID2, the downsample branch's output waits in SC_DS for bn3.  The walk runs twice:
    // a dry pass on a copy sizes each slot by its largest tenant, the second assigns the offsets.
    auto layout = [&](Net& n) {
    const bool fo = n.fwd_only != 0;
    n.x8_off = fo ? sc_take(n, SC_DS, (size_t)N * H * W * 8 * n.es) : act_alloc(n, (size_t)N * H * W * 8 * n.es);
    // stem: conv 7x7 s2 p3 (3 -> padded 8 input channels), bn, relu, maxpool 3x3 s2 p1
    add_conv(n, n.stem, H, W, 8, 64, 7, 2, 3, 0, n.x8_off, false, SC_Y);
    n.Hs = n.stem.g.Ho(); n.Ws = n.stem.g.Wo();
    add_bn(n, n.stem_bn, 64, (size_t)N * n.Hs * n.Ws, true, SC_B);
    n.Hp = (n.Hs + 2 - 3) / 2 + 1; n.Wp = (n.Ws + 2 - 3) / 2 + 1;
    n.pool_off = fo ? sc_take(n, SC_A, (size_t)N * n.Hp * n.Wp * 64 * n.es) : act_alloc(n, (size_t)N * n.Hp * n.Wp * 64 * n.es);
    n.poolidx_off = fo ? sc_take(n, SC_MID, (size_t)N * n.Hp * n.Wp * 64) : act_alloc(n, (size_t)N * n.Hp * n.Wp * 64);      // (written, never read, by a forward-only plan)
    size_t cur = n.pool_off;
    int cur_slot = SC_A;
    int Hc = n.Hp, Wc = n.Wp, Cc = 64;
    const int planes[4] = {64, 128, 256, 512};
    for (int L = 0; L < 4; ++L)
        for (int bi = 0; bi < layers[L]; ++bi) {
            n.blocks.emplace_back();
            Block& b = n.blocks.back();
            const int P = planes[L], stride = (bi == 0 && L > 0) ? 2 : 1;
            const int out_slot = cur_slot == SC_A ? SC_B : SC_A;
            b.in_off = cur;
            b.npix_in = (size_t)N * Hc * Wc;
            b.has_ds = (bi == 0);
            // parameter order follows torchvision Bottleneck: conv1,bn1,conv2,bn2,conv3,bn3,(downsample.0, downsample.1)
            add_conv(n, b.c1, Hc, Wc, Cc, P, 1, 1, 0, 0, cur, true, SC_Y);
            add_bn(n, b.b1, P, (size_t)N * Hc * Wc, true, SC_MID);
            add_conv(n, b.c2, Hc, Wc, P, P, 3, stride, 1, 0, b.b1.z_off, true, SC_Y);
            const int Ho = b.c2.g.Ho(), Wo = b.c2.g.Wo();
            add_bn(n, b.b2, P, (size_t)N * Ho * Wo, true, SC_MID2);     // (its own slot: with eval_fold conv2 reads z1 and writes z2 in one launch)
            add_conv(n, b.c3, Ho, Wo, P, P * 4, 1, 1, 0, 0, b.b2.z_off, true, SC_Y);
            add_bn(n, b.b3, P * 4, (size_t)N * Ho * Wo, true, out_slot);
            if (!n.f32 && !fo) b.b3.mask_off = act_alloc(n, (size_t)N * Ho * Wo * (P * 4) / 8);     // (the ReLU bit mask is the backward's)
            if (b.has_ds) {
                add_conv(n, b.cd, Hc, Wc, Cc, P * 4, 1, stride, 0, 0, cur, true, SC_Y);
                add_bn(n, b.bd, P * 4, (size_t)N * Ho * Wo, true, SC_DS);
                b.zd_off = b.bd.z_off;
            }
            cur = b.b3.z_off; cur_slot = out_slot; Hc = Ho; Wc = Wo; Cc = P * 4;
        }
    n.fc_w_idx = n.n_params++; n.param_numel.push_back(1000LL * 2048);
    n.fc_b_idx = n.n_params++; n.param_numel.push_back(1000);
    for (int i = 0; i < 3; ++i) {
        add_conv(n, n.up[i], Hc, Wc, Cc, 256, 4, 2, 1, 1, cur, true, SC_Y);
        if (n.deconv_bias) { n.up[i].bias_idx = n.n_params++; n.param_numel.push_back(256); }      // (.parameters() order: weight, bias, then the BN)
        Hc = n.up[i].g.Ho(); Wc = n.up[i].g.Wo(); Cc = 256;
        cur_slot = cur_slot == SC_A ? SC_B : SC_A;
        add_bn(n, n.up_bn[i], 256, (size_t)N * Hc * Wc, true, cur_slot);
        cur = n.up_bn[i].z_off;
    }
    // head: 1x1 conv with bias -> fp32
    n.head.g = ConvGeom{N, Hc, Wc, 256, K, 1, 1, 1, 0, 0, 0, 0};
    n.head.w_idx = n.n_params++; n.param_numel.push_back((long long)K * 256);
    n.head.bias_idx = n.n_params++; n.param_numel.push_back(K);
    n.head.in_off = cur;
    n.head.wf_off = wp_alloc(n, (size_t)K * 256 * n.es);
    n.head.wb_off = wp_alloc(n, (size_t)256 * 64 * 2);      // [256][1][64] zero-padded for dgrad
    n.head_out_off = fo ? sc_take(n, SC_Y, (size_t)N * Hc * Wc * K * 4) : act_alloc(n, (size_t)N * Hc * Wc * K * 4);
    n.Hout = Hc; n.Wout = Wc;
    return Hc * 65536 + Wc;
    };
    if (n.fwd_only) {
        Net dry = n;
        dry.sc.sizing = true;
        (void)layout(dry);
        for (int k = 0; k < SC_COUNT; ++k) { n.sc.need[k] = dry.sc.need[k]; n.sc.off[k] = act_alloc(n, dry.sc.need[k]); }
    }
    const int hw_out = layout(n);
    const int Hc = hw_out >> 16, Wc = hw_out & 0xffff;

    // transient workspace: BN stat slabs, coefficient vectors, gradient ping-pong buffers
    size_t max_slab = 0, max_act = 0;
    auto upd = [&](const ConvL& c) {
        const size_t rows = (size_t)conv_stat_rows(c.g);
        max_slab = std::max(max_slab, rows * 2 * c.g.Co * 4);
        max_act = std::max(max_act, (size_t)n.N * c.g.Ho() * c.g.Wo() * c.g.Co * 2);
        max_act = std::max(max_act, (size_t)n.N * c.g.Hi * c.g.Wi * c.g.Ci * 2);
    };
    upd(n.stem);
    for (auto& b : n.blocks) { upd(b.c1); upd(b.c2); upd(b.c3); if (b.has_ds) upd(b.cd); }
    for (int i = 0; i < 3; ++i) upd(n.up[i]);
    max_slab = std::max(max_slab, (size_t)1024 * 2 * 2048 * 4);   // bn backward partials: <=1024 rows x 2 x C
    size_t o = 0;
    n.ws_slab = o; o = align_up(o + max_slab);
    n.ws_slabf = o; o = align_up(o + max_slab);     // partial sums written by dgrad epilogues (the downsample BN keeps ws_slab)
    n.ws_coef = o; o = align_up(o + (size_t)3 * 2048 * 4 + 2 * 2048 * 4);
    if (n.fwd_only) {           // (no backward: statistics slabs and coefficient vectors only)
        n.ws_bytes = o;
        n.stem.g.pol = &n.policy;
        for (auto& b : n.blocks) { b.c1.g.pol = b.c2.g.pol = b.c3.g.pol = b.cd.g.pol = &n.policy; }
        for (int i = 0; i < 3; ++i) n.up[i].g.pol = &n.policy;
        n.head.g.pol = &n.policy;
        return np;
    }
    n.gbuf_bytes = align_up(2 * max_act);   // x2: the deconv-stage gradients are fp32
    for (int i = 0; i < 6; ++i) { n.ws_gbuf[i] = o; o += n.gbuf_bytes; }
    n.ws_dyhead = o; o = align_up(o + (size_t)N * Hc * Wc * 64 * 2);
    n.ws_dwtmp = o; o = align_up(o + std::max((size_t)64 * 56 * 8 * 4, (size_t)64 * 256 * 4));
    // every conv layer owns its dy buffer: the weight gradients of the whole pass are computed by ONE grouped launch after
    // the dgrad / BN-backward chain, so every dy must still be there
    auto dyb = [&](ConvL& c) { c.dy_off = o; o = align_up(o + (size_t)n.N * c.g.Ho() * c.g.Wo() * c.g.Co * 2); };
    dyb(n.stem);
    for (auto& b : n.blocks) { dyb(b.c1); dyb(b.c2); dyb(b.c3); if (b.has_ds) dyb(b.cd); }
    for (int i = 0; i < 3; ++i) dyb(n.up[i]);
    n.ws_wgpart = o;          // (last: its size follows the policy's split length, net_ws_bytes)
    n.ws_bytes = o;
    // every convolution of the plan dispatches with the plan's policy
    n.stem.g.pol = &n.policy;
    for (auto& b : n.blocks) { b.c1.g.pol = b.c2.g.pol = b.c3.g.pol = b.cd.g.pol = &n.policy; }
    for (int i = 0; i < 3; ++i) n.up[i].g.pol = &n.policy;
    n.head.g.pol = &n.policy;
    return np;
}

struct Pool {
    char* base; size_t off[6]; bool used[6] = {false, false, false, false, false, false};
    elem_t* get() { for (int i = 0; i < 6; ++i) if (!used[i]) { used[i] = true; return (elem_t*)(base + off[i]); } return nullptr; }
    void put(const void* p) { for (int i = 0; i < 6; ++i) if ((char*)p == base + off[i]) used[i] = false; }
};

int pack_conv(hipStream_t s, const Net& n, const ConvL& c, const void* const* params, char* wpack, bool with_bwd) {
    const float* w = (const float*)params[c.w_idx];
    const ConvGeom& g = c.g;
    elem_t* wf = (elem_t*)(wpack + c.wf_off);
    elem_t* wb = (elem_t*)(wpack + c.wb_off);
    const int T = g.KH * g.KW;
    if (n.f32 == 2) {
        // f16x2 packs: the fp32 GEMM layouts, every group of 8 values split into [8 h][8 l]
        if (g.smallc())
            return pw_pack_strided_split(s, w, wpack + c.wf_off, g.Co, g.KH, g.KWp(), g.KW, 8, 3, (long)g.KH * g.KW * 3, (long)g.KW * 3, 3, 1);
        if (g.transposed) return pw_transpose_split(s, w, wpack + c.wf_off, g.Ci, T, g.Co);
        return pw_f32_to_split(s, w, wpack + c.wf_off, (size_t)g.Co * T * g.Ci);
    }
    if (n.f32) {
        if (g.smallc())
            return pw_pack_strided_f32(s, w, (float*)(wpack + c.wf_off), g.Co, g.KH, g.KWp(), g.KW, 8, 3, (long)g.KH * g.KW * 3, (long)g.KW * 3, 3, 1);
        if (g.transposed) return pw_transpose_f32(s, w, (float*)(wpack + c.wf_off), g.Ci, T, g.Co);
        return UDAPOSE_OK;
    }
    if (g.smallc()) {
        // master: [Co][KH][KW][3] (channels_last of [Co,3,KH,KW]) -> [Co][KH][KWp][8]
        return pw_pack_strided(s, w, wf, g.Co, g.KH, g.KWp(), g.KW, 8, 3, (long)g.KH * g.KW * 3, (long)g.KW * 3, 3, 1);
    }
    if (!g.transposed) {
        // master physical [Co][T][Ci]: fprop pack is a cast, dgrad pack [Ci][T][Co] a per-tap transpose
        CK(pw_cast_f32_bf16(s, w, wf, (size_t)g.Co * T * g.Ci));
        if (with_bwd) CK(pw_transpose_cast(s, w, wb, g.Co, T, g.Ci));
    } else {
        // ConvTranspose2d master physical [Ci][T][Co] (channels_last of [Ci,Co,KH,KW]): dgrad pack is the cast
        CK(pw_transpose_cast(s, w, wf, g.Ci, T, g.Co));
        if (with_bwd) CK(pw_cast_f32_bf16(s, w, wb, (size_t)g.Ci * T * g.Co));
    }
    return UDAPOSE_OK;
}

int conv_bn_fwd(hipStream_t s, const Net& n, const ConvL& c, const BnL& b, const void* const* params, void* const* buffers, const char* wpack,
                char* act, char* ws, int training, float momentum, const elem_t* res, int relu, bool upd, bool no_apply = false) {
    ConvEpilogue e;
    float* slab = (float*)(ws + n.ws_slab);
    float* scale = (float*)(ws + n.ws_coef);
    float* shift = scale + 2048;
    float* save = (float*)(act + b.save_off);
    e.stats = training ? slab : nullptr;
    e.f32 = n.f32 != 0;
    e.split = n.f32 == 2;
    e.out_f32 = n.f32 == 2;       // (f16x2: the pre-BN output stays fp32; the BN apply writes the split z)
    // a bias in front of the BatchNorm (deconv_with_bias): added to the stored output; the statistics epilogue sees the raw
    // accumulators, so the finalize launch shifts the mean by it
    const float* pre_bias = c.bias_idx >= 0 ? (const float*)params[c.bias_idx] : nullptr;
    e.bias = pre_bias;
    // block outputs: the apply also saves the ReLU bit mask of z (1/16 of z's bytes) for the data gradients that mask with it
    unsigned char* mask = (b.mask_off && n.policy.bn3_mask && relu) ? (unsigned char*)(act + b.mask_off) : nullptr;
    const void* wptr = (n.f32 == 1 && !c.g.smallc() && !c.g.transposed) ? params[c.w_idx] : (const void*)(wpack + c.wf_off);
    const float* gamma = (const float*)params[b.g_idx];
    const float* beta = (const float*)params[b.b_idx];
    if (!training && !no_apply && !pre_bias && n.policy.eval_fold) {
        // Eval mode (validate(), train_human.py:461-500): the BatchNorm is a per-channel affine map known BEFORE the convolution, so the
        // conv's epilogue applies it from the fp32 accumulators - z = relu(conv * scale + shift (+ residual)) is written by the conv
        // itself: no apply launch, no pre-BN tensor, one rounding less than conv -> y -> apply
        CK(pw_bn_eval_coeff(s, b.C, gamma, beta, (const float*)buffers[b.rm_idx], (const float*)buffers[b.rv_idx], 1e-5f, scale, shift));
        e.stats = nullptr;
        e.scale = scale; e.bias = shift; e.relu = relu;
        e.res = res;
        e.out_f32 = 0;                    // (z has the plan's storage type: 16-bit, fp32, or the f16x2 split form)
        return conv_fprop(s, c.g, (const elem_t*)(act + c.in_off), (const elem_t*)wptr, act + b.z_off, e);
    }
    CK(conv_fprop(s, c.g, (const elem_t*)(act + c.in_off), (const elem_t*)wptr, act + c.y_off, e));
    if (training && !n.f32 && !no_apply && !pre_bias) {
        // wide, small-spatial layers: finalize + apply in ONE launch (channel-chunked work-groups, pointwise.hip)
        const int took = pw_bn_train_fused(s, (const elem_t*)(act + c.y_off), res, (elem_t*)(act + b.z_off), b.npix, b.C, slab, conv_stat_rows(c.g), gamma,
                                           beta, upd ? (float*)buffers[b.rm_idx] : nullptr, upd ? (float*)buffers[b.rv_idx] : nullptr,
                                           upd ? (long long*)buffers[b.nbt_idx] : nullptr, momentum, 1e-5f, save, relu, n.policy.bn_fwd_chunked | (n.policy.bn_xcd_rows ? (1 << 30) : 0), mask);
        if (took < 0) return took;
        if (took) return UDAPOSE_OK;
    }
    if (training && n.f32 == 2 && !no_apply && !pre_bias) {
        const int took = pw_bn_train_fused_split(s, (const float*)(act + c.y_off), res, act + b.z_off, b.npix, b.C, slab, conv_stat_rows(c.g), gamma, beta,
                                                 upd ? (float*)buffers[b.rm_idx] : nullptr, upd ? (float*)buffers[b.rv_idx] : nullptr,
                                                 upd ? (long long*)buffers[b.nbt_idx] : nullptr, momentum, 1e-5f, save, relu, n.policy.bn_fwd_chunked | (n.policy.bn_xcd_rows ? (1 << 30) : 0));
        if (took < 0) return took;
        if (took) return UDAPOSE_OK;
    }
    if (training)
        CK(pw_bn_finalize(s, slab, conv_stat_rows(c.g), b.C, (double)b.npix, gamma, beta, upd ? (float*)buffers[b.rm_idx] : nullptr,
                          upd ? (float*)buffers[b.rv_idx] : nullptr, upd ? (long long*)buffers[b.nbt_idx] : nullptr, momentum, 1e-5f, scale, shift,
                          save, save + b.C, pre_bias));
    else
        CK(pw_bn_eval_coeff(s, b.C, gamma, beta, (const float*)buffers[b.rm_idx], (const float*)buffers[b.rv_idx], 1e-5f, scale, shift));
    if (no_apply) return UDAPOSE_OK;      // (the caller's next launch applies scale / shift itself: the stem's fused pool)
    if (n.f32 == 2)
        return pw_bn_apply_split(s, (const float*)(act + c.y_off), res, act + b.z_off, b.npix * b.C, b.C, scale, shift, relu);
    if (n.f32)
        return pw_bn_apply_f32(s, (const float*)(act + c.y_off), (const float*)res, (float*)(act + b.z_off), b.npix * b.C, b.C, scale, shift, relu);
    return pw_bn_apply(s, (const elem_t*)(act + c.y_off), res, (elem_t*)(act + b.z_off), b.npix * b.C, b.C, scale, shift, relu, mask, n.policy.bn_xcd_rows >= 2);
}

}  // namespace

// ============================================================================ public (C++) entry points
void* net_create(const int layers[4], int K, int N, int H, int W, int mode) {
    if (K < 1 || K > 64 || N < 1 || H % 32 || W % 32 || (mode & 0xff) > 2 || (mode >> 10)) return nullptr;
    return build(layers, K, N, H, W, mode);
}
void net_destroy(void* h) {
    Net* n = (Net*)h;
    if (!n) return;
    if (n->d_runjobs) (void)hipFree(n->d_runjobs);
    if (n->upd.jobs) { (void)hipFree(n->upd.jobs); (void)hipFree(n->upd.blk_job); (void)hipFree(n->upd.blk_sub); }
    for (auto& g : n->wg_groups)
        for (int t = 0; t < WG_CLASSES; ++t) { if (g.d_tab[t]) (void)hipFree(g.d_tab[t]); if (g.d_blk[t]) (void)hipFree(g.d_blk[t]); }
    for (auto& g : n->wg_groups) { if (g.d_zero) (void)hipFree(g.d_zero); if (g.d_sum) (void)hipFree(g.d_sum); if (g.d_sum_blk) (void)hipFree(g.d_sum_blk); }
    delete n;
}
void net_set_policy(void* h, const Policy& p) { ((Net*)h)->policy = p; }
const Policy& net_get_policy(void* h) { return ((Net*)h)->policy; }
int net_num_params(void* h) { return ((Net*)h)->n_params; }
int net_num_buffers(void* h) { return ((Net*)h)->n_buffers; }
long long net_param_numel(void* h, int i) { return ((Net*)h)->param_numel[i]; }
size_t net_wpack_bytes(void* h) { return ((Net*)h)->wpack_bytes; }
size_t net_act_bytes(void* h) { return ((Net*)h)->act_bytes; }
namespace { size_t wg_partial_bytes(Net& n); }
size_t net_ws_bytes(void* h) {
    // the partial tiles of the split weight-gradient reductions (policy wgrad_det) close the workspace: their size follows the policy's
    // split length, which udapose_net_set_policy may have changed since the plan was created - sized here, when the caller asks
    Net& n = *(Net*)h;
    if (n.f32 || n.fwd_only || !n.policy.wgrad_group || !n.policy.wgrad_det) { n.ws_wgpart_bytes = 0; return n.ws_bytes; }
    n.ws_wgpart_bytes = wg_partial_bytes(n);
    return n.ws_bytes + n.ws_wgpart_bytes;
}
void net_out_shape(void* h, int* shp) { Net& n = *(Net*)h; shp[0] = n.N; shp[1] = n.K; shp[2] = n.Hout; shp[3] = n.Wout; }

namespace {
void add_pack_jobs(const Net& n, const ConvL& c, const void* const* params, char* wpack, bool with_bwd, std::vector<PackJobH>& jobs) {
    const ConvGeom& g = c.g;
    if (g.smallc()) return;                       // stem: strided gather, launched separately
    const float* w = (const float*)params[c.w_idx];
    const int T = g.KH * g.KW;
    elem_t* wf = (elem_t*)(wpack + c.wf_off);
    elem_t* wb = (elem_t*)(wpack + c.wb_off);
    const int sp = n.f32 == 2 ? 2 : 0;          // (kind bit 1: f16x2 split output, forward packs only)
    if (sp) with_bwd = false;
    if (!g.transposed) {
        jobs.push_back(PackJobH{w, wf, 0, 0, 0, sp, (long long)g.Co * T * g.Ci});
        if (with_bwd) jobs.push_back(PackJobH{w, wb, g.Co, T, g.Ci, 1, 0});
    } else {
        jobs.push_back(PackJobH{w, wf, g.Ci, T, g.Co, 1 | sp, 0});
        if (with_bwd) jobs.push_back(PackJobH{w, wb, 0, 0, 0, 0, (long long)g.Ci * T * g.Co});
    }
}
int build_pack_table(Net& n, Net::PackTab& tab, const void* const* params, char* wpack, bool with_bwd) {
    std::vector<PackJobH> jobs;
    for (auto& b : n.blocks) {
        add_pack_jobs(n, b.c1, params, wpack, with_bwd, jobs);
        add_pack_jobs(n, b.c2, params, wpack, with_bwd, jobs);
        add_pack_jobs(n, b.c3, params, wpack, with_bwd, jobs);
        if (b.has_ds) add_pack_jobs(n, b.cd, params, wpack, with_bwd, jobs);
    }
    for (int i = 0; i < 3; ++i) add_pack_jobs(n, n.up[i], params, wpack, with_bwd, jobs);
    jobs.push_back(PackJobH{(const float*)params[n.head.w_idx], (elem_t*)(wpack + n.head.wf_off), 0, 0, 0, n.f32 == 2 ? 2 : 0, (long long)n.K * 256});
    std::vector<int> bj, bs;
    for (size_t j = 0; j < jobs.size(); ++j) {
        const PackJobH& q = jobs[j];
        const long nb = (q.kind & 1) == 0 ? (long)((q.n + 8191) / 8192) : (long)((q.A + 31) / 32) * ((q.B + 31) / 32) * q.T;
        for (long k = 0; k < nb; ++k) { bj.push_back((int)j); bs.push_back((int)k); }
    }
    if (tab.jobs) { (void)hipFree(tab.jobs); (void)hipFree(tab.blk_job); (void)hipFree(tab.blk_sub); }
    if (hipMalloc(&tab.jobs, jobs.size() * sizeof(PackJobH)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMalloc((void**)&tab.blk_job, bj.size() * sizeof(int)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMalloc((void**)&tab.blk_sub, bs.size() * sizeof(int)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMemcpy(tab.jobs, jobs.data(), jobs.size() * sizeof(PackJobH), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMemcpy(tab.blk_job, bj.data(), bj.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMemcpy(tab.blk_sub, bs.data(), bs.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    tab.nblocks = (int)bj.size();
    tab.key0 = params[0]; tab.key1 = params[n.n_params - 1]; tab.keyw = wpack;
    return UDAPOSE_OK;
}
int build_run_jobs(Net& n, void* const* buffers) {
    std::vector<BnRunJob> jobs;
    auto one = [&](const BnL& b) {
        jobs.push_back(BnRunJob{b.save_off, (float*)buffers[b.rm_idx], (float*)buffers[b.rv_idx], (long long*)buffers[b.nbt_idx], b.C, 0});
    };
    one(n.stem_bn);
    for (auto& b : n.blocks) {
        one(b.b1); one(b.b2); one(b.b3);
        if (b.has_ds) one(b.bd);
    }
    for (int i = 0; i < 3; ++i) one(n.up_bn[i]);
    if (n.d_runjobs) { (void)hipFree(n.d_runjobs); n.d_runjobs = nullptr; }
    if (hipMalloc((void**)&n.d_runjobs, jobs.size() * sizeof(BnRunJob)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMemcpy(n.d_runjobs, jobs.data(), jobs.size() * sizeof(BnRunJob), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    n.n_runjobs = (int)jobs.size();
    n.runjobs_key = buffers[0];
    return UDAPOSE_OK;
}
}  // namespace

// Builds every device table the plan's calls need for THESE parameter / buffer / pack pointers (pack job tables, the
// running-statistics job table, the tap plans of all layer geometries): the only place where net_pack_weights, net_forward and
// net_apply_running's tables are allocated and uploaded (synchronously: call it outside stream capture, again whenever a
// pointer changes).  Those calls return UDAPOSE_ERR_NOT_PREPARED when their table does not match the pointers they are given.
int net_bind(void* h, const void* const* params, void* const* buffers, void* wpack_) {
    Net& n = *(Net*)h;
    char* wpack = (char*)wpack_;
    CK(conv_prepare(n.stem.g));
    for (auto& b : n.blocks) {
        CK(conv_prepare(b.c1.g)); CK(conv_prepare(b.c2.g)); CK(conv_prepare(b.c3.g));
        if (b.has_ds) CK(conv_prepare(b.cd.g));
    }
    for (int i = 0; i < 3; ++i) CK(conv_prepare(n.up[i].g));
    { ConvGeom hg = n.head.g; CK(conv_prepare(hg)); hg.Co = 64; CK(conv_prepare(hg)); }
    if (n.f32 != 1) {
        CK(build_pack_table(n, n.pack_fwd, params, wpack, false));
        CK(build_pack_table(n, n.pack_all, params, wpack, n.f32 == 0));
    }
    if (buffers) CK(build_run_jobs(n, buffers));
    return UDAPOSE_OK;
}

int net_pack_weights(void* h, hipStream_t s, const void* const* params, void* wpack_, int with_bwd) {
    Net& n = *(Net*)h;
    DbgSyncScope dbg(n.policy.debug_sync);
    char* wpack = (char*)wpack_;
    if (n.f32 != 1) {
        // ONE launch casts / transposes every weight through the job table net_bind built for these pointers
        if (n.f32 == 2) with_bwd = 0;
        Net::PackTab& tab = with_bwd ? n.pack_all : n.pack_fwd;
        if (!tab.jobs || tab.key0 != params[0] || tab.key1 != params[n.n_params - 1] || tab.keyw != (const void*)wpack)
            return UDAPOSE_ERR_NOT_PREPARED;
        CK(pack_conv(s, n, n.stem, params, wpack, false));
        CK(pw_pack_multi(s, tab.jobs, tab.blk_job, tab.blk_sub, tab.nblocks));
        if (with_bwd)   // head dgrad pack [256][1][64]: wb[ci][k] = w[k][ci], zero for k >= K
            CK(pw_pack_strided(s, (const float*)params[n.head.w_idx], (elem_t*)(wpack + n.head.wb_off), 256, 1, 1, 1, 64, n.K, 1, 0, 0, 256));
        return UDAPOSE_OK;
    }
    CK(pack_conv(s, n, n.stem, params, wpack, false));
    for (auto& b : n.blocks) {
        CK(pack_conv(s, n, b.c1, params, wpack, with_bwd));
        CK(pack_conv(s, n, b.c2, params, wpack, with_bwd));
        CK(pack_conv(s, n, b.c3, params, wpack, with_bwd));
        if (b.has_ds) CK(pack_conv(s, n, b.cd, params, wpack, with_bwd));
    }
    for (int i = 0; i < 3; ++i) CK(pack_conv(s, n, n.up[i], params, wpack, with_bwd));
    const float* hw = (const float*)params[n.head.w_idx];
    if (n.f32) return UDAPOSE_OK;      // the head reads its fp32 master directly
    CK(pw_cast_f32_bf16(s, hw, (elem_t*)(wpack + n.head.wf_off), (size_t)n.K * 256));
    if (with_bwd)   // [256][1][64]: wb[ci][k] = w[k][ci], zero for k >= K
        CK(pw_pack_strided(s, hw, (elem_t*)(wpack + n.head.wb_off), 256, 1, 1, 1, 64, n.K, 1, 0, 0, 256));
    return UDAPOSE_OK;
}

int net_forward(void* h, hipStream_t s, const float* x_nchw, const void* const* params, void* const* buffers, const void* wpack_, void* act_, void* ws_,
                float* out_nchw, int training, float momentum) {
    const Net& n = *(const Net*)h;
    DbgSyncScope dbg(n.policy.debug_sync);
    const bool upd = !(training & 2);               // bit 1 of `training`: defer the running-statistics update
    training &= 1;
    const char* wpack = (const char*)wpack_;
    char* act = (char*)act_;
    char* ws = (char*)ws_;
    if (n.f32 == 2) CK(pw_nchw_f32_to_nhwc_split(s, x_nchw, act + n.x8_off, n.N, 3, n.H * n.W, 8));
    else if (n.f32) CK(pw_nchw_f32_to_nhwc_f32(s, x_nchw, (float*)(act + n.x8_off), n.N, 3, n.H * n.W, 8));
    else CK(pw_nchw_f32_to_nhwc_bf16(s, x_nchw, (elem_t*)(act + n.x8_off), n.N, 3, n.H * n.W, 8));
    const bool stem_fused = n.policy.stem_fused && !n.f32;
    CK(conv_bn_fwd(s, n, n.stem, n.stem_bn, params, buffers, wpack, act, ws, training, momentum, nullptr, 1, upd, stem_fused));
    if (stem_fused)     // BN apply + ReLU + max-pool in one sweep; z of the stem is never materialised (the backward masks from y)
        CK(pw_bn_relu_maxpool3x3s2(s, (const elem_t*)(act + n.stem.y_off), (elem_t*)(act + n.pool_off), (unsigned char*)(act + n.poolidx_off), n.N,
                                   n.Hs, n.Ws, 64, (const float*)(ws + n.ws_coef), (const float*)(ws + n.ws_coef) + 2048));
    else if (n.f32 == 2)
        CK(pw_maxpool3x3s2_fwd_split(s, act + n.stem_bn.z_off, act + n.pool_off, (unsigned char*)(act + n.poolidx_off), n.N, n.Hs, n.Ws, 64));
    else if (n.f32)
        CK(pw_maxpool3x3s2_fwd_f32(s, (const float*)(act + n.stem_bn.z_off), (float*)(act + n.pool_off), (unsigned char*)(act + n.poolidx_off), n.N,
                                   n.Hs, n.Ws, 64));
    else
        CK(pw_maxpool3x3s2_fwd(s, (const elem_t*)(act + n.stem_bn.z_off), (elem_t*)(act + n.pool_off), (unsigned char*)(act + n.poolidx_off), n.N,
                               n.Hs, n.Ws, 64));
    for (auto& b : n.blocks) {
        CK(conv_bn_fwd(s, n, b.c1, b.b1, params, buffers, wpack, act, ws, training, momentum, nullptr, 1, upd));
        CK(conv_bn_fwd(s, n, b.c2, b.b2, params, buffers, wpack, act, ws, training, momentum, nullptr, 1, upd));
        const elem_t* res = (const elem_t*)(act + b.in_off);
        if (b.has_ds) {
            CK(conv_bn_fwd(s, n, b.cd, b.bd, params, buffers, wpack, act, ws, training, momentum, nullptr, 0, upd));
            res = (const elem_t*)(act + b.zd_off);
        }
        CK(conv_bn_fwd(s, n, b.c3, b.b3, params, buffers, wpack, act, ws, training, momentum, res, 1, upd));
    }
    for (int i = 0; i < 3; ++i) CK(conv_bn_fwd(s, n, n.up[i], n.up_bn[i], params, buffers, wpack, act, ws, training, momentum, nullptr, 1, upd));
    ConvEpilogue e;
    e.bias = (const float*)params[n.head.bias_idx];
    e.out_f32 = 1;
    e.f32 = n.f32 != 0;
    e.split = n.f32 == 2;
    const void* hwp = n.f32 == 1 ? params[n.head.w_idx] : (const void*)(wpack + n.head.wf_off);
    CK(conv_fprop(s, n.head.g, (const elem_t*)(act + n.head.in_off), (const elem_t*)hwp, act + n.head_out_off, e));
    return pw_nhwc_to_nchw_f32(s, act + n.head_out_off, 1, out_nchw, n.N, n.K, n.Hout * n.Wout, n.K, nullptr, nullptr);
}

namespace {
// backward of conv+bn(+relu): dz (grad wrt z) -> parameter grads, returns dx of the conv input in a pool buffer
// relu: 0 none, 1 mask from the saved z (bn3: z includes the residual), 2 mask recomputed from y (z is not read)
// pre:  the dgrad that produced dz already applied this BN's mask and left the partial sums in pre->slab (dz is g)
// next: the BN that consumes dx; this layer's dgrad masks dx for it and reduces its sums (filled by bn_stat_of)
int conv_bn_bwd(hipStream_t s, const Net& n, const ConvL& c, const BnL& b, const void* const* params, const char* wpack, char* act, char* ws,
                void* const* grads, float beta, Pool& pool, const void* dz, int dz_f32, elem_t* gout, int relu, const elem_t* dx_res,
                elem_t** dx_out, bool need_dx, int dx_f32, bool grouped_wgrad, const DgradBnStat* pre = nullptr, DgradBnStat* next = nullptr) {
    float* slab = (float*)(ws + n.ws_slab);
    float* coef = (float*)(ws + n.ws_coef) + 4096;
    const float* save = (const float*)(act + b.save_off);
    elem_t* dy = (elem_t*)(ws + c.dy_off);
    if (pre)
        CK(pw_bn_bwd_pre(s, dz, dz_f32, (const elem_t*)(act + c.y_off), dy, b.npix, b.C, (const float*)params[b.g_idx], save, save + b.C, pre->slab,
                         pre->rows, coef, (float*)grads[b.g_idx], (float*)grads[b.b_idx], beta, n.policy.bn_bwd_chunked | (n.policy.bn_xcd_rows ? (1 << 30) : 0) | (n.policy.bn_xcd_rows >= 2 ? (1 << 29) : 0),
                         n.policy.bn_bwd_pre_legacy));
    else
        CK(pw_bn_bwd(s, dz, dz_f32, (const elem_t*)(act + b.z_off), (const elem_t*)(act + c.y_off), dy, gout, b.npix, b.C, (const float*)params[b.g_idx],
                     save, save + b.C, relu, slab, coef, (float*)grads[b.g_idx], (float*)grads[b.b_idx], beta, (const float*)params[b.b_idx],
                     n.policy.bn_bwd_chunked));
    const elem_t* xin = (const elem_t*)(act + c.in_off);
    if (c.bias_idx >= 0 && beta == 0.f) {
        // bias in front of a training-mode BatchNorm: its gradient is the pixel sum of dy = gamma * invstd * (g - mean(g) - xhat * mean(g xhat)),
        // which is zero identically (autograd returns rounding noise around 0 there)
        if (pw_zero(s, grads[c.bias_idx], (size_t)c.g.Co * sizeof(float)) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
    }
    if (c.g.smallc() && grouped_wgrad && n.policy.wgrad_group_stem) {
        // (the stem's weight gradient joins the grouped launch in its row-tap form, run_wg_group)
    } else if (c.g.smallc()) {
        float* tmp = (float*)(ws + n.ws_dwtmp);
        CK(conv_wgrad(s, c.g, dy, xin, tmp, 0, -1));
        CK(pw_unpack_strided(s, tmp, (float*)grads[c.w_idx], c.g.Co, c.g.KH, c.g.KWp(), c.g.KW, 8, 3, (long)c.g.KH * c.g.KW * 3, (long)c.g.KW * 3, 3, 1,
                             beta));
    } else if (!grouped_wgrad) {
        CK(conv_wgrad(s, c.g, dy, xin, (float*)grads[c.w_idx], beta != 0.f, -1));
    }
    if (need_dx) {
        elem_t* dx = pool.get();
        if (!dx) return UDAPOSE_ERR_ARG;
        CK(conv_dgrad(s, c.g, dy, (const elem_t*)(wpack + c.wb_off), dx, dx_res, dx_f32, next));
        *dx_out = dx;
    }
    return UDAPOSE_OK;
}

// the consumer-BN description a dgrad epilogue needs: relu 1 = mask from the saved z, 2 = mask recomputed from y
DgradBnStat bn_stat_of(const Net& n, const ConvL& c, const BnL& b, const void* const* params, char* act, char* ws, int relu) {
    DgradBnStat st;
    const float* save = (const float*)(act + b.save_off);
    st.y = (const elem_t*)(act + c.y_off);
    st.z = relu == 1 ? (const elem_t*)(act + b.z_off) : nullptr;
    st.mask = (relu == 1 && b.mask_off && n.policy.bn3_mask) ? (const unsigned char*)(act + b.mask_off) : nullptr;
    st.mean = save; st.invstd = save + b.C;
    st.gamma = (const float*)params[b.g_idx]; st.beta = (const float*)params[b.b_idx];
    st.slab = (float*)(ws + n.ws_slabf);
    return st;
}

// ---- grouped weight gradients ------------------------------------------------------------------------------------------
// The weight gradient of a layer needs only that layer's dy and saved input, and nobody reads it before the optimizer.
// Launched layer by layer, most of these GEMMs are too small for the chip (a layer3 1x1: 4 GFLOP, 16 tiles of 128x128) and
// had to split their pixel reduction 8-64 ways to fill it (fp32 atomics into a zeroed dW: ~1.3 TB/s chip-wide).  One
// launch over ALL layers has thousands of tiles: most layers reduce all their pixels inside one work-group (plain stores,
// deterministic, no memset), take 128x128 tiles, and only the large-image layers are still split.

// first block of layer3: the backward of a data-parallel step is cut there (net_backward part 1 / part 2), because the
// gradients of everything above it - 94 % of the parameters - are complete at that point and their all-reduce can run under
// the rest of the backward
inline int split_block(const Net& n) { return n.layers[0] + n.layers[1]; }

// which layers a `part` of the grouped weight gradients covers: 0 all, 1 / 2 the data-parallel cut
struct PartSel { bool head_up; int hi, lo; bool stem; };
PartSel part_sel(const Net& n, int part) {
    const int nb = (int)n.blocks.size(), split = split_block(n);
    if (part == 1) return PartSel{true, nb - 1, split, false};
    if (part == 2) return PartSel{false, split - 1, 0, true};
    return PartSel{true, nb - 1, 0, true};
}

// (sizing = true: a dry walk that only adds up the bytes of partial tiles the part's split reductions need, into *need)
int build_wg_group(Net& n, Net::WgGroup& G, void* const* grads, float beta, int part, bool sizing = false, size_t* need = nullptr) {
    const PartSel sel = part_sel(n, part);
    const bool upper = sel.head_up, lower = sel.stem;
    std::vector<WgParams> tab[WG_CLASSES];
    struct Unit { int prob, z, nblk; long load; };
    std::vector<Unit> units[WG_CLASSES];
    G.zero.clear();
    G.rel.clear();
    G.rel_cls.clear();
    for (int t = 0; t < WG_CLASSES; ++t) G.flops[t] = 0.0;
    // Split reductions: every non-empty split z of a layer stores its partial tile at ws_wgpart + part_cur + z * span (plain stores,
    // WG_FLAG_DW_WS addressing); SumJob (part, dst, span, splits) lets pw_split_sum add them in split order into the gradient tensor
    // (dst = beta * dst + sum) - or, for the stem's row-tap form, into its padded scratch, from which the unpack launch goes on.
    std::vector<SumJob> sums;
    size_t part_cur = 0;
    const bool det = n.policy.wgrad_det && n.policy.wgrad_group;      // (the same condition sizes the workspace: net_ws_bytes)
    auto make_partial = [&](WgParams& p, long long dst_off, int dst_ws, int Cdim) -> int {
        const int ms_total = (p.M + 63) / 64, per = (ms_total + p.ksplit - 1) / p.ksplit;
        const int ks_eff = (ms_total + per - 1) / per;                      // splits that own at least one stage (the others have no unit)
        const size_t span = (size_t)p.rows_valid * p.wtaps * Cdim;
        if (span >= (1ull << 31) || (span & 3)) return UDAPOSE_ERR_UNSUPPORTED;
        p.flags = (p.flags & ~WG_FLAG_ATOMIC) | WG_FLAG_DW_WS;
        p.dw = (float*)(n.ws_wgpart + part_cur);
        p.part_stride = (unsigned)span;
        sums.push_back(SumJob{(long long)(n.ws_wgpart + part_cur), dst_off, (unsigned)span, (unsigned)span, ks_eff, dst_ws, dst_ws ? 0.f : beta, 0});
        part_cur = align_up(part_cur + (size_t)ks_eff * span * sizeof(float));
        return UDAPOSE_OK;
    };
    auto add_geom = [&](const ConvGeom& g, int w_idx, size_t dy_off, size_t in_off, int rows_valid) -> int {
        if (g.smallc()) return UDAPOSE_OK;
        WgParams p;
        double fl = 0.0;
        // byte offsets from the workspace / activation arena / gradient bases in place of pointers (see wgrad_dma_group_kernel)
        const ptrdiff_t drel = sizing ? 0 : (const char*)grads[w_idx] - (const char*)grads[0];
        G.rel.push_back({w_idx, drel});
        CK(conv_wgrad_params(g, (const elem_t*)dy_off, (const elem_t*)in_off, (float*)drel, rows_valid, &p, &fl));
        const int t = wgrad_group_plan(p, beta != 0.f, n.policy.wgrad_stages, n.policy);
        if (t < 0) return UDAPOSE_ERR_UNSUPPORTED;
        if (p.ksplit > 1) {
            const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
            if (det) CK(make_partial(p, (long long)drel, 0, swap ? p.Co : p.Ci));
            else if (beta == 0.f) G.zero.push_back({drel, (size_t)p.rows_valid * p.wtaps * (swap ? p.Co : p.Ci) * sizeof(float)});
        }
        const int prob = (int)tab[t].size();
        G.rel_cls.push_back(t);
        tab[t].push_back(p);
        G.flops[t] += fl;
        const int nblk = p.r_tiles * p.c_tiles * p.total_taps;
        const int ms_total = (p.M + 63) / 64, per = (ms_total + p.ksplit - 1) / p.ksplit;
        for (int z = 0; z < p.ksplit; ++z) {
            const int st = std::min(per, ms_total - z * per);
            if (st > 0) units[t].push_back(Unit{prob, z, nblk, (long)nblk * (st + 4)});
        }
        return UDAPOSE_OK;
    };
    auto add = [&](const ConvL& c) -> int { return add_geom(c.g, c.w_idx, c.dy_off, c.in_off, -1); };
    if (upper) {   // head: dy is channel-padded to 64, only the K real rows of dW exist
        ConvGeom hg = n.head.g;
        hg.Co = 64;
        CK(add_geom(hg, n.head.w_idx, n.ws_dyhead, n.head.in_off, n.K));
        G.flops[1] -= 2.0 * n.N * n.Hout * n.Wout * 256.0 * (64 - n.K);   // (count the K real channels only)
    }
    if (lower && n.policy.wgrad_group_stem) {
        // stem (Ci == 8): row-tap form into the padded [Co][KH][8][8] scratch in the workspace (always zeroed, always split:
        // 8192 stages), unpacked into the real [Co][KH][KW][3] gradient after the launch
        WgParams p;
        double fl = 0.0;
        CK(conv_wgrad_params(n.stem.g, (const elem_t*)n.stem.dy_off, (const elem_t*)n.stem.in_off, (float*)n.ws_dwtmp, -2, &p, &fl));
        p.flags |= WG_FLAG_DW_WS;
        const int t = wgrad_group_plan(p, 1, n.policy.wgrad_stages, n.policy);      // (accumulate = 1: atomics into the zeroed scratch)
        if (t != 1) return UDAPOSE_ERR_UNSUPPORTED;
        if (det) CK(make_partial(p, (long long)n.ws_dwtmp, 1, 64));     // ... or partial tiles, summed into the scratch
        const int prob = (int)tab[t].size();
        tab[t].push_back(p);
        G.flops[t] += fl;
        const int nblk = p.r_tiles * p.c_tiles * p.total_taps;
        const int ms_total = (p.M + 63) / 64, per = (ms_total + p.ksplit - 1) / p.ksplit;
        for (int z = 0; z < p.ksplit; ++z) {
            const int st = std::min(per, ms_total - z * per);
            if (st > 0) units[t].push_back(Unit{prob, z, nblk, (long)nblk * (st + 4)});
        }
    }
    if (upper) for (int i = 2; i >= 0; --i) CK(add(n.up[i]));
    for (int bi = sel.hi; bi >= sel.lo; --bi) {
        Block& b = n.blocks[bi];
        CK(add(b.c3)); CK(add(b.c2));
        if (b.has_ds) CK(add(b.cd));
        CK(add(b.c1));
    }
    if (sizing) { if (need) *need = part_cur; return UDAPOSE_OK; }
    if (part_cur > n.ws_wgpart_bytes) return UDAPOSE_ERR_NOT_PREPARED;      // (the policy's split length changed after the workspace was sized: udapose_net_ws_bytes)
    for (int t = 0; t < WG_CLASSES; ++t) {
        if (G.d_tab[t]) { (void)hipFree(G.d_tab[t]); G.d_tab[t] = nullptr; }
        if (G.d_blk[t]) { (void)hipFree(G.d_blk[t]); G.d_blk[t] = nullptr; }
        G.per_xcd[t] = 0;
        if (tab[t].empty()) continue;
        // deal whole (layer, split) units to the 8 XCDs, largest first, each to the least loaded XCD
        std::stable_sort(units[t].begin(), units[t].end(), [](const Unit& a, const Unit& b) { return a.load > b.load; });
        std::vector<WgGroupBlk> lst[8];
        long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (const Unit& u : units[t]) {
            int x = 0;
            for (int k = 1; k < 8; ++k) if (load[k] < load[x]) x = k;
            load[x] += u.load;
            for (int l = 0; l < u.nblk; ++l) lst[x].push_back(WgGroupBlk{u.prob, u.z * u.nblk + l});
        }
        size_t per = 0;
        for (int k = 0; k < 8; ++k) per = std::max(per, lst[k].size());
        std::vector<WgGroupBlk> flat(8 * per, WgGroupBlk{-1, 0});
        for (int k = 0; k < 8; ++k) std::copy(lst[k].begin(), lst[k].end(), flat.begin() + k * per);
        if (hipMalloc((void**)&G.d_tab[t], tab[t].size() * sizeof(WgParams)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        if (hipMalloc((void**)&G.d_blk[t], flat.size() * sizeof(WgGroupBlk)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        if (hipMemcpy(G.d_tab[t], tab[t].data(), tab[t].size() * sizeof(WgParams), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        if (hipMemcpy(G.d_blk[t], flat.data(), flat.size() * sizeof(WgGroupBlk), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        G.per_xcd[t] = (int)per;
    }
    if (G.d_zero) { (void)hipFree(G.d_zero); G.d_zero = nullptr; }
    G.n_zero = 0;
    bool granular = !G.zero.empty();
    for (auto& z : G.zero) granular = granular && (z.second % 16 == 0) && ((((size_t)(const char*)grads[0]) + (size_t)z.first) % 16 == 0);
    if (granular) {
        std::vector<ZeroJob> zj;
        for (auto& z : G.zero) zj.push_back(ZeroJob{(long long)z.first, (long long)(z.second / 16)});
        if (hipMalloc((void**)&G.d_zero, zj.size() * sizeof(ZeroJob)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        if (hipMemcpy(G.d_zero, zj.data(), zj.size() * sizeof(ZeroJob), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        G.n_zero = (int)zj.size();
    }
    // the split-sum launch's tables: jobs, and one block per UDAPOSE_SPLIT_SUM_CHUNK elements of a job
    if (G.d_sum) { (void)hipFree(G.d_sum); G.d_sum = nullptr; }
    if (G.d_sum_blk) { (void)hipFree(G.d_sum_blk); G.d_sum_blk = nullptr; }
    G.n_sum_blk = 0;
    if (!sums.empty()) {
        std::vector<int> blk;
        for (size_t j = 0; j < sums.size(); ++j)
            for (unsigned c = 0; c < (sums[j].n + UDAPOSE_SPLIT_SUM_CHUNK - 1u) / UDAPOSE_SPLIT_SUM_CHUNK; ++c) { blk.push_back((int)j); blk.push_back((int)c); }
        if (hipMalloc((void**)&G.d_sum, sums.size() * sizeof(SumJob)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        if (hipMalloc((void**)&G.d_sum_blk, blk.size() * sizeof(int)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        if (hipMemcpy(G.d_sum, sums.data(), sums.size() * sizeof(SumJob), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        if (hipMemcpy(G.d_sum_blk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        G.n_sum_blk = (int)(blk.size() / 2);
    }
    G.k_beta = beta; G.k_stages = n.policy.wgrad_stages; G.k_part = part;
    return UDAPOSE_OK;
}

// bytes of partial tiles a whole backward pass needs with the plan's current policy (a dry walk of the table builder; part 0 covers parts 1 / 2)
size_t wg_partial_bytes(Net& n) {
    Net::WgGroup G;
    size_t need = 0;
    if (build_wg_group(n, G, nullptr, 0.f, 0, true, &need) != UDAPOSE_OK) return 0;
    return need;
}

// The tables hold OFFSETS (relative to the arenas and to grads[0]), so one pair of tables - overwrite and accumulate mode -
// serves every pass whose gradient tensors keep their relative placement (both per-pass gradient buffers of a step do).
Net::WgGroup* find_wg_group(Net& n, void* const* grads, float beta, int part) {
    for (auto& g : n.wg_groups) {
        if (g.k_beta != beta || g.k_stages != n.policy.wgrad_stages || g.k_part != part) continue;
        bool same = true;
        for (auto& r : g.rel)
            if ((const char*)grads[r.first] - (const char*)grads[0] != r.second) { same = false; break; }
        if (same) return &g;
    }
    return nullptr;
}
// net_bind_grads: allocate + upload the grouped weight-gradient tables (both accumulate modes) for this gradient placement.
// Synchronous - never inside a stream capture; net_backward itself never builds them.
int bind_wg_groups(Net& n, void* const* grads) {
    // 3 parts x 2 accumulate modes per gradient placement (a few KB of device tables each).  Tables are never evicted or rebuilt
    // in place: launches captured in a hipGraph hold their device pointers, and a process only ever uses a handful of
    // gradient placements (two per-pass buffers per network).
    for (const int part : {0, 1, 2})
        for (const float beta : {0.f, 1.f}) {
            if (find_wg_group(n, grads, beta, part)) continue;
            n.wg_groups.emplace_back();
            Net::WgGroup* G = &n.wg_groups.back();
            const int rc = build_wg_group(n, *G, grads, beta, part);
            if (rc != UDAPOSE_OK) { n.wg_groups.pop_back(); return rc; }
            G->last_use = ++n.wg_tick;
        }
    return UDAPOSE_OK;
}
// what precedes the grouped launches of one pass (the clears of the atomic form) and what follows them (the split sums, the stem's unpack)
int wg_before(hipStream_t s, Net& n, Net::WgGroup* G, char* ws, void* const* grads, bool with_stem) {
    G->last_use = ++n.wg_tick;
    if (G->d_zero) {
        CK(pw_zero_multi(s, G->d_zero, G->n_zero, grads[0]));
    } else {
        for (auto& z : G->zero)
            if (pw_zero(s, (char*)grads[0] + z.first, z.second) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
    }
    const ConvGeom& sg = n.stem.g;
    if (with_stem && !(n.policy.wgrad_det && n.policy.wgrad_group) && pw_zero(s, ws + n.ws_dwtmp, (size_t)sg.Co * sg.KH * 8 * 8 * sizeof(float)) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
    return UDAPOSE_OK;
}
int wg_after(hipStream_t s, Net& n, Net::WgGroup* G, char* ws, void* const* grads, float beta, bool with_stem, bool sums_done = false) {
    if (G->n_sum_blk && !sums_done) CK(pw_split_sum(s, G->d_sum, G->d_sum_blk, G->n_sum_blk, ws, (char*)grads[0]));
    const ConvGeom& sg = n.stem.g;
    if (with_stem)
        CK(pw_unpack_strided(s, (const float*)(ws + n.ws_dwtmp), (float*)grads[n.stem.w_idx], sg.Co, sg.KH, sg.KWp(), sg.KW, 8, 3,
                             (long)sg.KH * sg.KW * 3, (long)sg.KW * 3, 3, 1, beta));
    return UDAPOSE_OK;
}
int run_wg_group(hipStream_t s, Net& n, const char* act, char* ws, void* const* grads, float beta, int part) {
    Net::WgGroup* G = find_wg_group(n, grads, beta, part);
    if (!G) return UDAPOSE_ERR_NOT_PREPARED;
    const bool with_stem = part_sel(n, part).stem && n.policy.wgrad_group_stem;
    CK(wg_before(s, n, G, ws, grads, with_stem));
    for (int t = 0; t < WG_CLASSES; ++t) {
        if (!G->per_xcd[t]) continue;
        const int tok = conv_prof_before(s, 2, G->flops[t]);
        const int rc = wgrad_group_launch(s, t, G->d_tab[t], G->d_blk[t], G->per_xcd[t], act, ws, grads[0]);
        conv_prof_after(s, tok);
        CK(rc);
    }
    return wg_after(s, n, G, ws, grads, beta, with_stem);
}
// The grouped weight gradients of TWO passes of this plan (each with its own arenas, gradient tensors and accumulate mode) as ONE
// launch per tile class: the step's two student passes end at about the same time and their weight-gradient launches are fully
// exposed there; one grid of twice the size has half the tail (measured on the launches alone: 2498 us for 64 images against 2 x 1353).
int run_wg_pair(hipStream_t s, Net& n, const char* actA, char* wsA, void* const* gradsA, float betaA, const char* actB, char* wsB,
                void* const* gradsB, float betaB, int part) {
    Net::WgGroup* GA = find_wg_group(n, gradsA, betaA, part);
    Net::WgGroup* GB = find_wg_group(n, gradsB, betaB, part);
    if (!GA || !GB) return UDAPOSE_ERR_NOT_PREPARED;
    bool same = true;
    for (int t = 0; t < WG_CLASSES; ++t) same = same && GA->per_xcd[t] == GB->per_xcd[t];
    if (!same) {        // (different table shapes: two launches)
        CK(run_wg_group(s, n, actA, wsA, gradsA, betaA, part));
        return run_wg_group(s, n, actB, wsB, gradsB, betaB, part);
    }
    const bool with_stem = part_sel(n, part).stem && n.policy.wgrad_group_stem;
    CK(wg_before(s, n, GA, wsA, gradsA, with_stem));
    CK(wg_before(s, n, GB, wsB, gradsB, with_stem));
    for (int t = 0; t < WG_CLASSES; ++t) {
        if (!GA->per_xcd[t]) continue;
        const int tok = conv_prof_before(s, 2, GA->flops[t] + GB->flops[t]);
        const int rc = wgrad_group_launch(s, t, GA->d_tab[t], GA->d_blk[t], GA->per_xcd[t], actA, wsA, gradsA[0], GB->d_tab[t], GB->d_blk[t], actB, wsB,
                                          gradsB[0]);
        conv_prof_after(s, tok);
        CK(rc);
    }
    // (equal accumulate modes: the two passes' split sums share their job table - one launch over both workspaces)
    const bool pair_sum = betaA == betaB && GA->n_sum_blk > 0 && GA->n_sum_blk == GB->n_sum_blk;
    if (pair_sum) CK(pw_split_sum(s, GA->d_sum, GA->d_sum_blk, GA->n_sum_blk, wsA, (char*)gradsA[0], wsB, (char*)gradsB[0]));
    CK(wg_after(s, n, GA, wsA, gradsA, betaA, with_stem, pair_sum));
    return wg_after(s, n, GB, wsB, gradsB, betaB, with_stem, pair_sum);
}
}  // namespace

int net_wgrad_pair(void* h, hipStream_t s, const void* actA, void* wsA, void* const* gradsA, float betaA, const void* actB, void* wsB,
                   void* const* gradsB, float betaB, int part) {
    Net& n = *(Net*)h;
    if (part < 0 || part > 2 || n.f32 || !n.policy.wgrad_group) return UDAPOSE_ERR_ARG;
    DbgSyncScope dbg(n.policy.debug_sync);
    return run_wg_pair(s, n, (const char*)actA, (char*)wsA, gradsA, betaA, (const char*)actB, (char*)wsB, gradsB, betaB, part);
}

// grads[i] (fp32, same physical layout as params[i]) = beta*grads[i] + d loss / d params[i]; beta in {0,1}
// part 0: the whole backward (one grouped weight-gradient launch per tile class at its end).
// part 1 / part 2: the same backward cut after the first block of layer3 (split_block): part 1 = head, deconvs, layer4, layer3
// and THEIR weight gradients - 94 % of the parameters, a contiguous suffix of the flat gradient buffer, final when part 1 ends;
// part 2 = layer2, layer1, stem and theirs, continuing from the gradient part 1 left in the workspace.  Data parallel:
// the all-reduce of the suffix runs under part 2 (engine.py).  The two parts together enqueue exactly the kernels of part 0
// except that the weight gradients are two grouped launches per tile class instead of one.
// phase 0: the gradient chain, then the grouped weight-gradient launches of this part; 1: the chain only; 2: the grouped launches
// only (every layer owns its dy buffer in the workspace, so a caller may run them on another stream under the chain of part 2).
int net_backward(void* h, hipStream_t s, const float* dout_nchw, const void* const* params, const void* wpack_, void* act_, void* ws_,
                 void* const* grads, float beta, int part, int phase) {
    Net& n = *(Net*)h;
    if (part < 0 || part > 2 || phase < 0 || phase > 2) return UDAPOSE_ERR_ARG;
    if (phase != 0 && !n.policy.wgrad_group) return UDAPOSE_ERR_ARG;
    if (phase == 2) {
        if (n.f32 || n.fwd_only) return UDAPOSE_ERR_UNSUPPORTED;
        DbgSyncScope dbg2(n.policy.debug_sync);
        return run_wg_group(s, n, (const char*)act_, (char*)ws_, grads, beta, part);
    }
    DbgSyncScope dbg(n.policy.debug_sync);
    if (n.f32 || n.fwd_only) return UDAPOSE_ERR_UNSUPPORTED;   // fp32 / f16x2 modes and forward-only plans keep nothing for a backward
    const char* wpack = (const char*)wpack_;
    char* act = (char*)act_;
    char* ws = (char*)ws_;
    Pool pool;
    pool.base = ws;
    for (int i = 0; i < 6; ++i) pool.off[i] = n.ws_gbuf[i];
    const bool grouped = n.policy.wgrad_group != 0;
    if (grouped && (!find_wg_group(n, grads, beta, part))) return UDAPOSE_ERR_NOT_PREPARED;     // (before anything is enqueued)
    const int HWo = n.Hout * n.Wout;
    const int split = split_block(n);
    const bool fused = n.policy.bn_bwd_fused != 0;
    DgradBnStat cur, nxt;
    bool have = false;
    elem_t* dz = nullptr;
    if (part != 2) {
    // head
    elem_t* dyh = (elem_t*)(ws + n.ws_dyhead);
    CK(pw_nchw_f32_to_nhwc_bf16(s, dout_nchw, dyh, n.N, n.K, HWo, 64));
    ConvGeom hg = n.head.g;
    hg.Co = 64;   // dy is channel-padded to 64 (one 64-wide K step)
    float* tmp = (float*)(ws + n.ws_dwtmp);
    if (!grouped) {
        CK(conv_wgrad(s, hg, dyh, (const elem_t*)(act + n.head.in_off), tmp, 0, n.K));
        CK(pw_unpack_strided(s, tmp, (float*)grads[n.head.w_idx], n.K, 1, 1, 1, 256, 256, 256, 0, 0, 1, beta));
    }
    dz = pool.get();
    // Gradients entering the BatchNorm backward of the three deconv layers are kept in fp32: close to the loss the BN
    // projection (g - mean(g) - xhat*mean(g*xhat)) cancels ~90 % of g, so bf16 rounding of g is amplified ~10x in dy
    // (measured against the backward of the bf16-storage emulation: 2.4 % / 8 % / 14 % relative error per layer with
    // bf16 g).  The tensors are small (N*64*64*256 and below).
    //
    // Fused chain (g_bn_bwd_fused): every dgrad launch knows the BatchNorm that consumes its output; its epilogue applies
    // that BN's ReLU mask and reduces sum(g), sum(g*xhat) per m-tile, so a BN backward is ONE launch (column sums of the
    // slab + apply) for the wide layers and finalize + apply for the others.  `cur` describes the pending statistics of dz.
    if (fused) { cur = bn_stat_of(n, n.up[2], n.up_bn[2], params, act, ws, 2); have = true; }
    CK(conv_dgrad(s, hg, dyh, (const elem_t*)(wpack + n.head.wb_off), dz, nullptr, 1, have ? &cur : nullptr));
    // (the head's bias gradient is off the gradient chain: launched behind the first data gradient, not in front of it)
    CK(pw_plane_sum(s, dout_nchw, (float*)grads[n.head.bias_idx], n.N, n.K, HWo, beta));
    // deconv stack
    for (int i = 2; i >= 0; --i) {
        elem_t* dx = nullptr;
        if (fused) {
            if (i > 0) nxt = bn_stat_of(n, n.up[i - 1], n.up_bn[i - 1], params, act, ws, 2);
            else nxt = bn_stat_of(n, n.blocks.back().c3, n.blocks.back().b3, params, act, ws, 1);
        }
        CK(conv_bn_bwd(s, n, n.up[i], n.up_bn[i], params, wpack, act, ws, grads, beta, pool, dz, 1, nullptr, 2, nullptr, &dx, true, i > 0, grouped,
                       have ? &cur : nullptr, fused ? &nxt : nullptr));
        if (fused) cur = nxt;
        pool.put(dz);
        dz = dx;
    }
    } else {
        // part 2 resumes where part 1 stopped: the gradient entering block split-1 sits in the pool buffer part 1 ended on
        // (the buffer sequence is a function of the plan alone), masked for that block's bn3 with its sums in the slab
        if (split < 1 || split >= (int)n.blocks.size() || n.split_dz_idx < 0) return UDAPOSE_ERR_ARG;
        pool.used[n.split_dz_idx] = true;
        dz = (elem_t*)(ws + pool.off[n.split_dz_idx]);
        if (fused) {
            cur = bn_stat_of(n, n.blocks[split - 1].c3, n.blocks[split - 1].b3, params, act, ws, 1);
            cur.rows = conv_dgrad_stat_rows(n.blocks[split].c1.g);
            have = true;
        }
    }
    // bottlenecks, last to first
    const int bi_hi = part == 2 ? split - 1 : (int)n.blocks.size() - 1, bi_lo = part == 1 ? split : 0;
    for (int bi = bi_hi; bi >= bi_lo; --bi) {
        Block& b = n.blocks[bi];
        // bn3 (+ReLU of the block output): g = masked dz feeds the skip branch (written in place unless the producing dgrad
        // already masked it)
        elem_t *dz2 = nullptr, *dz1 = nullptr, *dxd = nullptr, *dxin = nullptr;
        if (fused) nxt = bn_stat_of(n, b.c2, b.b2, params, act, ws, 2);
        CK(conv_bn_bwd(s, n, b.c3, b.b3, params, wpack, act, ws, grads, beta, pool, dz, 0, dz, 1, nullptr, &dz2, true, 0, grouped, have ? &cur : nullptr,
                       fused ? &nxt : nullptr));
        if (fused) { cur = nxt; nxt = bn_stat_of(n, b.c1, b.b1, params, act, ws, 2); }
        CK(conv_bn_bwd(s, n, b.c2, b.b2, params, wpack, act, ws, grads, beta, pool, dz2, 0, nullptr, 2, nullptr, &dz1, true, 0, grouped,
                       have ? &cur : nullptr, fused ? &nxt : nullptr));
        if (fused) cur = nxt;
        pool.put(dz2);
        const elem_t* skip = dz;
        if (b.has_ds) {
            // the downsample BN shares g with bn3 (already masked): its own reduce / apply, no ReLU
            CK(conv_bn_bwd(s, n, b.cd, b.bd, params, wpack, act, ws, grads, beta, pool, dz, 0, nullptr, 0, nullptr, &dxd, true, 0, grouped));
            skip = dxd;
        }
        const bool chain = fused && bi > 0;      // dxin feeds bn3 of the previous block (block 0: the max-pool backward)
        if (chain) nxt = bn_stat_of(n, n.blocks[bi - 1].c3, n.blocks[bi - 1].b3, params, act, ws, 1);
        CK(conv_bn_bwd(s, n, b.c1, b.b1, params, wpack, act, ws, grads, beta, pool, dz1, 0, nullptr, 2, skip, &dxin, true, 0, grouped,
                       have ? &cur : nullptr, chain ? &nxt : nullptr));
        if (chain) cur = nxt;
        pool.put(dz1);
        if (dxd) pool.put(dxd);
        pool.put(dz);
        dz = dxin;
    }
    if (part == 1) {
        for (int i = 0; i < 6; ++i) if ((char*)dz == ws + pool.off[i]) n.split_dz_idx = i;      // (the same value on every pass)
        if (grouped && phase == 0) CK(run_wg_group(s, n, act, ws, grads, beta, 1));
        if (beta == 0.f) {       // backbone.fc lies in part 1's suffix of the gradient buffer (not part of forward: zero)
            if (pw_zero(s, grads[n.fc_w_idx], (size_t)1000 * 2048 * 4) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
            if (pw_zero(s, grads[n.fc_b_idx], (size_t)1000 * 4) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
        }
        return UDAPOSE_OK;
    }
    // stem: maxpool -> bn/relu -> conv (no input gradient)
    if (n.policy.stem_fused >= 2 && grouped && n.policy.wgrad_group_stem) {
        // the max-pool backward is gathered inside the BN backward's two sweeps: the full-resolution gradient is never stored
        const BnL& sb = n.stem_bn;
        const float* save = (const float*)(act + sb.save_off);
        CK(pw_bn_bwd_pooled(s, dz, (const unsigned char*)(act + n.poolidx_off), n.Hs, n.Ws, (const elem_t*)(act + n.stem.y_off),
                            (elem_t*)(ws + n.stem.dy_off), sb.npix, sb.C, (const float*)params[sb.g_idx], save, save + sb.C, (float*)(ws + n.ws_slab),
                            (float*)(ws + n.ws_coef) + 4096, (float*)grads[sb.g_idx], (float*)grads[sb.b_idx], beta, (const float*)params[sb.b_idx]));
        pool.put(dz);
    } else {
    elem_t* dzs = pool.get();
    CK(pw_maxpool3x3s2_bwd(s, dz, (const unsigned char*)(act + n.poolidx_off), dzs, n.N, n.Hs, n.Ws, 64));
    pool.put(dz);
    elem_t* none = nullptr;
    CK(conv_bn_bwd(s, n, n.stem, n.stem_bn, params, wpack, act, ws, grads, beta, pool, dzs, 0, nullptr, 2, nullptr, &none, false, 0, grouped));
    pool.put(dzs);
    }
    if (grouped && phase == 0) CK(run_wg_group(s, n, act, ws, grads, beta, part));
    // backbone.fc is not part of the forward: zero gradient when overwriting
    if (beta == 0.f && part == 0) {
        if (pw_zero(s, grads[n.fc_w_idx], (size_t)1000 * 2048 * 4) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
        if (pw_zero(s, grads[n.fc_b_idx], (size_t)1000 * 4) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
    }
    return UDAPOSE_OK;
}

// Apply the running-statistics update (momentum form, counter += 1) of every BN layer from the batch statistics saved in
// `act` by a forward that ran with the update deferred (training | 2): keeps the reference's update ORDER when two
// forwards of one module run concurrently on different streams.
int net_apply_running(void* h, hipStream_t s, const void* act_, void* const* buffers, float momentum) {
    Net& n = *(Net*)h;
    // one launch for all layers through the job table net_bind built for these buffers
    if (!n.d_runjobs || n.runjobs_key != buffers[0]) return UDAPOSE_ERR_NOT_PREPARED;
    return pw_bn_running_update_multi(s, n.d_runjobs, n.n_runjobs, 2048, act_, momentum);
}
int net_bind_grads(void* h, void* const* grads) {
    Net& n = *(Net*)h;
    if (n.f32) return UDAPOSE_OK;
    if (n.fwd_only) return UDAPOSE_ERR_UNSUPPORTED;
    return bind_wg_groups(n, grads);
}
// index (in .parameters() order) of the first parameter whose gradient is final when net_backward part 1 has run: the first
// parameter of layer3's first block.  Everything from there on (layer3, layer4, fc, upsampling, head) is part 1's.
long long net_grad_split_param(void* h) {
    Net& n = *(Net*)h;
    const int split = split_block(n);
    if (split < 1 || split >= (int)n.blocks.size()) return -1;
    return n.blocks[split].c1.w_idx;
}

// ---- fused optimizer tail: Adam on the student, EMA into the teacher, and the element-type weight packs of BOTH networks'
// plans in one sweep (optim.hip opt_tail_k).  hs / ht: the student's and the teacher's plans (same architecture).
// h_m / h_v: host arrays of the Adam moments per parameter index, NULL entries for parameters without gradient (backbone.fc:
// EMA only).  bind builds the device job table (allocates: outside capture); the update itself only launches.
int net_bind_update(void* hs, void* ht, void* const* params_s, void* const* grads, void* const* h_m, void* const* h_v, void* const* params_t,
                    void* wpack_s_, void* wpack_t_) {
    Net& n = *(Net*)hs;
    const Net& nt = *(const Net*)ht;
    // teacher in the f16x2 mode (the reference's precision mix: fp16 student, fp32-grade teacher): its split packs are not written
    // by this sweep (the caller re-packs the teacher's plan with udapose_net_pack_weights after it), the EMA still is
    const bool t_split = nt.f32 == 2;
    if (n.f32 || nt.f32 == 1 || n.n_params != nt.n_params || (!t_split && n.wpack_bytes != nt.wpack_bytes)) return UDAPOSE_ERR_UNSUPPORTED;
    char* ws_ = (char*)wpack_s_;
    char* wt_ = (char*)wpack_t_;
    const size_t jb = opt_tail_job_bytes();
    std::vector<char> jobs;
    std::vector<int> bj, bs;
    std::vector<char> covered(n.n_params, 0);
    auto push = [&](int idx, void* sd, void* td, void* sx, void* tx, int A, int T, int B) -> int {
        const long long numel = n.param_numel[idx];
        const int adam = (h_m[idx] && h_v[idx] && grads[idx]) ? 1 : 0;
        if (A) {
            const uintptr_t al = (uintptr_t)params_s[idx] | (uintptr_t)params_t[idx] | (adam ? ((uintptr_t)grads[idx] | (uintptr_t)h_m[idx] | (uintptr_t)h_v[idx]) : 0);
            if ((al & 15) || (A & 63) || (B & 63) || (long long)A * T * B != numel) return UDAPOSE_ERR_UNSUPPORTED;
        }
        jobs.resize(jobs.size() + jb);
        opt_tail_job_fill(jobs.data() + jobs.size() - jb, (float*)params_s[idx], (const float*)grads[idx], (float*)h_m[idx], (float*)h_v[idx],
                          (float*)params_t[idx], sd, td, sx, tx, A, T, B, adam, numel);
        const int j = (int)(jobs.size() / jb) - 1;
        const long nb = A ? (long)(A / 64) * (B / 64) * T : (long)((numel + opt_chunk() - 1) / opt_chunk());
        for (long k = 0; k < nb; ++k) { bj.push_back(j); bs.push_back((int)k); }
        covered[idx] = 1;
        return UDAPOSE_OK;
    };
    auto conv = [&](const ConvL& c) -> int {
        const ConvGeom& g = c.g;
        const int T = g.KH * g.KW;
        if (g.smallc()) return push(c.w_idx, nullptr, nullptr, nullptr, nullptr, 0, 0, 0);     // stem: packed by its own strided launch
        if (!g.transposed)   // master [Co][T][Ci]: fprop pack = cast, dgrad pack = per-tap transpose; the teacher needs the fprop pack
            return push(c.w_idx, ws_ + c.wf_off, t_split ? nullptr : wt_ + c.wf_off, ws_ + c.wb_off, nullptr, g.Co, T, g.Ci);
        // ConvTranspose2d master [Ci][T][Co]: dgrad pack = cast, fprop pack = per-tap transpose
        return push(c.w_idx, ws_ + c.wb_off, nullptr, ws_ + c.wf_off, t_split ? nullptr : wt_ + c.wf_off, g.Ci, T, g.Co);
    };
    CK(conv(n.stem));
    for (auto& b : n.blocks) {
        CK(conv(b.c1)); CK(conv(b.c2)); CK(conv(b.c3));
        if (b.has_ds) CK(conv(b.cd));
    }
    for (int i = 0; i < 3; ++i) CK(conv(n.up[i]));
    CK(push(n.head.w_idx, ws_ + n.head.wf_off, t_split ? nullptr : wt_ + n.head.wf_off, nullptr, nullptr, 0, 0, 0));      // [K][256]: the fprop pack is a cast
    for (int i = 0; i < n.n_params; ++i)
        if (!covered[i]) CK(push(i, nullptr, nullptr, nullptr, nullptr, 0, 0, 0));                   // BN vectors, head bias, backbone.fc
    Net::UpdTab& u = n.upd;
    if (u.jobs) { (void)hipFree(u.jobs); (void)hipFree(u.blk_job); (void)hipFree(u.blk_sub); u.jobs = nullptr; }
    if (hipMalloc(&u.jobs, jobs.size()) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMalloc((void**)&u.blk_job, bj.size() * sizeof(int)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMalloc((void**)&u.blk_sub, bs.size() * sizeof(int)) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMemcpy(u.jobs, jobs.data(), jobs.size(), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMemcpy(u.blk_job, bj.data(), bj.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    if (hipMemcpy(u.blk_sub, bs.data(), bs.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    u.nblocks = (int)bj.size();
    u.k_ps = params_s[0]; u.k_pt = params_t[0]; u.k_g = grads[0]; u.k_m = h_m[0]; u.k_ws = wpack_s_; u.k_wt = wpack_t_;
    return UDAPOSE_OK;
}

int net_fused_update(void* hs, void* ht, hipStream_t s, void* const* params_s, void* const* grads, void* const* h_m, void* const* params_t,
                     void* wpack_s_, void* wpack_t_, float lr, float beta1, float beta2, float eps, float wd, int step, float gscale,
                     float* dev_state, float alpha, float oma, int do_adam, long long grad2_delta) {
    Net& n = *(Net*)hs;
    const Net& nt = *(const Net*)ht;
    DbgSyncScope dbg(n.policy.debug_sync);
    const Net::UpdTab& u = n.upd;
    if (!u.jobs || u.k_ps != params_s[0] || u.k_pt != params_t[0] || u.k_g != grads[0] || u.k_m != h_m[0] || u.k_ws != wpack_s_ || u.k_wt != wpack_t_)
        return UDAPOSE_ERR_NOT_PREPARED;
    CK(opt_tail(s, u.jobs, u.blk_job, u.blk_sub, u.nblocks, lr, beta1, beta2, eps, wd, step, gscale, dev_state, alpha, oma, do_adam, grad2_delta, 1));
    // the two packs that are not a cast or a per-tap transpose of a whole tensor: the stem's 3 -> 8 channel gather (both
    // networks) and the head's zero-padded dgrad pack (student)
    CK(pack_conv(s, n, n.stem, (const void* const*)params_s, (char*)wpack_s_, false));
    if (nt.f32 != 2) CK(pack_conv(s, nt, nt.stem, (const void* const*)params_t, (char*)wpack_t_, false));
    CK(pw_pack_strided(s, (const float*)params_s[n.head.w_idx], (elem_t*)((char*)wpack_s_ + n.head.wb_off), 256, 1, 1, 1, 64, n.K, 1, 0, 0, 256));
    return UDAPOSE_OK;
}
