// Host-side description of one convolution and the tap plans (fprop / dgrad / wgrad) the igemm kernels execute.
#pragma once
#include <atomic>
#include <mutex>
#include <vector>
#include "igemm.h"

// Explicit dispatch policy.  A default-constructed Policy IS the measured production policy; nothing in the library reads an
// environment variable or a mutable global to choose a kernel.  A network plan owns a copy (udapose_net_set_policy), a
// single convolution call names one through its descriptor (udapose_conv_desc.policy, NULL = default).  The non-default
// values exist for tests (force a code path) and tuning (A/B runs through bench.py flags).
struct Policy {
    int igemm_tile = -1;        // force one igemm tile configuration id (-1: the heuristics of igemm_pick_tile)
    int igemm_h3 = 1;           // run-staged 3x3 form: 0 off, 1 measured per-shape policy, 2 / 3: force the 64- / 128-row form
    int igemm_lean = 1;         // lean 1x1 form (saddr LDS-DMA loads) where eligible
    int igemm_short_lds = 1;    // one-stage LDS request for launches whose K loop is one stage
    int igemm_tap0 = 1;         // 1x1 kernels skip the tap-table read
    int wgrad_tile = -1;        // force a weight-gradient tile id (-1: heuristics)
    int wgrad_ksplit = -1;      // force the pixel-split count of a per-layer weight-gradient launch
    int wgrad_fastgeo = 2;      // weight-gradient loader on power-of-two maps: 0 general, 1 bit-field pixel coordinates (pointer selects),
                                // 2 the same through buffer_load ... lds with out-of-range zero fill and an unrolled ring (wgrad_fast2_body)
    int wgrad_group = 1;        // net backward: one grouped weight-gradient launch per tile class (0: layer by layer)
    int wgrad_stages = 128;     // grouped launch: 64-pixel stages a work-group reduces before a layer's pixel range is split
    int wgrad_group_stem = 1;   // the Ci == 8 stem joins the 64x64 group in its row-tap form
    int bn_bwd_fused = 1;       // net backward: dgrad epilogues mask for the consumer BatchNorm and reduce its backward sums
    int bn_fwd_chunked = 1;     // BN forward: finalize + apply in one channel-chunked launch where it pays
    int bn_bwd_chunked = 1;     // BN backward: channel-chunked forms without a finalize launch
    int bn_bwd_pre_legacy = 0;  // BN backward from pre-reduced sums through the generic apply kernel (A/B)
    int igemm_wg_min = 512;     // 128x64 tiles as soon as they give this many work-groups (else 64x64): 2 per CU measured best in-step
    int wgrad_row3 = 1;         // 3x3 stride-1 convs: one weight-gradient work-group per (64x64 tile, filter row), three taps sharing the staged
                                // operands (a third of the LDS fill per FLOP).  As fast as the 128x128 one-tap form alone; in the grouped launch
                                // with the 64x64 kernel at four work-groups per CU: 1366 vs 1387 us per pass alone, -0.10 ms per step
    int bn3_mask = 1;           // block outputs: the forward saves the ReLU bit mask, the data gradients read it instead of z (0: read z)
    int stem_fused = 1;         // stem: 1 = BN apply + ReLU + max-pool in one sweep; 2 = also the max-pool backward gathered inside the BN backward's
                                // two sweeps (0.2 GB less traffic, but 99 + 87 us against 55 + 30 + 48 us for the three separate launches: neutral in the step)
    int igemm_big_min = 0;      // > 0: tile 4 (128x128, 2-stage ring) when Co % 128 == 0 and the 128x64 grid has >= this many work-groups
    int patch_conv = 2;         // reflection-padded 3x3 stride-1 convolutions (the style network) through the patch-staged kernels of patchconv.hip
                                // (the input patch staged once instead of once per tap): 0 = never (the igemm for every layer), 1 = the 64 -> 3
                                // and 3 -> 64 end layers only, 2 = the trunk layers too (128 pixels x 64 channels per work-group), 3 = 128
                                // channels per work-group in the 16-bit form where Co % 128 == 0 (measured equal to 2)
    int eval_fold = 1;          // eval-mode forwards (validate()): BatchNorm's running-statistics scale / shift, the residual and the ReLU are applied in
                                // the convolution's epilogue - z is written by the conv, no BN-apply launch, no pre-BN tensor (0: conv + apply launches)
    int bn_xcd_rows = 1;        // BatchNorm apply kernels (forward and backward, chunked and streaming forms): XCD k processes the k-th eighth of the pixel
                                // rows, the rows the implicit GEMMs' work-groups on XCD k produce and consume (each XCD owns a contiguous range of
                                // m-tiles there), so activations cross the conv <-> BatchNorm kernel boundaries through that XCD's L2
                                // (tools/probe/l2_handoff.hip: 17.9 against 6.7 TB/s); bit-identical results, -0.06..-0.15 ms per step (r4_ab_runs.txt)
    int wgrad_det = 1;          // grouped weight gradients: split pixel reductions store per-split partial tiles that ONE launch adds in split order
                                // (bit-reproducible gradients; 0: fp32 atomics into cleared tensors, arrival order - rounds 1-5)
    int igemm_ns3_k = 0;        // 64x64 igemm tiles: 3-stage ring from this K on, 2-stage below (0 = 2048)
    int debug_sync = 0;         // net calls: synchronise after every stage and report the first failing source line
    unsigned long long* timeline = nullptr;   // device buffer for per-work-group timeline stamps (tuning), normally null
};
inline const Policy& default_policy() { static const Policy p; return p; }

// hipFuncSetAttribute (dynamic LDS size) is a per-DEVICE property of a kernel: run `f` once per (call site, device ordinal).
template <typename F>
inline void once_per_device(std::atomic<unsigned long long>& done, std::mutex& mu, F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    std::lock_guard<std::mutex> lk(mu);
    if (done.load(std::memory_order_relaxed) & bit) return;
    f();
    done.fetch_or(bit, std::memory_order_release);
}

struct ConvGeom {
    int N, Hi, Wi, Ci;      // logical input (for transposed: the small side)
    int Co, KH, KW, stride, pad;
    int transposed;         // 1 = ConvTranspose2d(k, stride, pad, output_padding = 0)
    int reflect;            // reflection padding (style net)
    int upsample;           // nearest x2 upsample folded into the loader (style decoder); Hi/Wi are the PHYSICAL dims
    const Policy* pol = nullptr;   // dispatch policy of this call (null: the default = production policy)
    const Policy& policy() const { return pol ? *pol : default_policy(); }
    int Ho() const { return transposed ? (Hi - 1) * stride - 2 * pad + KH : (((Hi << upsample) + 2 * pad - KH) / stride + 1); }
    int Wo() const { return transposed ? (Wi - 1) * stride - 2 * pad + KW : (((Wi << upsample) + 2 * pad - KW) / stride + 1); }
    bool smallc() const { return Ci == 8; }
    int KWp() const { return smallc() ? ((KW + 7) & ~7) : KW; }   // taps padded so that 8 taps fill a 64-wide K step
    int wtaps() const { return KH * KWp(); }
};

struct TapPlan {
    std::vector<IgTap> taps;
    int nclass = 1;
    IgClass cls[4];
    const IgTap* d_taps = nullptr;   // device copy (owned by the plan cache)
};

// direction 0: fprop (also the plan wgrad walks), 1: dgrad, 2: row-tap wgrad of a Ci == 8 conv.  Plans (a few hundred bytes of
// device memory each) are cached per (device, geometry class); building one allocates and copies synchronously, so a plan
// that is missing while `stream` is being captured is NOT built: the call returns null (UDAPOSE_ERR_NOT_PREPARED at the ABI).
// conv_prepare builds the plans of a geometry up front (udapose_conv_prepare / udapose_net_bind).
const TapPlan* get_tap_plan(const ConvGeom& g, int direction, hipStream_t stream = nullptr);
int conv_prepare(const ConvGeom& g);

int igemm_pick_tile(int M, int Co, int nclass, int K, int h3_ok, const Policy& pol);   // h3_ok: 3x3 stride-1 pad-1 same-size conv (conv_h3_ok)
int conv_h3_ok(const ConvGeom& g);
int igemm_stat_rows(int M, int Co, int nclass, int tile);
int igemm_launch(IgParams& p, int tile, hipStream_t stream, const Policy& pol);
int wgrad_pick_tile(int Rdim, int Cdim, int smallc, const Policy& pol);
int wgrad_launch(WgParams& p, int tile, int accumulate, hipStream_t stream, const Policy& pol);
// Grouped wgrad (many layers, one launch per tile class).  wgrad_group_plan completes p for the group kernels and returns
// the tile class (0 = 128x128, 1 = 64x64) or < 0 when the layer needs its own launch; stages_per_block bounds a work-group's
// pixel range (longer reductions are split and accumulated with fp32 atomics into a zeroed dW).
#define WG_CLASSES 2      // tile classes of a grouped launch: 0 = 128x128, 1 = 64x64 (+ filter-row form)
int wgrad_group_plan(WgParams& p, int accumulate, int stages_per_block, const Policy& pol);
// the x / dy / dw fields of the table entries are byte offsets from the three bases
int wgrad_group_launch(hipStream_t stream, int tile, const WgParams* d_tab, const WgGroupBlk* d_blk, int per_xcd, const void* x_base,
                       const void* dy_base, void* dw_base, const WgParams* d_tab2 = nullptr, const WgGroupBlk* d_blk2 = nullptr,
                       const void* x_base2 = nullptr, const void* dy_base2 = nullptr, void* dw_base2 = nullptr);

struct ConvEpilogue {
    const elem_t* res = nullptr;
    const float* bias = nullptr;
    const float* scale = nullptr;   // per-channel factor applied to the fp32 result before the bias (igemm path only; eval-mode BN folding)
    float* stats = nullptr;
    int relu = 0;
    int out_f32 = 0;
    int f32 = 0;            // x, w, res, y are fp32 (exact fp32 MFMA path; forward only)
    int split = 0;          // x, w, res (and y unless out_f32) are f16x2 split tensors: the fp32-grade mode (forward only)
};
// y = conv(x, w_fwd[Co][wtaps][Ci])
int conv_fprop(hipStream_t s, const ConvGeom& g, const elem_t* x, const elem_t* w_fwd, void* y, const ConvEpilogue& e);
// patch-staged forms of the style network's end layers (patchconv.hip); conv_fprop routes to them when patch_conv_ok
int patch_conv_ok(const ConvGeom& g, const ConvEpilogue& e);
int patch_conv_fprop(hipStream_t s, const ConvGeom& g, const void* x, const void* w_fwd, void* y, const ConvEpilogue& e);
// The BatchNorm whose backward consumes a dgrad's output (IgParams::bs_*): the dgrad epilogue masks dx with that BN's ReLU
// and leaves the partial sums of g and g * xhat in slab[rows][2][C]; `rows` is set by conv_dgrad.
struct DgradBnStat {
    const elem_t* y = nullptr;      // the BN's input (pre-BN conv output)
    const elem_t* z = nullptr;      // mask source (BN + residual + ReLU output) or null: mask recomputed from y
    const unsigned char* mask = nullptr;   // ... or the ReLU bit mask the forward saved (bit e of byte i: channel 8i+e of the flat NHWC tensor is > 0): 1/16 of z's bytes
    const float* mean = nullptr; const float* invstd = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
    float* slab = nullptr;
    int rows = 0;
};
// dx = conv^T(dy, w_bwd[Ci][wtaps][Co]) (+ res)
int conv_dgrad(hipStream_t s, const ConvGeom& g, const elem_t* dy, const elem_t* w_bwd, void* dx, const elem_t* res, int out_f32,
               DgradBnStat* bs = nullptr);
// dw (fp32, [Co][wtaps][Ci]; transposed: [Ci][wtaps][Co]) (+)= ...;  rows_valid < 0 -> all rows
int conv_wgrad(hipStream_t s, const ConvGeom& g, const elem_t* dy, const elem_t* x, float* dw, int accumulate, int rows_valid);
// the same problem as a parameter block (for the grouped launch); returns the layer's algorithmic FLOPs in *flops
int conv_wgrad_params(const ConvGeom& g, const elem_t* dy, const elem_t* x, float* dw, int rows_valid, WgParams* out, double* flops);
int conv_prof_before(hipStream_t s, int kind, double flops);
void conv_prof_after(hipStream_t s, int token);
int conv_stat_rows(const ConvGeom& g);
int conv_dgrad_stat_rows(const ConvGeom& g);   // rows of a DgradBnStat slab for this layer's dgrad
