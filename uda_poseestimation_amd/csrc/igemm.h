// Parameter block shared by the implicit-GEMM convolution kernels (fprop, dgrad, sub-pixel deconv) and wgrad.
#pragma once
#include "common.h"

// One filter tap of one output class: input offset (dy,dx) relative to (i*s, j*s) and the weight slab index.
struct IgTap { int dy, dx, widx, cls; };   // cls: owning output class (used by wgrad)
// An output class (sub-pixel phase): rows m=(n,i,j) over an Hg x Wg grid write output pixel (i*os+oa, j*os+ob).
struct IgClass { int tap_off, ntaps, oa, ob; };

#define IG_FLAG_REFLECT 1     // reflection padding instead of zero padding
#define IG_FLAG_UPSAMPLE 2    // input is read through a nearest x2 upsample (logical dims = 2x physical)
#define IG_FLAG_RELU 4
#define IG_FLAG_OUT_F32 8     // y is fp32 NHWC instead of bf16 NHWC
#define IG_FLAG_F32 128      // x, w, res and y are fp32 (exact fp32 MFMA path, forward only)
#define IG_FLAG_SPLIT 64     // with IG_FLAG_F32: x, w, res (and y, unless IG_FLAG_OUT_F32) are f16x2 split tensors (common.h): the
                             // fp32-grade mode, three fp16 MFMAs per 32-deep K step; forward only
#define IG_FLAG_MIRROR 2048   // 3x3 run-staged form: taps are the mirrored ones of a data gradient (dy, dx) = (1 - kh, 1 - kw)
#define IG_FLAG_TAP0 1024     // every class has at most the one tap (dy, dx, widx) = (0, 0, 0): no tap-table read
#define IG_FLAG_BSMASK 32     // bs_z points at the consumer BN's saved ReLU bit mask (one byte per 8 channels) instead of its output z
#define IG_FLAG_SMALLC 16     // Ci == 8: one 32-wide K step covers 4 taps (stem / first VGG conv)

struct IgParams {
    const elem_t* x;        // NHWC bf16 [N, Hi, Wi, Ci]
    const elem_t* w;        // [Co][wtaps][Ci] bf16 (K-contiguous per output channel)
    void* y;                // NHWC [N, Ho, Wo, Co] bf16 (or fp32)
    const elem_t* res;      // optional residual, same shape as y (bf16), added before ReLU
    const float* bias;      // optional [Co]
    const float* scale;     // optional [Co]: the fp32 result is multiplied by it before the bias (eval-mode BatchNorm folded into the conv:
                            // y = conv * gamma * invstd + (beta - mean * gamma * invstd), net.hip conv_bn_fwd)
    float* stats;           // optional per-channel partial sums: [stat_rows][2][Co] fp32 (sum, sumsq of fp32 acc)
    const IgTap* taps;      // device tap table
    int N, Hi, Wi, Ci;
    int Ho, Wo, Co;
    int Hg, Wg;             // row grid per class
    int s, os;              // input stride multiplier, output stride multiplier
    int M;                  // N*Hg*Wg rows per class
    int wtaps;              // taps per output channel in w (row length = wtaps*Ci)
    int flags;
    int nclass;
    IgClass cls[4];
    FastDiv div_hw, div_w;  // m / (Hg*Wg), rem / Wg
    int m_tiles, n_tiles;
    int tap0;               // host: set by the conv entry points when every tap of the plan is (0, 0, 0) (1x1 kernels)
    unsigned long long* dbg;   // tuning: per-work-group timeline stamps [blocks][8] (s_memrealtime, 100 MHz), normally null
    // dgrad only: the BatchNorm that consumes this launch's output dz in the backward chain.  When bs_y is set the epilogue
    // applies that BN's ReLU mask to dz (so y receives g = dz * mask), and writes per-m-tile partial sums of g and g * xhat to
    // `stats` ([stat_rows][2][Co], the same slab layout as the forward statistics): the separate reduce launch disappears.
    const elem_t* bs_y;        // the consumer BN's input (pre-BN conv output), same shape as y
    const elem_t* bs_z;        // mask source z > 0 (BN + residual + ReLU); null: mask recomputed from y*scale + shift > 0
    const float* bs_mean;      // [Co] saved batch mean / inverse std of the consumer BN
    const float* bs_invstd;
    const float* bs_gamma;     // [Co] (mask recomputation only)
    const float* bs_beta;
};

// wgrad: dW[r][tap][c] (fp32) (+)= sum_m P[pixP][r] * Q[pixQ][c]
struct WgParams {
    const elem_t* dy;       // NHWC [N, Ho, Wo, Co]   (gradient of the conv output)
    const elem_t* x;        // NHWC [N, Hi, Wi, Ci]   (conv input)
    float* dw;              // fp32, layout [R][wtaps][C] where (R,C) = (Co,Ci) or, with swap, (Ci,Co)
    const IgTap* taps;
    int N, Hi, Wi, Ci;
    int Ho, Wo, Co;
    int Hg, Wg;
    int s, os;
    int M;
    int wtaps;
    int flags;              // IG_FLAG_SMALLC ; WG_FLAG_SWAP ; WG_FLAG_ACCUM
    int nclass;
    IgClass cls[4];
    FastDiv div_hw, div_w;
    int ksplit;             // number of M splits (grid.z)
    int msteps_per_split;   // 32-row steps per split
    int r_tiles, c_tiles;
    int total_taps;
    int rows_valid;         // dW rows actually stored (<= R; head: Co is padded to 32 in dy)
    int kw;                 // grouped Ci == 8 form only: real taps per filter row (the 8th column chunk is padding)
    unsigned part_stride;   // != 0: split bz stores its partial tile with plain stores at element offset bz * part_stride from dw (deterministic
                            // split reductions of the grouped launches: pw_split_sum adds the splits in order); 0: direct stores / atomics
};
struct WgGroupBlk { int prob, local; };   // grouped wgrad: problem index (< 0: padding) and linear block index inside it
#define WG_FLAG_SWAP 32       // rows of dW come from x (deconv weight layout [Ci][tap][Co])
#define WG_FLAG_DW_WS 256     // grouped launch: dw is an offset into the workspace (dy base), not into the gradient buffer
#define WG_FLAG_FASTGEO 512   // loader: stride-1 same-size conv on power-of-two maps (bit-field pixel coordinates, 32-bit offsets)
#define WG_FLAG_FAST2 16384    // fast geometry, second loader form: buffer loads with out-of-range zero fill, unrolled ring (wgrad_fast2_body)
#define WG_FLAG_ROW3_OK 8192  // geometry: 3x3, stride 1, pad 1, plain conv (set by conv_wgrad_params)
#define WG_FLAG_ROW3 4096     // one work-group per (tile, filter ROW): the three taps of the row share one dy stage and one x window (wgrad_row3_body)
#define WG_FLAG_ATOMIC 64     // accumulate into dw with fp32 atomics (ksplit>1 or beta=1)
