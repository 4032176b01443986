// Host-side description of one convolution and the tap plans (fprop / dgrad / wgrad) the igemm kernels execute.
#pragma once
#include <vector>
#include "igemm.h"

struct ConvGeom {
    int N, Hi, Wi, Ci;      // logical input (for transposed: the small side)
    int Co, KH, KW, stride, pad;
    int transposed;         // 1 = ConvTranspose2d(k, stride, pad, output_padding = 0)
    int reflect;            // reflection padding (style net)
    int upsample;           // nearest x2 upsample folded into the loader (style decoder); Hi/Wi are the PHYSICAL dims
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

// direction 0: fprop (also the plan wgrad walks), 1: dgrad
const TapPlan* get_tap_plan(const ConvGeom& g, int direction);

int igemm_pick_tile(int M, int Co, int nclass, int K, int h3_ok = 0);   // h3_ok: 3x3 stride-1 pad-1 same-size conv (conv_h3_ok)
int conv_h3_ok(const ConvGeom& g);
int igemm_stat_rows(int M, int Co, int nclass, int tile);
int igemm_launch(IgParams& p, int tile, hipStream_t stream);
int wgrad_pick_tile(int Rdim, int Cdim, int smallc);
int wgrad_launch(WgParams& p, int tile, int accumulate, hipStream_t stream);
// Grouped wgrad (many layers, one launch per tile class).  wgrad_group_plan completes p for the group kernels and returns
// the tile class (0 = 128x128, 1 = 64x64) or < 0 when the layer needs its own launch; stages_per_block bounds a work-group's
// pixel range (longer reductions are split and accumulated with fp32 atomics into a zeroed dW).
int wgrad_group_plan(WgParams& p, int accumulate, int stages_per_block);
// the x / dy / dw fields of the table entries are byte offsets from the three bases
int wgrad_group_launch(hipStream_t stream, int tile, const WgParams* d_tab, const WgGroupBlk* d_blk, int per_xcd, const void* x_base,
                       const void* dy_base, void* dw_base);

struct ConvEpilogue {
    const elem_t* res = nullptr;
    const float* bias = nullptr;
    float* stats = nullptr;
    int relu = 0;
    int out_f32 = 0;
    int f32 = 0;            // x, w, res, y are fp32 (exact fp32 MFMA path; forward only)
};
// y = conv(x, w_fwd[Co][wtaps][Ci])
int conv_fprop(hipStream_t s, const ConvGeom& g, const elem_t* x, const elem_t* w_fwd, void* y, const ConvEpilogue& e);
// The BatchNorm whose backward consumes a dgrad's output (IgParams::bs_*): the dgrad epilogue masks dx with that BN's ReLU
// and leaves the partial sums of g and g * xhat in slab[rows][2][C]; `rows` is set by conv_dgrad.
struct DgradBnStat {
    const elem_t* y = nullptr;      // the BN's input (pre-BN conv output)
    const elem_t* z = nullptr;      // mask source (BN + residual + ReLU output) or null: mask recomputed from y
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
