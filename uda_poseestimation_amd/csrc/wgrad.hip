// Weight-gradient implicit GEMM for gfx950: dW[r][tap][c] (fp32) = sum over output rows m of P[pix_P(m)][r] * Q[pix_Q(m)][c].
//
// The reduction runs over pixels, which is the NON-contiguous dimension of both NHWC operands, so both MFMA operands
// need a transpose.  Tiles are staged [32 pixels][channels] in LDS (coalesced 16-byte global loads along channels) and
// read back with the gfx950 transposing LDS read ds_read_b64_tr_b16, which hands each lane 4 consecutive k (pixels) of
// one channel: two reads give the 8-deep fragment of MFMA 16x16x32.  The LDS row stride is channels*2 + 32 bytes so that
// the 8 pixel rows touched by one 32-lane half fall on 8 different 32-byte bank slots (conflict-free).
// Both operands use the same pixel<->k permutation (k = 8g+4h+q  <->  LDS row 16h+4g+q), so the sum is unchanged.
//
// grid = (r_tiles*c_tiles, taps (or tap groups of 4 when Ci==8), ksplit).  ksplit>1 or accumulate -> fp32 atomics.
#include <stdlib.h>
#include "conv_plan.h"

namespace {

template <int RT, int CT, int WR, int WC>
struct WgCfg {
    static constexpr int TR = RT / WR, TC = CT / WC;
    static constexpr int MT = TR / 16, NT = TC / 16;
    static constexpr int PSTR = RT * 2 + 32, QSTR = CT * 2 + 32;     // LDS row strides (bytes)
    static constexpr int P_BYTES = 32 * PSTR, Q_BYTES = 32 * QSTR;
    static constexpr int P_CH = (32 * RT / 8 + 255) / 256, Q_CH = (32 * CT / 8 + 255) / 256;
    static constexpr int LDS_BYTES = 2 * (P_BYTES + Q_BYTES);
};

template <int RT, int CT, int WR, int WC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgParams p) {
    using C = WgCfg<RT, CT, WR, WC>;
    constexpr int TR = C::TR, TC = C::TC, MT = C::MT, NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const bool smallc = (p.flags & IG_FLAG_SMALLC) != 0;
    const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
    const int c_tile = blockIdx.x % p.c_tiles, r_tile = blockIdx.x / p.c_tiles;
    const int r0 = r_tile * RT, c0 = c_tile * CT;

    // operand roles: P supplies dW rows, Q supplies dW columns
    const int Rdim = swap ? p.Ci : p.Co;
    const int Cdim = swap ? p.Co : p.Ci;
    const bool p_is_x = swap;

    const int tap_base = smallc ? blockIdx.y * 4 : blockIdx.y;
    IgTap tp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tp[i] = p.taps[tap_base + ((smallc) ? i : 0)];
    const IgClass cls = p.cls[tp[0].cls];

    const int ms0 = blockIdx.z * p.msteps_per_split;
    int ms1 = ms0 + p.msteps_per_split;
    const int ms_total = (p.M + 31) >> 5;
    if (ms1 > ms_total) ms1 = ms_total;

    constexpr int PCPR = RT / 8, QCPR = CT / 8;
    u32x4 rp[C::P_CH], rq[C::Q_CH];

    auto load_op = [&](bool is_x, int row, int chn, int m, const IgTap& t, int dim_base, int dim_lim) -> u32x4 {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m >= p.M || chn >= dim_lim) return v;
        const uint32_t n = fdiv((uint32_t)m, p.div_hw);
        const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Hg * p.Wg);
        const uint32_t ii = fdiv(rem, p.div_w);
        const uint32_t jj = rem - ii * (uint32_t)p.Wg;
        if (is_x) {
            const int hi = (int)ii * p.s + t.dy, wi = (int)jj * p.s + t.dx;
            if ((unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi)
                v = *(const u32x4*)(p.x + (((size_t)n * p.Hi + hi) * p.Wi + wi) * p.Ci + chn);
        } else {
            const size_t opix = ((size_t)n * p.Ho + (ii * p.os + cls.oa)) * p.Wo + (jj * p.os + cls.ob);
            v = *(const u32x4*)(p.dy + opix * p.Co + chn);
        }
        return v;
    };

    auto issue_loads = [&](int ms) {
        const int mb = ms << 5;
#pragma unroll
        for (int i = 0; i < C::P_CH; ++i) {
            const int q = tid + 256 * i;
            const int row = q / PCPR, cc = q % PCPR;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < 32) {
                if (p_is_x && smallc) v = load_op(true, row, 0, mb + row, tp[cc & 3], 0, 8);
                else v = load_op(p_is_x, row, r0 + cc * 8, mb + row, tp[0], r0, Rdim);
            }
            rp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < C::Q_CH; ++i) {
            const int q = tid + 256 * i;
            const int row = q / QCPR, cc = q % QCPR;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < 32) {
                if (!p_is_x && smallc) v = load_op(true, row, 0, mb + row, tp[cc & 3], 0, 8);
                else v = load_op(!p_is_x, row, c0 + cc * 8, mb + row, tp[0], c0, Cdim);
            }
            rq[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
        char* P = smem + buf * (C::P_BYTES + C::Q_BYTES);
        char* Q = P + C::P_BYTES;
#pragma unroll
        for (int i = 0; i < C::P_CH; ++i) {
            const int q = tid + 256 * i;
            const int row = q / PCPR, cc = q % PCPR;
            if (row < 32) *(u32x4*)(P + row * C::PSTR + cc * 16) = rp[i];
        }
#pragma unroll
        for (int i = 0; i < C::Q_CH; ++i) {
            const int q = tid + 256 * i;
            const int row = q / QCPR, cc = q % QCPR;
            if (row < 32) *(u32x4*)(Q + row * C::QSTR + cc * 16) = rq[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposing-read lane addressing: group g = lane>>4 owns k block g; lane 4q+pp of the group supplies the
    // address of block row q, columns 4pp..4pp+3; read h covers LDS rows 16h + 4g + q.
    const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int trow = 4 * g + qq;

    if (ms0 < ms1) {
        issue_loads(ms0);
        store_lds(0);
    }
    __syncthreads();
    for (int ms = ms0; ms < ms1; ++ms) {
        const int buf = (ms - ms0) & 1;
        if (ms + 1 < ms1) issue_loads(ms + 1);
        const char* P = smem + buf * (C::P_BYTES + C::Q_BYTES);
        const char* Q = P + C::P_BYTES;
        elem8 af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int colb = (wr * TR + i * 16 + 4 * pp) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, P + trow * C::PSTR + colb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, P + (16 + trow) * C::PSTR + colb));
            union { struct { s16x4 a, b; } s; elem8 v; } u;
            u.s.a = lo; u.s.b = hi;
            af[i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int colb = (wc * TC + j * 16 + 4 * pp) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, Q + trow * C::QSTR + colb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, Q + (16 + trow) * C::QSTR + colb));
            union { struct { s16x4 a, b; } s; elem8 v; } u;
            u.s.a = lo; u.s.b = hi;
            bfr[j] = u.v;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = UDAPOSE_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
        if (ms + 1 < ms1) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row = r][col = c]; lane holds col = lane&15, rows (lane>>4)*4 + reg
    const bool atomic = (p.flags & WG_FLAG_ATOMIC) != 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + wr * TR + i * 16 + (lane >> 4) * 4 + r;
                const int ccl = wc * TC + j * 16 + (lane & 15);      // column inside the tile
                if (rr >= Rdim || rr >= p.rows_valid) continue;
                size_t off;
                if (smallc) {
                    // columns (or rows) of the Ci==8 operand enumerate 4 taps x 8 channels
                    if (!swap) {
                        const IgTap& t = tp[(ccl >> 3) & 3];
                        if (ccl >= 32) continue;
                        off = ((size_t)rr * p.wtaps + t.widx) * 8 + (ccl & 7);
                    } else {
                        continue;   // not used: no transposed conv with 8 input channels
                    }
                } else {
                    const int cc = c0 + ccl;
                    if (cc >= Cdim) continue;
                    off = ((size_t)rr * p.wtaps + tp[0].widx) * Cdim + cc;
                }
                const float v = acc[i][j][r];
                if (atomic) atomicAdd(p.dw + off, v);
                else p.dw[off] = v;
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA version (tiles 128x128 and 64x64, Ci >= 64): NS-deep ring of [64 pixels][channels] stages filled by
// global_load_lds_dwordx4 (no staging registers), counted vmcnt + one raw s_barrier per stage.  The DMA writes 1 KiB
// contiguously, so rows cannot be padded: the 16-byte chunk index is XOR-swizzled on the SOURCE side instead
// (chunk ^ ((row>>1)&3)<<1 for 128-byte rows, chunk ^ (row&7)<<1 for 256-byte rows), which makes the transposing reads
// of one 32-lane half (8 pixel rows x 32 bytes) hit 16 distinct 16-byte bank slots.
__device__ u32x4 g_wzero16[2];

template <int N> __device__ __forceinline__ void wg_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

template <int CH> __device__ __forceinline__ int wswz(int row) {   // CH = channels per LDS row (64 or 128)
    return CH == 64 ? (((row >> 1) & 3) << 1) : ((row & 7) << 1);
}

template <int RT, int CT, int WR, int WC, int NS, int PX = 64>   // PX = pixels (K rows) per LDS stage: 64 or 32
struct WdCfg {
    static constexpr int TR = RT / WR, TC = CT / WC;
    static constexpr int MT = TR / 16, NT = TC / 16;
    static constexpr int PROW = RT * 2, QROW = CT * 2;               // bytes per LDS row
    static constexpr int NW = WR * WC;                                // waves per work-group (4; 8 in the 256x256 tile)
    static constexpr int P_BYTES = PX * PROW, Q_BYTES = PX * QROW;
    static constexpr int P_PW = P_BYTES / (1024 * NW), Q_PW = Q_BYTES / (1024 * NW); // DMA instructions per wave per stage
    static constexpr int STAGE1 = P_BYTES + Q_BYTES;
    static constexpr int LDS_BYTES = NS * STAGE1;
};

// One work-group: tile bx of dW tap `by`, pixel split bz of problem p.
// Split reductions (ksplit > 1) come in two forms: fp32 atomics into a zeroed dW (WG_FLAG_ATOMIC; the per-layer launches), or - the grouped
// launches' form, round 6 - PARTIAL TILES: gp.part_stride != 0, dw points into the pass's workspace and split bz stores its tile with plain
// stores at element offset bz * part_stride; pw_split_sum adds the splits in split order afterwards (bit-reproducible).
template <int RT, int CT, int WR, int WC, int NS, int PX = 64, bool FAST = false>
__device__ __forceinline__ void wgrad_dma_body(const WgParams& gp, const uint32_t bx, const uint32_t by, const uint32_t bz, char* smem,
                                               const uintptr_t x_base = 0, const uintptr_t dy_base = 0, const uintptr_t dw_base = 0) {
    // scalar copies of the fields used below (gp may live in global memory: read it once, up front, into SGPRs)
    struct {
        const elem_t* dy; const elem_t* x; float* dw; const IgTap* taps;
        int Hi, Wi, Ci, Ho, Wo, Co, Hg, Wg, s, os, M, wtaps, flags, ksplit, c_tiles, rows_valid, kw;
        FastDiv div_hw, div_w;
    } p = {(const elem_t*)((uintptr_t)gp.dy + dy_base), (const elem_t*)((uintptr_t)gp.x + x_base), (float*)((uintptr_t)gp.dw + ((gp.flags & WG_FLAG_DW_WS) ? dy_base : dw_base)), gp.taps, gp.Hi, gp.Wi, gp.Ci, gp.Ho, gp.Wo, gp.Co, gp.Hg, gp.Wg, gp.s, gp.os, gp.M, gp.wtaps, gp.flags, gp.ksplit,
           gp.c_tiles, gp.rows_valid, gp.kw, gp.div_hw, gp.div_w};
    using C = WdCfg<RT, CT, WR, WC, NS, PX>;
    constexpr int TR = C::TR, TC = C::TC, MT = C::MT, NT = C::NT, P_PW = C::P_PW, Q_PW = C::Q_PW;
    constexpr int LPS = P_PW + Q_PW;
    constexpr int P_RPI = 1024 / C::PROW, Q_RPI = 1024 / C::QROW;    // rows per DMA instruction (8 or 4)
    constexpr int P_CPR = C::PROW / 16, Q_CPR = C::QROW / 16;        // 16-byte chunks per row (8 or 16)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
    const int c_tile = bx % p.c_tiles, r_tile = bx / p.c_tiles;
    const int r0 = r_tile * RT, c0 = c_tile * CT;
    // grouped Ci == 8 form (the 3-channel stem): one tap of the table is a filter ROW; its 8 column taps x 8 channels are the
    // 64 "channels" of the Q operand (contiguous in x, since C == 8), validity is per 16-byte chunk = per column tap
    const bool rowtap = (p.flags & IG_FLAG_SMALLC) != 0;
    const int Rdim = swap ? p.Ci : p.Co, Cdim = rowtap ? 64 : (swap ? p.Co : p.Ci);
    const bool p_is_x = swap;
    const IgTap tp = p.taps[by];
    struct { int oa, ob; } cls = {gp.cls[0].oa, gp.cls[0].ob};       // (select chain: no dynamic index into the parameter block)
    if (tp.cls == 1) { cls.oa = gp.cls[1].oa; cls.ob = gp.cls[1].ob; }
    if (tp.cls == 2) { cls.oa = gp.cls[2].oa; cls.ob = gp.cls[2].ob; }
    if (tp.cls == 3) { cls.oa = gp.cls[3].oa; cls.ob = gp.cls[3].ob; }
    const int ms_total = (p.M + PX - 1) / PX;                        // PX-pixel stages
    const int per = (ms_total + p.ksplit - 1) / p.ksplit;
    const int ms0 = bz * per;
    int ms1 = ms0 + per;
    if (ms1 > ms_total) ms1 = ms_total;
    const int nsteps = ms1 > ms0 ? ms1 - ms0 : 0;
    const size_t part_off = (size_t)bz * gp.part_stride;
    const char* zsrc = (const char*)g_wzero16;

    // Address generation is the issue-slot hog of this loop (measured 24 VALU instructions per MFMA when every row was
    // decomposed into (n,i,j) twice per stage), so: P and Q rows of a lane are the same pixels (RT == CT), operands
    // whose pixel index is linear in m skip the decomposition entirely (dy of every plain conv; x of 1x1 stride-1 convs).
    static_assert(RT == CT, "P and Q share their row mapping");
    const bool dy_lin = p.os == 1 && cls.oa == 0 && cls.ob == 0 && p.Hg == p.Ho && p.Wg == p.Wo;
    const bool x_lin = p.s == 1 && tp.dy == 0 && tp.dx == 0 && p.Hg == p.Hi && p.Wg == p.Wi;
    const int lrow = lane / P_CPR, pch = lane % P_CPR;
    // Fast geometry (every stride-1 convolution of the 256x256 networks: 1x1 and 3x3 pad 1 on power-of-two maps): the x pixel of
    // row m and tap (dy, dx) is m + dy*W + dx, (i, j) are bit fields of m, and element offsets fit 32 bits with 24-bit factors -
    // a stage's four source addresses cost ~35 vector instructions instead of ~200 (64-bit multiplies, two exact divisions per
    // row): the loop has 16 MFMAs (256 cycles) per wave and stage and was issue-bound on its address arithmetic (18 % MFMA use).
    const bool pow2hw = ((p.Hi & (p.Hi - 1)) | (p.Wi & (p.Wi - 1))) == 0;
    // (FAST: a separate instantiation of the body, chosen per problem from WG_FLAG_FASTGEO - set by wg_fastgeo_ok - so that the two
    // loaders do not add up their registers: the kernel sits at its 128-register budget)
    (void)pow2hw;
    const int toff = tp.dy * p.Wi + tp.dx;
    const bool center = tp.dy == 0 && tp.dx == 0;
    const unsigned w_mask = (unsigned)p.Wi - 1u, hw_mask = (unsigned)(p.Hi * p.Wi) - 1u;
    const int lgw = 31 - __builtin_clz((unsigned)(p.Wi > 0 ? p.Wi : 1));
    auto issue_stage = [&](int st, int buf) {
        const int mb = (ms0 + st) * PX;
        char* P = smem + buf * C::STAGE1;
        char* Q = P + C::P_BYTES;
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < P_PW; ++i) {
                const int row = (i * 4 + wid) * P_RPI + lrow;
                const int lc = pch ^ wswz<RT>(row);
                const int m = mb + row;
                const bool okd = m < p.M;
                bool okx = okd;
                if (!center) {
                    const unsigned rem = (unsigned)m & hw_mask;
                    const int ii = (int)(rem >> lgw), jj = (int)(rem & w_mask);
                    okx = okd && (unsigned)(ii + tp.dy) < (unsigned)p.Hi && (unsigned)(jj + tp.dx) < (unsigned)p.Wi;
                }
                const int pc = r0 + lc * 8, qc = c0 + lc * 8;
                const unsigned od = __umul24((unsigned)m, (unsigned)p.Co) + (unsigned)pc;            // P rows come from dy (no swap)
                const unsigned ox = __umul24((unsigned)(m + toff), (unsigned)p.Ci) + (unsigned)qc;   // Q rows from x
                const char* sp = (okd && pc < Rdim) ? (const char*)p.dy + (size_t)od * 2 : zsrc;
                const char* sq = (okx && qc < Cdim) ? (const char*)p.x + (size_t)ox * 2 : zsrc;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                                 (__attribute__((address_space(3))) void*)(P + (i * 4 + wid) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq,
                                                 (__attribute__((address_space(3))) void*)(Q + (i * 4 + wid) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
        for (int i = 0; i < P_PW; ++i) {
            const int row = (i * 4 + wid) * P_RPI + lrow;
            const int lc = pch ^ wswz<RT>(row);
            const int m = mb + row;
            const bool mok = m < p.M;
            const char* sx = zsrc;
            const char* sd = zsrc;
            if (mok) {
                uint32_t n = 0, ii = 0, jj = 0;
                if (!(dy_lin && x_lin)) {
                    n = fdiv((uint32_t)m, p.div_hw);
                    const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Hg * p.Wg);
                    ii = fdiv(rem, p.div_w);
                    jj = rem - ii * (uint32_t)p.Wg;
                }
                if (x_lin) sx = (const char*)(p.x + (size_t)m * p.Ci);
                else {
                    const int hi = (int)ii * p.s + tp.dy, wi = (int)jj * p.s + tp.dx;
                    const int wl = rowtap ? wi + lc : wi;            // this lane's chunk is column tap lc
                    if ((unsigned)hi < (unsigned)p.Hi && (unsigned)wl < (unsigned)p.Wi && (!rowtap || lc < p.kw))
                        sx = (const char*)(p.x + (((long long)n * p.Hi + hi) * p.Wi + wi) * p.Ci);   // (+ lc * 16 bytes below)
                }
                if (dy_lin) sd = (const char*)(p.dy + (size_t)m * p.Co);
                else {
                    const uint32_t oh = ii * p.os + cls.oa, ow = jj * p.os + cls.ob;
                    if (oh < (uint32_t)p.Ho && ow < (uint32_t)p.Wo) sd = (const char*)(p.dy + (((size_t)n * p.Ho + oh) * p.Wo + ow) * p.Co);
                }
            }
            const char* sp = p_is_x ? sx : sd;
            const char* sq = p_is_x ? sd : sx;
            const int pc = r0 + lc * 8, qc = c0 + lc * 8;
            sp = (sp != zsrc && pc < Rdim) ? sp + (size_t)pc * 2 : zsrc;
            sq = (sq != zsrc && qc < Cdim) ? sq + (size_t)qc * 2 : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(P + (i * 4 + wid) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq,
                                             (__attribute__((address_space(3))) void*)(Q + (i * 4 + wid) * 1024), 16, 0, 0);
        }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int issued = 0;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nsteps) { issue_stage(st, st); ++issued; }

    const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st % NS;
        if (issued - st - 1 >= NS - 2) wg_wait_vmcnt<LPS*(NS - 2)>();
        else wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (issued < nsteps) { issue_stage(issued, issued % NS); ++issued; }
        const char* P = smem + buf * C::STAGE1;
        const char* Q = P + C::P_BYTES;
#pragma unroll
        for (int kk = 0; kk < PX / 32; ++kk) {
            elem8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int col = wr * TR + i * 16 + 4 * pp;           // first of this lane's 4 columns
                union { struct { s16x4 a, b; } s; elem8 v; } u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 32 * kk + 16 * h + 4 * g + qq;
                    const int off = row * C::PROW + ((((col >> 3) ^ wswz<RT>(row)) << 4) | ((col & 4) << 1));
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, P + off));
                    if (h == 0) u.s.a = v; else u.s.b = v;
                }
                af[i] = u.v;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = wc * TC + j * 16 + 4 * pp;
                union { struct { s16x4 a, b; } s; elem8 v; } u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 32 * kk + 16 * h + 4 * g + qq;
                    const int off = row * C::QROW + ((((col >> 3) ^ wswz<CT>(row)) << 4) | ((col & 4) << 1));
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, Q + off));
                    if (h == 0) u.s.a = v; else u.s.b = v;
                }
                bfr[j] = u.v;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = UDAPOSE_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
        }
    }

    const bool atomic = (p.flags & WG_FLAG_ATOMIC) != 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + wr * TR + i * 16 + (lane >> 4) * 4 + r;
                const int cc = c0 + wc * TC + j * 16 + (lane & 15);
                if (rr >= Rdim || rr >= p.rows_valid || cc >= Cdim) continue;
                const size_t off = ((size_t)rr * p.wtaps + tp.widx) * Cdim + cc + part_off;
                const float v = acc[i][j][r];
                if (atomic) atomicAdd(p.dw + off, v);
                else p.dw[off] = v;
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast-geometry body, second form (WG_FLAG_FAST2): the same tiles, ring and MFMA loop as wgrad_dma_body<.., FAST = true>, with the
// per-stage instruction stream cut down.  That loop issues 63 vector + 58 scalar instructions per 16 MFMAs and is bound by
// instruction issue (rocprofv3: 3.9 non-MFMA vector instructions per MFMA, 31 % MFMA-busy with the chip to itself):
//   * operands are fetched with buffer_load_dwordx4 ... lds: a lane's byte offset inside a stage is a CONSTANT, the stage's
//     position is the scalar offset, and a lane whose tap falls outside the image gets an out-of-range offset, for which the
//     hardware writes zeros into LDS (tools/probe/buf_lds.hip) - no 64-bit address arithmetic, no pointer selects, no zero page;
//   * the ring is unrolled over its NS buffers, so LDS destinations (M0) and fragment read offsets are compile-time constants;
//   * pixel counts are multiples of the stage (power-of-two maps), so there is no row-validity test at all for centre taps.
template <int U> struct WIC { static constexpr int value = U; };
template <int N, typename F> __device__ __forceinline__ void w_static_for(F&& f) {
    if constexpr (N > 0) { w_static_for<N - 1>(f); f(WIC<N - 1>{}); }
}
template <int RT, int CT, int WR, int WC, int NS, int PX>
__device__ __forceinline__ void wgrad_fast2_body(const WgParams& gp, const uint32_t bx, const uint32_t by, const uint32_t bz, char* smem,
                                                 const uintptr_t x_base = 0, const uintptr_t dy_base = 0, const uintptr_t dw_base = 0) {
    struct {
        const elem_t* dy; const elem_t* x; float* dw; const IgTap* taps;
        int Hi, Wi, Ci, Co, M, wtaps, flags, ksplit, c_tiles, rows_valid;
    } p = {(const elem_t*)((uintptr_t)gp.dy + dy_base), (const elem_t*)((uintptr_t)gp.x + x_base),
           (float*)((uintptr_t)gp.dw + ((gp.flags & WG_FLAG_DW_WS) ? dy_base : dw_base)), gp.taps, gp.Hi, gp.Wi, gp.Ci, gp.Co, gp.M, gp.wtaps,
           gp.flags, gp.ksplit, gp.c_tiles, gp.rows_valid};
    using C = WdCfg<RT, CT, WR, WC, NS, PX>;
    constexpr int TR = C::TR, TC = C::TC, MT = C::MT, NT = C::NT, P_PW = C::P_PW, Q_PW = C::Q_PW;
    constexpr int LPS = P_PW + Q_PW, NW = C::NW;
    // (rows per 1 KiB DMA piece and 16-byte chunks per row, for the dy side P and the x side Q: RT != CT in the 128x256 / 256x128 tiles)
    constexpr int P_RPI = 1024 / C::PROW, P_CPR = C::PROW / 16, Q_RPI = 1024 / C::QROW, Q_CPR = C::QROW / 16;
    constexpr int OOB = 0x7fffffff;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WC, wc = wid % WC;
    const int c_tile = bx % p.c_tiles, r_tile = bx / p.c_tiles;
    const int r0 = r_tile * RT, c0 = c_tile * CT;
    // (the tap is wave-uniform; said explicitly, so that the x buffer descriptor built from it lives in scalar registers - a
    // descriptor the compiler cannot prove uniform costs a readfirstlane waterfall loop per load)
    const IgTap tp0 = p.taps[by];
    struct { int dy, dx, widx; } tp = {__builtin_amdgcn_readfirstlane(tp0.dy), __builtin_amdgcn_readfirstlane(tp0.dx),
                                       __builtin_amdgcn_readfirstlane(tp0.widx)};
    const int toff = tp.dy * p.Wi + tp.dx;
    const bool center = tp.dy == 0 && tp.dx == 0;
    const int ms_total = p.M / PX;                                    // (M % PX == 0: checked on the host)
    const int per = (ms_total + p.ksplit - 1) / p.ksplit;
    const int ms0 = bz * per;
    int ms1 = ms0 + per;
    if (ms1 > ms_total) ms1 = ms_total;
    const int nsteps = ms1 > ms0 ? ms1 - ms0 : 0;
    const size_t part_off = (size_t)bz * gp.part_stride;
    const unsigned w_mask = (unsigned)p.Wi - 1u, hw_mask = (unsigned)(p.Hi * p.Wi) - 1u;
    const int lgw = 31 - __builtin_clz((unsigned)p.Wi);
    // raw buffers over the two tensors; the x buffer starts at the tap's pixel offset (possibly in front of the tensor: every pixel
    // that would be read from outside the tensor is a tap outside the image and gets the out-of-range offset instead)
    __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.M * p.Co * 2, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (long long)toff * p.Ci * 2), 0, p.M * p.Ci * 2, 0x00020000);
    int vd[P_PW], vx[Q_PW], rowl[Q_PW];
#pragma unroll
    for (int i = 0; i < P_PW; ++i) {
        const int row = (i * NW + wid) * P_RPI + lane / P_CPR;
        const int lc = (lane % P_CPR) ^ wswz<RT>(row);
        vd[i] = (r0 + lc * 8 < p.Co) ? (row * p.Co + r0 + lc * 8) * 2 : OOB;
    }
#pragma unroll
    for (int i = 0; i < Q_PW; ++i) {
        const int row = (i * NW + wid) * Q_RPI + lane / Q_CPR;
        const int lc = (lane % Q_CPR) ^ wswz<CT>(row);
        rowl[i] = row;
        vx[i] = (c0 + lc * 8 < p.Ci) ? (row * p.Ci + c0 + lc * 8) * 2 : OOB;
    }
    const int sd_step = PX * p.Co * 2, sx_step = PX * p.Ci * 2;

    auto issue_stage = [&](int st, auto ub) __attribute__((always_inline)) {
        constexpr int UB = decltype(ub)::value;
        const int mb = (ms0 + st) * PX;
        const int sd = (ms0 + st) * sd_step, sx = (ms0 + st) * sx_step;      // scalar byte offsets of the stage
        char* P = smem + UB * C::STAGE1;
        char* Q = P + C::P_BYTES;
#pragma unroll
        for (int i = 0; i < (P_PW > Q_PW ? P_PW : Q_PW); ++i) {
            if (i < P_PW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (__attribute__((address_space(3))) void*)(P + (i * NW + wid) * 1024), 16, vd[i < P_PW ? i : 0], sd, 0, 0);
            if (i < Q_PW) {
                int ox = vx[i < Q_PW ? i : 0];
                if (!center) {
                    const unsigned rem = (unsigned)(mb + rowl[i < Q_PW ? i : 0]) & hw_mask;
                    const int ii = (int)(rem >> lgw), jj = (int)(rem & w_mask);
                    const bool ok = (unsigned)(ii + tp.dy) < (unsigned)p.Hi && (unsigned)(jj + tp.dx) < (unsigned)p.Wi;
                    ox = ok ? ox : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(Q + (i * NW + wid) * 1024), 16, ox, sx, 0, 0);
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int issued = 0;
    w_static_for<NS - 1>([&](auto u) {
        constexpr int U = decltype(u)::value;
        if (U < nsteps) { issue_stage(U, u); ++issued; }
    });

    const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    // lane part of the fragment read offsets (the stage buffer, kk, h and the fragment index are compile-time constants)
    int a_lo[MT][2], b_lo[NT][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 16 * h + 4 * g + qq;                        // (+ 32 * kk: the swizzle only looks at row bits 0..2, PROW * 32 is added below)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int col = wr * TR + i * 16 + 4 * pp;
            a_lo[i][h] = row * C::PROW + ((((col >> 3) ^ wswz<RT>(row)) << 4) | ((col & 4) << 1));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = wc * TC + j * 16 + 4 * pp;
            b_lo[j][h] = C::P_BYTES + row * C::QROW + ((((col >> 3) ^ wswz<CT>(row)) << 4) | ((col & 4) << 1));
        }
    }

    auto compute = [&](auto ub) __attribute__((always_inline)) {
        constexpr int UB = decltype(ub)::value;
        const char* S = smem + UB * C::STAGE1;
#pragma unroll
        for (int kk = 0; kk < PX / 32; ++kk) {
            elem8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                union { struct { s16x4 a, b; } s; elem8 v; } u;
                u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, S + a_lo[i][0] + kk * 32 * C::PROW));
                u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, S + a_lo[i][1] + kk * 32 * C::PROW));
                af[i] = u.v;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                union { struct { s16x4 a, b; } s; elem8 v; } u;
                u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, S + b_lo[j][0] + kk * 32 * C::QROW));
                u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, S + b_lo[j][1] + kk * 32 * C::QROW));
                bfr[j] = u.v;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = UDAPOSE_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
        }
    };

    for (int st0 = 0; st0 < nsteps; st0 += NS) {
        w_static_for<NS>([&](auto u) {
            constexpr int U = decltype(u)::value;
            const int st = st0 + U;
            if (st < nsteps) {
                if (issued - st - 1 >= NS - 2) wg_wait_vmcnt<LPS*(NS - 2)>();
                else wg_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (issued < nsteps) { issue_stage(issued, WIC<(U + NS - 1) % NS>{}); ++issued; }
                compute(u);
            }
        });
    }

    const bool atomic = (p.flags & WG_FLAG_ATOMIC) != 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + wr * TR + i * 16 + (lane >> 4) * 4 + r;
                const int cc = c0 + wc * TC + j * 16 + (lane & 15);
                if (rr >= p.Co || rr >= p.rows_valid || cc >= p.Ci) continue;
                const size_t off = ((size_t)rr * p.wtaps + tp.widx) * p.Ci + cc + part_off;
                const float v = acc[i][j][r];
                if (atomic) atomicAdd(p.dw + off, v);
                else p.dw[off] = v;
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// Filter-ROW form for 3x3, stride-1, pad-1 convolutions on power-of-two maps (W <= 64): one work-group computes the 64 x 64 tile
// of dW for the THREE taps (dy, -1), (dy, 0), (dy, +1) of one filter row.  The three taps read the same dy pixels and x pixels that
// differ by one column, so a stage holds ONE dy tile [64 pixels][64 channels] and ONE x window [64 + 2 pixels (padded to 72)][64
// channels] starting one pixel early; tap dx reads the window at row offset dx + 1.  LDS fill per FLOP is a third of the
// one-tap form at the same tile size.  Measured (tools/exp_wgrad_row3.py): as fast as the 128x128 one-tap form, not faster - what
// bounds these kernels is LDS bandwidth as a whole (per work-group and 64-pixel stage 17 KB of DMA writes + 64 KB of transposing
// reads; 48 KB for the 128x128 one-tap form).  In the grouped launch (64x64 kernel, 2-stage ring, four work-groups per CU) it is worth
// 1.5 % of the weight-gradient time and -0.10 ms per step.  Selected by Policy::wgrad_row3 (on by default).
// Zero padding: a pixel whose row i + dy falls outside the image gets a zero dy row (it contributes to none of the three taps);
// the column wrap (j = 0 with dx = -1, j = W-1 with dx = +1: the neighbour in memory belongs to another image row) is cut out of
// the dy FRAGMENTS per tap with loop-invariant lane masks (stages are 64 pixels, W divides 64: a lane's pixels keep their j).
struct Row3Cfg {
    static constexpr int PX = 64, P_BYTES = PX * 128, QR = PX + 8, Q_BYTES = QR * 128, STAGE1 = P_BYTES + Q_BYTES, DUMP = 1024;
    static constexpr int lds_bytes(int ns) { return ns * STAGE1 + DUMP; }
};
template <int NS>
__device__ __forceinline__ void wgrad_row3_body(const WgParams& gp, const uint32_t bx, const uint32_t by, const uint32_t bz, char* smem,
                                                const uintptr_t x_base = 0, const uintptr_t dy_base = 0, const uintptr_t dw_base = 0) {
    using R = Row3Cfg;
    constexpr int PX = R::PX, LPS = 5;                                 // 2 dy pieces + 3 x-window pieces per wave and stage
    struct {
        const elem_t* dy; const elem_t* x; float* dw;
        int Hi, Wi, Ci, Co, M, wtaps, flags, ksplit, c_tiles, rows_valid;
    } p = {(const elem_t*)((uintptr_t)gp.dy + dy_base), (const elem_t*)((uintptr_t)gp.x + x_base),
           (float*)((uintptr_t)gp.dw + ((gp.flags & WG_FLAG_DW_WS) ? dy_base : dw_base)), gp.Hi, gp.Wi, gp.Ci, gp.Co, gp.M, gp.wtaps, gp.flags,
           gp.ksplit, gp.c_tiles, gp.rows_valid};
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int c_tile = bx % p.c_tiles, r_tile = bx / p.c_tiles;
    const int r0 = r_tile * 64, c0 = c_tile * 64;
    const int dyk = (int)by - 1;                                       // this work-group's filter row: dy = -1, 0, +1
    const int ms_total = (p.M + PX - 1) / PX;
    const int per = (ms_total + p.ksplit - 1) / p.ksplit;
    const int ms0 = bz * per;
    int ms1 = ms0 + per;
    if (ms1 > ms_total) ms1 = ms_total;
    const int nsteps = ms1 > ms0 ? ms1 - ms0 : 0;
    const size_t part_off = (size_t)bz * gp.part_stride;
    const char* zsrc = (const char*)g_wzero16;
    char* const dump = smem + NS * R::STAGE1;
    const unsigned w_mask = (unsigned)p.Wi - 1u, hw_mask = (unsigned)(p.Hi * p.Wi) - 1u;
    const int lgw = 31 - __builtin_clz((unsigned)p.Wi);
    const int lrow = lane >> 3, pch = lane & 7;
    const int xoff0 = dyk * p.Wi - 1;                                  // x pixel of window row 0 relative to the stage's first pixel

    auto issue_stage = [&](int st, int buf) {
        const int mb = (ms0 + st) * PX;
        char* P = smem + buf * R::STAGE1;
        char* Q = P + R::P_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 4 + wid) * 8 + lrow;
            const int lc = pch ^ wswz<64>(row);
            const int m = mb + row;
            const bool ok = m < p.M && (unsigned)((int)(((unsigned)m & hw_mask) >> lgw) + dyk) < (unsigned)p.Hi;
            const unsigned od = __umul24((unsigned)m, (unsigned)p.Co) + (unsigned)(r0 + lc * 8);
            const char* sp = ok ? (const char*)p.dy + (size_t)od * 2 : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(P + (i * 4 + wid) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = i * 4 + wid;                                  // wave-uniform piece index; pieces 9..11 do not exist
            const int row = k * 8 + lrow;
            const int lc = pch ^ wswz<64>(row);
            const int xp = mb + xoff0 + row;
            const bool ok = k < 9 && (unsigned)xp < (unsigned)p.M;
            const unsigned ox = __umul24((unsigned)(ok ? xp : 0), (unsigned)p.Ci) + (unsigned)(c0 + lc * 8);
            const char* sq = ok ? (const char*)p.x + (size_t)ox * 2 : zsrc;
            char* dst = k < 9 ? Q + k * 1024 : dump;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x4 acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int issued = 0;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nsteps) { issue_stage(st, st); ++issued; }

    const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    // fragment element e of a lane is pixel row 32*kk + 16*(e >> 2) + 4*g + (e & 3) of the stage; stages start at multiples of 64
    // and W divides 64, so that pixel's column j is the same in every stage: the wrap masks are computed once
    unsigned mL[2][4], mR[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            unsigned l = 0u, r = 0u;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int e = 2 * v + hh;
                const unsigned j = (unsigned)(32 * kk + 16 * (e >> 2) + 4 * g + (e & 3)) & w_mask;
                if (j != 0u) l |= 0xFFFFu << (16 * hh);
                if (j != w_mask) r |= 0xFFFFu << (16 * hh);
            }
            mL[kk][v] = l; mR[kk][v] = r;
        }

    for (int st = 0; st < nsteps; ++st) {
        const int buf = st % NS;
        if (issued - st - 1 >= NS - 2) wg_wait_vmcnt<LPS*(NS - 2)>();
        else wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (issued < nsteps) { issue_stage(issued, issued % NS); ++issued; }
        const char* P = smem + buf * R::STAGE1;
        const char* Q = P + R::P_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            union F8 { struct { s16x4 a, b; } s; elem8 v; u32x4 w; };
            F8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = wr * 32 + i * 16 + 4 * pp;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 32 * kk + 16 * h + 4 * g + qq;
                    const int off = row * 128 + ((((col >> 3) ^ wswz<64>(row)) << 4) | ((col & 4) << 1));
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, P + off));
                    if (h == 0) af[i].s.a = v; else af[i].s.b = v;
                }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                F8 bfr[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = wc * 32 + j * 16 + 4 * pp;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = 32 * kk + 16 * h + 4 * g + qq + t;       // window row = pixel row + 1 + dx
                        const int off = row * 128 + ((((col >> 3) ^ wswz<64>(row)) << 4) | ((col & 4) << 1));
                        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, Q + off));
                        if (h == 0) bfr[j].s.a = v; else bfr[j].s.b = v;
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    F8 a = af[i];
                    if (t == 0) { a.w[0] &= mL[kk][0]; a.w[1] &= mL[kk][1]; a.w[2] &= mL[kk][2]; a.w[3] &= mL[kk][3]; }
                    if (t == 2) { a.w[0] &= mR[kk][0]; a.w[1] &= mR[kk][1]; a.w[2] &= mR[kk][2]; a.w[3] &= mR[kk][3]; }
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[t][i][j] = UDAPOSE_MFMA_16x16x32(a.v, bfr[j].v, acc[t][i][j]);
                }
            }
        }
    }

    const bool atomic = (p.flags & WG_FLAG_ATOMIC) != 0;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r;
                    const int cc = c0 + wc * 32 + j * 16 + (lane & 15);
                    if (rr >= p.Co || rr >= p.rows_valid || cc >= p.Ci) continue;
                    const size_t off = ((size_t)rr * p.wtaps + (3 * by + t)) * p.Ci + cc + part_off;      // tap (dy, dx) -> weight slab 3*(dy+1) + (dx+1)
                    const float v = acc[t][i][j][r];
                    if (atomic) atomicAdd(p.dw + off, v);
                    else p.dw[off] = v;
                }
}

template <int RT, int CT, int WR, int WC, int NS>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware order: the split index (pixel range) is the slowest coordinate, so each XCD's L2 serves one pixel range
    const uint32_t gxy = gridDim.x * gridDim.y;
    const uint32_t lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gxy * gridDim.z);
    const uint32_t bz = lin / gxy, bxy = lin - bz * gxy;
    const uint32_t by = bxy / gridDim.x, bx = bxy - by * gridDim.x;
    if constexpr (RT == 64 && CT == 64) {
        if (p.flags & WG_FLAG_ROW3) { wgrad_row3_body<NS>(p, bx, by, bz, smem); return; }
    }
    if (p.flags & WG_FLAG_FAST2) wgrad_fast2_body<RT, CT, WR, WC, NS, 64>(p, bx, by, bz, smem);
    else if (p.flags & WG_FLAG_FASTGEO) wgrad_dma_body<RT, CT, WR, WC, NS, 64, true>(p, bx, by, bz, smem);
    else wgrad_dma_body<RT, CT, WR, WC, NS>(p, bx, by, bz, smem);
}

// Grouped form: ONE launch computes the weight gradients of many layers.  blk is an [8][per_xcd] table (work-group b runs
// on XCD b & 7, the hardware's round-robin, and takes entry [b & 7][b >> 3]); an entry names a problem of `tab` and the
// work-group's linear index inside that problem's (tile, tap, split) grid, or prob < 0 = padding.  The host deals whole
// (problem, split) units to XCDs by load, so the tiles that re-read one pixel range share an L2.  The table holds byte
// OFFSETS in its x / dy / dw fields, relative to the three bases passed per launch, so one table serves every pass.
template <int RT, int CT, int WR, int WC, int NS, int PX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void wgrad_dma_group_kernel(const WgParams* __restrict__ tab, const WgGroupBlk* __restrict__ blk,
                                                              const uint32_t per_xcd, const char* x_base, const char* dy_base, char* dw_base,
                                                              const WgParams* __restrict__ tab2, const WgGroupBlk* __restrict__ blk2,
                                                              const char* x_base2, const char* dy_base2, char* dw_base2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // PAIR launch (tab2 != null): the groups of TWO passes of one plan (same table shape, their own arenas and gradient buffers) in one
    // grid, interleaved slot by slot - the second pass's heavy work-groups start beside the first's instead of behind its tail.
    uint32_t slot = blockIdx.x >> 3;
    if (tab2) {
        if (slot & 1u) { tab = tab2; blk = blk2; x_base = x_base2; dy_base = dy_base2; dw_base = dw_base2; }
        slot >>= 1;
    }
    const WgGroupBlk b = blk[(blockIdx.x & 7u) * per_xcd + slot];
    if (b.prob < 0) return;
    const WgParams& p = tab[b.prob];
    const uint32_t gx = (uint32_t)(p.r_tiles * p.c_tiles), gxy = gx * (uint32_t)p.total_taps;
    const uint32_t bz = (uint32_t)b.local / gxy, bxy = (uint32_t)b.local - bz * gxy;
    const uint32_t by = bxy / gx, bx = bxy - by * gx;
    if constexpr (RT == 64 && CT == 64 && PX == 64) {
        if (p.flags & WG_FLAG_ROW3) { wgrad_row3_body<NS>(p, bx, by, bz, smem, (uintptr_t)x_base, (uintptr_t)dy_base, (uintptr_t)dw_base); return; }
    }
    if (p.flags & WG_FLAG_FAST2) wgrad_fast2_body<RT, CT, WR, WC, NS, PX>(p, bx, by, bz, smem, (uintptr_t)x_base, (uintptr_t)dy_base, (uintptr_t)dw_base);
    else if (p.flags & WG_FLAG_FASTGEO) wgrad_dma_body<RT, CT, WR, WC, NS, PX, true>(p, bx, by, bz, smem, (uintptr_t)x_base, (uintptr_t)dy_base, (uintptr_t)dw_base);
    else wgrad_dma_body<RT, CT, WR, WC, NS, PX>(p, bx, by, bz, smem, (uintptr_t)x_base, (uintptr_t)dy_base, (uintptr_t)dw_base);
}

template <int RT, int CT, int WR, int WC, int NS, int PX>
int launch_wd_group(const WgParams* d_tab, const WgGroupBlk* d_blk, int per_xcd, const void* x_base, const void* dy_base, void* dw_base,
                    hipStream_t stream, const WgParams* d_tab2 = nullptr, const WgGroupBlk* d_blk2 = nullptr, const void* x_base2 = nullptr,
                    const void* dy_base2 = nullptr, void* dw_base2 = nullptr) {
    using C = WdCfg<RT, CT, WR, WC, NS, PX>;
    constexpr int LDS = (RT == 64 && CT == 64 && PX == 64 && Row3Cfg::lds_bytes(NS) > C::LDS_BYTES) ? Row3Cfg::lds_bytes(NS) : C::LDS_BYTES;
    static std::atomic<unsigned long long> attr_done{0};
    static std::mutex attr_mu;
    once_per_device(attr_done, attr_mu, [] {
        (void)hipFuncSetAttribute((const void*)wgrad_dma_group_kernel<RT, CT, WR, WC, NS, PX>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    hipLaunchKernelGGL((wgrad_dma_group_kernel<RT, CT, WR, WC, NS, PX>), dim3(8 * per_xcd * (d_tab2 ? 2 : 1)), dim3(256), LDS, stream, d_tab, d_blk,
                       (uint32_t)per_xcd, (const char*)x_base, (const char*)dy_base, (char*)dw_base, d_tab2, d_blk2, (const char*)x_base2,
                       (const char*)dy_base2, (char*)dw_base2);
    return udapose_check_launch();
}

template <int RT, int CT, int WR, int WC, int NS>
int launch_wd(WgParams& p, hipStream_t stream) {
    using C = WdCfg<RT, CT, WR, WC, NS>;
    constexpr int LDS = (RT == 64 && CT == 64 && Row3Cfg::lds_bytes(NS) > C::LDS_BYTES) ? Row3Cfg::lds_bytes(NS) : C::LDS_BYTES;
    const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
    const int Rdim = swap ? p.Ci : p.Co, Cdim = swap ? p.Co : p.Ci;
    p.r_tiles = (Rdim + RT - 1) / RT;
    p.c_tiles = (Cdim + CT - 1) / CT;
    static std::atomic<unsigned long long> attr_done{0};
    static std::mutex attr_mu;
    once_per_device(attr_done, attr_mu, [] {
        (void)hipFuncSetAttribute((const void*)wgrad_dma_kernel<RT, CT, WR, WC, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    dim3 grid(p.r_tiles * p.c_tiles, p.total_taps, p.ksplit);
    hipLaunchKernelGGL((wgrad_dma_kernel<RT, CT, WR, WC, NS>), grid, dim3(256), LDS, stream, p);
    return udapose_check_launch();
}

template <int RT, int CT, int WR, int WC>
int launch_wg(WgParams& p, hipStream_t stream) {
    using C = WgCfg<RT, CT, WR, WC>;
    const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
    const bool smallc = (p.flags & IG_FLAG_SMALLC) != 0;
    const int Rdim = swap ? p.Ci : p.Co, Cdim = swap ? p.Co : p.Ci;
    p.r_tiles = (Rdim + RT - 1) / RT;
    p.c_tiles = smallc ? 1 : (Cdim + CT - 1) / CT;
    const int tapsy = smallc ? p.total_taps / 4 : p.total_taps;
    dim3 grid(p.r_tiles * p.c_tiles, tapsy, p.ksplit);
    hipLaunchKernelGGL((wgrad_kernel<RT, CT, WR, WC>), grid, dim3(256), C::LDS_BYTES, stream, p);
    return udapose_check_launch();
}

}  // namespace

// host side of the loader's fast geometry (wgrad_dma_body): stride-1 same-size convolution on power-of-two maps, 24-bit factors
static bool wg_fastgeo_ok(const WgParams& p, const Policy& pol) {
    const bool pow2 = ((p.Hi & (p.Hi - 1)) | (p.Wi & (p.Wi - 1))) == 0;
    return pol.wgrad_fastgeo && !(p.flags & (IG_FLAG_SMALLC | WG_FLAG_SWAP)) && p.s == 1 && p.os == 1 && p.nclass == 1 && p.Hg == p.Hi && p.Wg == p.Wi &&
           p.Hg == p.Ho && p.Wg == p.Wo && pow2 && p.Hi > 0 && p.M < (1 << 24) && p.Ci < (1 << 24) && p.Co < (1 << 24) &&
           (long long)p.M * (p.Ci > p.Co ? p.Ci : p.Co) < (1ll << 31);
}

// the filter-row form (wgrad_row3_body): 3x3 stride-1 pad-1 plain conv in the loader's fast geometry, W <= 64 (a 64-pixel stage
// holds whole image rows, so the column-wrap masks are loop invariants), 64-channel tiles on both sides
static bool wg_row3_ok(const WgParams& p, const Policy& pol) {
    return pol.wgrad_row3 && (p.flags & WG_FLAG_ROW3_OK) && p.total_taps == 9 && p.wtaps == 9 && wg_fastgeo_ok(p, pol) && p.Wi >= 8 && p.Wi <= 64 &&
           p.Ci % 64 == 0 && p.Co % 64 == 0;
}

// tile ids: 0 = 128x128, 1 = 64x64, 2 = 64x32 (Ci==8 stem), 3 = 32x128 (narrow-row: head)
int wgrad_pick_tile(int Rdim, int Cdim, int smallc, const Policy& pol) {
    if (smallc) return 2;
    if (pol.wgrad_tile >= 0 && Rdim > 32) return pol.wgrad_tile;
    if (Rdim <= 32) return 3;
    return 1;    // refined in wgrad_launch once the tap count is known (128x128 only for large weight tensors)
}

int wgrad_launch(WgParams& p, int tile, int accumulate, hipStream_t stream, const Policy& pol) {
    const bool smallc = (p.flags & IG_FLAG_SMALLC) != 0;
    const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
    if (p.Co % 8 != 0 || p.Ci % 8 != 0) return UDAPOSE_ERR_ARG;
    if (smallc && (p.Ci != 8 || swap || (p.total_taps & 7))) return UDAPOSE_ERR_ARG;
    p.div_hw = make_fastdiv((uint32_t)(p.Hg * p.Wg));
    p.div_w = make_fastdiv((uint32_t)p.Wg);
    const int Rdim = swap ? p.Ci : p.Co, Cdim = swap ? p.Co : p.Ci;
    static const int RT[4] = {128, 64, 64, 32}, CT[4] = {128, 64, 32, 128};
    if (tile < 0 || tile > 3) return UDAPOSE_ERR_ARG;
    // measured (tools/tune_conv.py, LDS-DMA kernels): 128x128 tiles for multi-tap convs with >= 128 channels on both sides
    // (3x3 trunk convs, 4x4 deconvs), 64x64 otherwise
    if (tile == 1 && pol.wgrad_tile < 0 && Rdim >= 128 && Cdim >= 128 && p.total_taps >= 9) tile = 0;
    if ((tile == 0 || tile == 1) && pol.wgrad_tile < 0 && wg_row3_ok(p, pol)) {      // three taps per work-group: a third of the LDS fill per FLOP
        tile = 1;
        p.flags |= WG_FLAG_ROW3;
        p.total_taps = 3;
    }
    const long tiles = (long)((Rdim + RT[tile] - 1) / RT[tile]) * (smallc ? 1 : (Cdim + CT[tile] - 1) / CT[tile]) *
                       (smallc ? p.total_taps / 4 : p.total_taps);
    const bool dma = !smallc && (tile == 0 || tile == 1) && (p.Ci % 64 == 0) && (p.Co % 64 == 0);
    const int ms_total = dma ? (p.M + 63) / 64 : (p.M + 31) / 32;
    // split the pixel reduction until ~512 work-groups exist (2 per CU), keeping >= 16 (128x128) / 4 (64x64) stages per
    // split; every extra split adds one fp32 atomic pass over the weight tensor (chip-wide atomic rate 1.3 TB/s)
    const int min_stages = tile == 0 ? 16 : 4;
    int ks = 1;
    while (tiles * ks < 512 && ms_total / (ks * 2) >= min_stages) ks *= 2;
    if (pol.wgrad_ksplit > 0) { ks = pol.wgrad_ksplit; while (ks > 1 && ms_total / ks < 1) ks /= 2; }
    p.ksplit = ks;
    p.part_stride = 0;
    p.msteps_per_split = (ms_total + ks - 1) / ks;
    if (ks > 1 || accumulate) p.flags |= WG_FLAG_ATOMIC; else p.flags &= ~WG_FLAG_ATOMIC;
    if (ks > 1 && !accumulate) {
        const size_t n = (size_t)Rdim * p.wtaps * Cdim;
        if (pw_zero(stream, p.dw, n * sizeof(float)) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
    }
    if (wg_fastgeo_ok(p, pol)) p.flags |= WG_FLAG_FASTGEO; else p.flags &= ~WG_FLAG_FASTGEO;
    if ((p.flags & WG_FLAG_FASTGEO) && !(p.flags & WG_FLAG_ROW3) && pol.wgrad_fastgeo >= 2 && p.M % 64 == 0) p.flags |= WG_FLAG_FAST2; else p.flags &= ~WG_FLAG_FAST2;
    if (dma) return tile == 0 ? launch_wd<128, 128, 2, 2, 2>(p, stream) : launch_wd<64, 64, 2, 2, 4>(p, stream);
    switch (tile) {
        case 0: return launch_wg<128, 128, 2, 2>(p, stream);
        case 1: return launch_wg<64, 64, 2, 2>(p, stream);
        case 2: return launch_wg<64, 32, 2, 2>(p, stream);
        case 3: return launch_wg<32, 128, 1, 4>(p, stream);
        default: return UDAPOSE_ERR_ARG;
    }
}

int wgrad_group_plan(WgParams& p, int accumulate, int stages_per_block, const Policy& pol) {
    const bool smallc = (p.flags & IG_FLAG_SMALLC) != 0;
    const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
    if (smallc && (swap || p.Co % 64 != 0 || p.kw <= 0 || p.kw > 8)) return -1;
    if (!smallc && (p.Ci % 64 != 0 || p.Co % 64 != 0)) return -1;
    const int Rdim = swap ? p.Ci : p.Co, Cdim = smallc ? 64 : (swap ? p.Co : p.Ci);
    p.div_hw = make_fastdiv((uint32_t)(p.Hg * p.Wg));
    p.div_w = make_fastdiv((uint32_t)p.Wg);
    // inside a group the other layers fill the chip, so a layer takes the 128x128 tile (half the L2->LDS bytes per FLOP of
    // 64x64) whenever both of its dimensions allow
    int tile = (Rdim >= 128 && Cdim >= 128) ? 0 : 1;
    if (wg_row3_ok(p, pol)) {       // filter-row form: 64x64 tiles, one work-group per (tile, filter row)
        tile = 1;
        p.flags |= WG_FLAG_ROW3;
        p.total_taps = 3;
    }
    const int TRr = tile == 0 ? 128 : 64, TCc = tile == 1 ? 64 : 128;
    p.r_tiles = (Rdim + TRr - 1) / TRr;
    p.c_tiles = (Cdim + TCc - 1) / TCc;
    const int ms_total = (p.M + 63) / 64;
    int ks = (ms_total + stages_per_block / 2) / stages_per_block;
    if (ks < 1) ks = 1;
    p.ksplit = ks;
    p.part_stride = 0;              // (the caller turns a split problem into the partial-tile form: net.hip build_wg_group)
    p.msteps_per_split = (ms_total + ks - 1) / ks;
    if (ks > 1 || accumulate) p.flags |= WG_FLAG_ATOMIC; else p.flags &= ~WG_FLAG_ATOMIC;
    if (wg_fastgeo_ok(p, pol)) p.flags |= WG_FLAG_FASTGEO; else p.flags &= ~WG_FLAG_FASTGEO;
    if ((p.flags & WG_FLAG_FASTGEO) && !(p.flags & WG_FLAG_ROW3) && pol.wgrad_fastgeo >= 2 && p.M % 64 == 0) p.flags |= WG_FLAG_FAST2; else p.flags &= ~WG_FLAG_FAST2;
    return tile;
}

int wgrad_group_launch(hipStream_t stream, int tile, const WgParams* d_tab, const WgGroupBlk* d_blk, int per_xcd, const void* x_base,
                       const void* dy_base, void* dw_base, const WgParams* d_tab2, const WgGroupBlk* d_blk2, const void* x_base2,
                       const void* dy_base2, void* dw_base2) {
    if (per_xcd <= 0) return UDAPOSE_OK;
    // 32-pixel stages for the 128x128 tile (32 KB of LDS, 128 VGPRs: four resident work-groups per CU instead of two with
    // 64-pixel stages) and a 2-stage ring of 64-pixel stages for the 64x64 tile (35 KB with the filter-row form's reserve: four per
    // CU; round 1 ran three stages = three per CU, with the buffer-load loader two measure -4.5 % alone and -0.05 ms in the step):
    // occupancy beats prefetch depth here as in the igemm (a 3-stage ring for the 128x128 tile at three per CU: +40 % alone in round 2; re-measured in
    // round 6 inside the step: pair launch 3.28 against 2.60 ms, step +0.7 ms - profiles/r6_ab_runs.txt 4)
    if (tile == 0) return launch_wd_group<128, 128, 2, 2, 2, 32>(d_tab, d_blk, per_xcd, x_base, dy_base, dw_base, stream, d_tab2, d_blk2, x_base2, dy_base2, dw_base2);
    return launch_wd_group<64, 64, 2, 2, 2, 64>(d_tab, d_blk, per_xcd, x_base, dy_base, dw_base, stream, d_tab2, d_blk2, x_base2, dy_base2, dw_base2);
}
