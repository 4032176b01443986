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
#include "igemm.h"

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
        bf16x8 af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int colb = (wr * TR + i * 16 + 4 * pp) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, P + trow * C::PSTR + colb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, P + (16 + trow) * C::PSTR + colb));
            union { struct { s16x4 a, b; } s; bf16x8 v; } u;
            u.s.a = lo; u.s.b = hi;
            af[i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int colb = (wc * TC + j * 16 + 4 * pp) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, Q + trow * C::QSTR + colb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, Q + (16 + trow) * C::QSTR + colb));
            union { struct { s16x4 a, b; } s; bf16x8 v; } u;
            u.s.a = lo; u.s.b = hi;
            bfr[j] = u.v;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
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

int g_wgrad_tile_override = -1, g_wgrad_ksplit_override = -1;   // debug/tuning hooks

// tile ids: 0 = 128x128, 1 = 64x64, 2 = 64x32 (Ci==8 stem), 3 = 32x128 (narrow-row: head)
int wgrad_pick_tile(int Rdim, int Cdim, int smallc) {
    if (smallc) return 2;
    if (g_wgrad_tile_override >= 0 && Rdim > 32) return g_wgrad_tile_override;
    if (Rdim <= 32) return 3;
    return 1;    // refined in wgrad_launch once the tap count is known (128x128 only for large weight tensors)
}

int wgrad_launch(WgParams& p, int tile, int accumulate, hipStream_t stream) {
    const bool smallc = (p.flags & IG_FLAG_SMALLC) != 0;
    const bool swap = (p.flags & WG_FLAG_SWAP) != 0;
    if (p.Co % 8 != 0 || p.Ci % 8 != 0) return UDAPOSE_ERR_ARG;
    if (smallc && (p.Ci != 8 || swap || (p.total_taps & 7))) return UDAPOSE_ERR_ARG;
    p.div_hw = make_fastdiv((uint32_t)(p.Hg * p.Wg));
    p.div_w = make_fastdiv((uint32_t)p.Wg);
    const int Rdim = swap ? p.Ci : p.Co, Cdim = swap ? p.Co : p.Ci;
    static const int RT[4] = {128, 64, 64, 32}, CT[4] = {128, 64, 32, 128};
    // measured (tools/tune_conv.py): 64x64 tiles win except for large weight tensors with a short pixel reduction
    if (tile == 1 && g_wgrad_tile_override < 0 && Rdim >= 128 && Cdim >= 128 &&
        (long)((Rdim + 127) / 128) * ((Cdim + 127) / 128) * p.total_taps >= 256)
        tile = 0;
    const long tiles = (long)((Rdim + RT[tile] - 1) / RT[tile]) * (smallc ? 1 : (Cdim + CT[tile] - 1) / CT[tile]) *
                       (smallc ? p.total_taps / 4 : p.total_taps);
    const int ms_total = (p.M + 31) / 32;
    // split the pixel reduction until ~1024 workgroups exist, keeping >= 4 steps per split; every extra split adds one
    // fp32 atomic pass over the weight tensor (chip-wide atomic rate 1.3 TB/s)
    int ks = 1;
    while (tiles * ks < 1024 && ms_total / (ks * 2) >= 4) ks *= 2;
    if (g_wgrad_ksplit_override > 0) { ks = g_wgrad_ksplit_override; while (ks > 1 && ms_total / ks < 1) ks /= 2; }
    p.ksplit = ks;
    p.msteps_per_split = (ms_total + ks - 1) / ks;
    if (ks > 1 || accumulate) p.flags |= WG_FLAG_ATOMIC; else p.flags &= ~WG_FLAG_ATOMIC;
    if (ks > 1 && !accumulate) {
        const size_t n = (size_t)Rdim * p.wtaps * Cdim;
        if (hipMemsetAsync(p.dw, 0, n * sizeof(float), stream) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
    }
    switch (tile) {
        case 0: return launch_wg<128, 128, 2, 2>(p, stream);
        case 1: return launch_wg<64, 64, 2, 2>(p, stream);
        case 2: return launch_wg<64, 32, 2, 2>(p, stream);
        case 3: return launch_wg<32, 128, 1, 4>(p, stream);
        default: return UDAPOSE_ERR_ARG;
    }
}
