// Patch-staged 3x3 convolutions for the two ENDS of the style network (Style_net.py:32-62 decoder's last layer 64 -> 3, :64-118 encoder's
// first layer 3 -> 64 with the 1x1 pre-convolution folded in), 256x256 maps, reflection padding.
//
// The tap-staged implicit GEMM (igemm.hip) fetches the A operand once per filter tap: nine L2 -> LDS passes over the input.  For the trunk's
// layers that traffic is amortised over 64-128 output channels per tile; these two layers have 3 output channels (16 with padding: 268 MB of
// input fetched nine times for 2.4 GFLOP of useful work) resp. 3 input channels (8 with padding: every tap a 16-byte gather per pixel), and
// ran at 22 / 40 TFLOP/s - 5x / 3x above the time their HBM bytes need.  Here a work-group stages the (TH + 2) x 66 pixel patch of its
// TH x 64 output tile ONCE (reflection applied while staging), the filter lives in registers as ready-made MFMA B fragments, and the nine
// taps are nine shifted fragment reads of the same LDS patch:
//   * patch3x3_co16_kernel: Ci = 64, Co <= 16 (one 16-wide MFMA column block; lanes of absent channels feed zeros), fp32 output;
//   * patch3x3_ci8_kernel:  Ci = 8 (3 real channels), Co = 64, a K step of 32 = four taps x 8 channels, 16-bit or split output.
// Both in the 16-bit element type (one v_mfma_f32_16x16x32 per fragment pair) and in the f16x2 split form (common.h: three fp16 MFMAs per
// fragment pair into two accumulators, combined as in igemm.hip).  HBM-bound by construction: algorithmic bytes = input + output once.
#include "conv_plan.h"

namespace {

struct PcParams {
    const void* x;
    const void* w;
    const float* bias;
    void* y;
    int N, H, W, Co, relu, out_f32;
};

__device__ __forceinline__ int reflect1(int v, int n) { return v < 0 ? -v : (v >= n ? 2 * n - 2 - v : v); }

// ---------------------------------------------------------------------------------------------------------------- Ci = 64, Co <= 16
// LDS patch: pixel p = r * 66 + c at byte p * 128 (16-bit: its 64 channels; split: 32 channels as [8 h][8 l] x 4 - the split form stages and
// multiplies the two channel halves one after the other, so that both forms hold TH = 4 rows in 50.7 KB: three work-groups per CU), the
// 16-byte chunks XOR-swizzled with (p >> 1) & 7 as in igemm.hip: the 16 pixels of a fragment read fall into 16 different bank groups.
template <bool SP, int TH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SP ? 2 : 3))) void patch3x3_co16_kernel(const PcParams p) {
    constexpr int PW = 66, PR = TH + 2, NPX = PR * PW;
    constexpr int PARTS = SP ? 2 : 1, ROWB = 128 * PARTS;
    constexpr int NCH = NPX * 8, ITER = (NCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: an XCD owns a contiguous range of tiles (vertical neighbours share two of their PR patch rows through its L2)
    const int tiles_x = p.W / 64, tiles_y = p.H / TH;
    const uint32_t total = (uint32_t)tiles_x * tiles_y * p.N;
    uint32_t t = xcd_remap(blockIdx.x, total);
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int x0 = tx * 64, y0 = ty * TH;
    const char* xin = (const char*)p.x + (size_t)n * p.H * p.W * ROWB;
    const int co = lane & 15, q = lane >> 4;
    const bool co_ok = co < p.Co;

    // TH pixel blocks of 16 per wave: block mb = wid * TH + i -> tile row mb / 4, columns (mb % 4) * 16 ..
    f32x4 acc[TH], acc2[SP ? TH : 1];
#pragma unroll
    for (int i = 0; i < TH; ++i) {
        acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int part = 0; part < PARTS; ++part) {
        // ---- stage the patch: loads first (16 bytes per lane each, IB in flight), then the swizzled LDS stores; the split form in two
        // batches (its two accumulator sets and (h, l) filter fragments leave fewer registers for loads in flight)
        constexpr int NB = SP ? 2 : 1, IB = (ITER + NB - 1) / NB;
        if (part > 0) __syncthreads();             // (every wave has read the previous half's fragments)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            u32x4 regs[IB];
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int e = tid + (bt * IB + i) * 256;
                if (e < NCH) {
                    const int px = e >> 3, ch = e & 7;
                    const int r = px / PW, c = px - r * PW;
                    const int yy = reflect1(y0 - 1 + r, p.H), xx = reflect1(x0 - 1 + c, p.W);
                    regs[i] = *(const u32x4*)(xin + ((size_t)yy * p.W + xx) * ROWB + part * 128 + ch * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int e = tid + (bt * IB + i) * 256;
                if (e < NCH) {
                    const int px = e >> 3, ch = e & 7;
                    *(u32x4*)(smem + px * 128 + ((ch ^ ((px >> 1) & 7)) << 4)) = regs[i];
                }
            }
        }
        // ---- the filter as B fragments: lane (co = lane & 15, q = lane >> 4) holds 8 channels of tap tp: 16-bit (h * 4 + q) * 8 .. of K
        // half h; split: channels part * 32 + q * 8 .. as (h, l)
        u32x4 b0[9], b1[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const size_t row = ((size_t)(co_ok ? co : 0) * 9 + tp) * 64;
            if constexpr (SP) {
                const char* wp = (const char*)p.w + (row + part * 32 + q * 8) * 4;
                b0[tp] = *(const u32x4*)wp;
                b1[tp] = *(const u32x4*)(wp + 16);
            } else {
                b0[tp] = *(const u32x4*)((const elem_t*)p.w + row + q * 8);
                b1[tp] = *(const u32x4*)((const elem_t*)p.w + row + (4 + q) * 8);
            }
            if (!co_ok) { b0[tp] = (u32x4){0, 0, 0, 0}; b1[tp] = (u32x4){0, 0, 0, 0}; }
        }
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int dy = tp / 3, dx = tp % 3;
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const int mb = wid * TH + i;
                const int px = ((mb >> 2) + dy) * PW + (mb & 3) * 16 + dx + (lane & 15);
                const char* prow = smem + px * 128;
                const int sw = (px >> 1) & 7;
                if constexpr (SP) {
                    const half8 ah = *(const half8*)(prow + (((2 * q) ^ sw) << 4));
                    const half8 al = *(const half8*)(prow + (((2 * q + 1) ^ sw) << 4));
                    const half8 bh = __builtin_bit_cast(half8, b0[tp]), bl = __builtin_bit_cast(half8, b1[tp]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc2[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc2[i], 0, 0, 0);
                } else {
                    const elem8 a0 = *(const elem8*)(prow + ((q ^ sw) << 4));
                    const elem8 a1 = *(const elem8*)(prow + (((4 + q) ^ sw) << 4));
                    acc[i] = UDAPOSE_MFMA_16x16x32(a0, __builtin_bit_cast(elem8, b0[tp]), acc[i]);
                    acc[i] = UDAPOSE_MFMA_16x16x32(a1, __builtin_bit_cast(elem8, b1[tp]), acc[i]);
                }
            }
        }
    }
    // ---- epilogue: lane (co, q) holds pixels q * 4 .. + 4 of each block; fp32 [pixel][Co] output
    if (co_ok) {
        const float b = p.bias ? p.bias[co] : 0.f;
        float* yo = (float*)p.y + (size_t)n * p.H * p.W * p.Co;
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            const int mb = wid * TH + i;
            const int yy = y0 + (mb >> 2), xb = x0 + (mb & 3) * 16 + q * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][r];
                if constexpr (SP) v += acc2[i][r] * UDAPOSE_SP_INV;
                v += b;
                if (p.relu) v = v > 0.f ? v : 0.f;
                yo[((size_t)yy * p.W + xb + r) * p.Co + co] = v;
            }
        }
    }
}

template <bool SP, int TH>
int launch_co16(const PcParams& p, hipStream_t s) {
    constexpr int LDS = (TH + 2) * 66 * 128;
    static std::atomic<unsigned long long> attr_done{0};
    static std::mutex attr_mu;
    once_per_device(attr_done, attr_mu, [] {
        (void)hipFuncSetAttribute((const void*)patch3x3_co16_kernel<SP, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    const long long total = (long long)(p.W / 64) * (p.H / TH) * p.N;
    if (total <= 0 || total > 0x7fffffffll) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL((patch3x3_co16_kernel<SP, TH>), dim3((unsigned)total), dim3(256), LDS, s, p);
    return udapose_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------- Ci = 8, Co = 64
// Patch of 16-byte pixels (split: an h plane and an l plane of 16-byte pixels, so that the 16 pixels of a fragment read stay contiguous).
// K = 12 taps x 8 channels = three 32-deep steps (taps 9..11 are zero fragments); lane group q of step s owns tap 4s + q.  The FILTER is the
// MFMA's row operand here: D[co][pixel], so a lane ends up with consecutive channels of ONE pixel (eight, through the row <-> channel
// assignment below) = one 16-byte chunk of the output row; the rows of a wave's 64 pixels are assembled in LDS and stored lane-linearly.
// Weights: the igemm's Ci == 8 pack [Co][3][KWp = 8][8] (taps kw >= 3 are padding), in the split form [8 h][8 l] per tap.
// PERSISTENT: the grid is one round of resident work-groups, each walks tiles blockIdx.x, + gridDim.x, ... (all on its own XCD); the filter
// fragments are loaded once, the next tile's patch is in flight (registers) while the current one is multiplied and stored - a work-group's
// life has one exposed memory round trip instead of three per tile (per-tile form: 95 / 205 us; the stores saturate the memory pipeline and
// every dependent load behind them took ~4 us).
template <bool SP, int TH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SP ? 2 : 3))) void patch3x3_ci8_kernel(const PcParams p) {
    constexpr int PW = 66, PR = TH + 2, NPX = PR * PW;
    constexpr int PLANES = SP ? 2 : 1, NCH = NPX * PLANES, ITER = (NCH + 255) / 256;
    constexpr int RC = SP ? 16 : 8;               // 16-byte chunks per output pixel
    static_assert(TH == 4, "one tile row per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int tiles_x = p.W / 64, tiles_y = p.H / TH;
    const uint32_t total = (uint32_t)tiles_x * tiles_y * p.N;
    char* const tr = smem + NCH * 16 + wid * (64 * RC * 16);      // this wave's output rows: [64 pixels][RC chunks], swizzled

    // ---- filter fragments, once.  Block (half, jb): MFMA row i = (qq, r) = (i >> 2, i & 3) stands for channel half * 32 + qq * 8 + jb * 4 + r,
    // so that lane group q ends up with the EIGHT consecutive channels half * 32 + q * 8 .. + 8 of its pixel (rows q * 4 + r of blocks jb = 0, 1):
    // one 16-byte chunk (split: the (h, l) chunk pair) of the pixel's output row.
    u32x4 wh[2][2][3], wl[SP ? 2 : 1][SP ? 2 : 1][SP ? 3 : 1];
    float* const bias_l = (float*)(smem + NCH * 16 + 4 * 64 * RC * 16);      // [64], behind the waves' output rows
    if (tid < 64) bias_l[tid] = p.bias ? p.bias[tid] : 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int tp = s3 * 4 + q;
                const int ch = half * 32 + (l15 >> 2) * 8 + jb * 4 + (l15 & 3);
                const size_t e = (((size_t)ch * 3 + tp / 3) * 8 + tp % 3) * 8;
                wh[half][jb][s3] = (u32x4){0, 0, 0, 0};
                if constexpr (SP) wl[half][jb][s3] = (u32x4){0, 0, 0, 0};
                if (tp < 9) {
                    if constexpr (SP) {
                        wh[half][jb][s3] = *(const u32x4*)((const char*)p.w + e * 4);
                        wl[half][jb][s3] = *(const u32x4*)((const char*)p.w + e * 4 + 16);
                    } else {
                        wh[half][jb][s3] = *(const u32x4*)((const elem_t*)p.w + e);
                    }
                }
            }
    }

    u32x4 regs[ITER];
    auto fetch = [&](uint32_t tl) __attribute__((always_inline)) {
        uint32_t t = xcd_remap(tl, total);
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int x0 = tx * 64, y0 = ty * TH;
        const char* xin = (const char*)p.x + (size_t)n * p.H * p.W * (16 * PLANES);
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int e = tid + i * 256;
            if (e < NCH) {
                const int px = SP ? (e >> 1) : e, pl = SP ? (e & 1) : 0;
                const int r = px / PW, c = px - r * PW;
                const int yy = reflect1(y0 - 1 + r, p.H), xx = reflect1(x0 - 1 + c, p.W);
                regs[i] = *(const u32x4*)(xin + ((size_t)yy * p.W + xx) * (16 * PLANES) + pl * 16);
            }
        }
    };
    uint32_t tl = blockIdx.x;
    if (tl < total) fetch(tl);
#pragma unroll 1
    for (; tl < total; tl += gridDim.x) {
        __syncthreads();                          // every wave is done with the previous tile's patch
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int e = tid + i * 256;
            if (e < NCH) {
                const int px = SP ? (e >> 1) : e, pl = SP ? (e & 1) : 0;
                *(u32x4*)(smem + pl * (NPX * 16) + px * 16) = regs[i];
            }
        }
        uint32_t t = xcd_remap(tl, total);
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        char* yo = (char*)p.y + ((size_t)n * p.H * p.W + (size_t)(ty * TH + wid) * p.W + tx * 64) * (RC * 16);
        if (tl + gridDim.x < total) fetch(tl + gridDim.x);       // in flight during this tile's MFMAs and stores
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4 bs0 = *(const f32x4*)(bias_l + half * 32 + q * 8), bs1 = *(const f32x4*)(bias_l + half * 32 + q * 8 + 4);
            // MG pixel blocks at a time (the split form's two accumulator sets and (h, l) filter fragments leave room for two)
            constexpr int MG = SP ? 2 : 4;
#pragma unroll
            for (int m0 = 0; m0 < 4; m0 += MG) {
                f32x4 acc[2][MG], acc2[SP ? 2 : 1][SP ? MG : 1];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int mi = 0; mi < MG; ++mi) {
                        acc[jb][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if constexpr (SP) acc2[jb][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int tp = s3 * 4 + q;
                    const bool ok = tp < 9;
                    const int tpc = ok ? tp : 0;
                    const int pbase = (wid + tpc / 3) * PW + tpc % 3 + l15;
#pragma unroll
                    for (int mi = 0; mi < MG; ++mi) {
                        const int px = pbase + (m0 + mi) * 16;
                        u32x4 xh = *(const u32x4*)(smem + px * 16);
                        if (!ok) xh = (u32x4){0, 0, 0, 0};
                        if constexpr (SP) {
                            u32x4 xl = *(const u32x4*)(smem + NPX * 16 + px * 16);
                            if (!ok) xl = (u32x4){0, 0, 0, 0};
                            const half8 bh = __builtin_bit_cast(half8, xh), bl = __builtin_bit_cast(half8, xl);
#pragma unroll
                            for (int jb = 0; jb < 2; ++jb) {
                                const half8 ah = __builtin_bit_cast(half8, wh[half][jb][s3]), al = __builtin_bit_cast(half8, wl[half][jb][s3]);
                                acc[jb][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[jb][mi], 0, 0, 0);
                                acc2[jb][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc2[jb][mi], 0, 0, 0);
                                acc2[jb][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc2[jb][mi], 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int jb = 0; jb < 2; ++jb)
                                acc[jb][mi] = UDAPOSE_MFMA_16x16x32(__builtin_bit_cast(elem8, wh[half][jb][s3]), __builtin_bit_cast(elem8, xh), acc[jb][mi]);
                        }
                    }
                }
                // lane (pixel l15 of block mb, q) -> its chunk of the pixel's row in the transposition region (XOR-swizzled: the 16 pixels of
                // a write phase and the 16 consecutive chunks of a read phase fall into different bank groups)
#pragma unroll
                for (int mi = 0; mi < MG; ++mi) {
                    float v[8];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float t2 = acc[jb][mi][r];
                            if constexpr (SP) t2 += acc2[jb][mi][r] * UDAPOSE_SP_INV;
                            t2 += jb ? bs1[r] : bs0[r];
                            v[jb * 4 + r] = p.relu ? (t2 > 0.f ? t2 : 0.f) : t2;
                        }
                    const int px = (m0 + mi) * 16 + l15;
                    char* row = tr + px * (RC * 16);
                    if constexpr (SP) {
                        half8 h8, l8;
                        sp_split8(v, h8, l8);
                        const int c = (half * 4 + q) * 2, sw = px & 15;
                        *(half8*)(row + ((c ^ sw) << 4)) = h8;
                        *(half8*)(row + (((c + 1) ^ sw) << 4)) = l8;
                    } else {
                        elem8 o8;
#pragma unroll
                        for (int r = 0; r < 8; ++r) o8[r] = (elem_t)v[r];
                        *(elem8*)(row + (((half * 4 + q) ^ ((px >> 1) & 7)) << 4)) = o8;
                    }
                }
            }
        }
        // lane-linear read-out: a store instruction writes 1 KB of CONSECUTIVE bytes of the tile row (8 / 4 whole pixels).  (The direct form
        // - every lane its own 16 bytes, 128 bytes apart - ran at the L2's request rate, 2.3 TB/s of writes; half lines were no faster.)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < RC; ++it) {
            const int L = it * 64 + lane;
            const int px = L / RC, c = L % RC;
            const int sw = SP ? (px & 15) : ((px >> 1) & 7);
            *(u32x4*)(yo + (size_t)L * 16) = *(const u32x4*)(tr + px * (RC * 16) + ((c ^ sw) << 4));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool SP>
int launch_ci8(const PcParams& p, hipStream_t s) {
    constexpr int TH = 4, LDS = (TH + 2) * 66 * 16 * (SP ? 2 : 1) + 4 * 64 * (SP ? 256 : 128) + 256;       // patch + the waves' output rows + bias
    const long long total = (long long)(p.W / 64) * (p.H / TH) * p.N;
    if (total <= 0 || total > 0x7fffffffll) return UDAPOSE_ERR_ARG;
    static std::atomic<unsigned long long> attr_done{0};
    static std::mutex attr_mu;
    once_per_device(attr_done, attr_mu, [] {
        (void)hipFuncSetAttribute((const void*)patch3x3_ci8_kernel<SP, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    // one round of resident work-groups (256 CUs x 3 / 2 by registers), a multiple of 8 so that a work-group's tiles share its XCD
    const long long slots = 256 * (SP ? 2 : 3);
    const unsigned grid = (unsigned)(total < slots ? total : slots);
    hipLaunchKernelGGL((patch3x3_ci8_kernel<SP, TH>), dim3(grid), dim3(256), LDS, s, p);
    return udapose_check_launch();
}

}  // namespace

// 1 when conv_fprop hands this geometry / epilogue to a patch-staged kernel
int patch_conv_ok(const ConvGeom& g, const ConvEpilogue& e) {
    if (!g.policy().patch_conv || g.transposed || g.upsample || !g.reflect || g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return 0;
    if (e.f32 || e.res || e.stats || g.Wi % 64 != 0 || g.Hi < 2 || g.Wi < 2) return 0;
    if (g.Ci == 64 && g.Co <= 16 && e.out_f32 && g.Hi % 4 == 0) return 1;
    if (g.Ci == 8 && g.Co == 64 && !e.out_f32 && g.Hi % 4 == 0) return 1;
    return 0;
}

int patch_conv_fprop(hipStream_t s, const ConvGeom& g, const void* x, const void* w_fwd, void* y, const ConvEpilogue& e) {
    PcParams p{x, w_fwd, e.bias, y, g.N, g.Hi, g.Wi, g.Co, e.relu ? 1 : 0, e.out_f32 ? 1 : 0};
    if (g.Ci == 64 && g.Co <= 16) return e.split ? launch_co16<true, 4>(p, s) : launch_co16<false, 4>(p, s);
    if (g.Ci == 8 && g.Co == 64) return e.split ? launch_ci8<true>(p, s) : launch_ci8<false>(p, s);
    return UDAPOSE_ERR_UNSUPPORTED;
}
