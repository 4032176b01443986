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

template <int U> struct IC { static constexpr int value = U; };

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

// ---------------------------------------------------------------------------------------------------------------- Ci >= 64, Co % 64 == 0
// The style network's trunk layers.  Tile = 128 output pixels (TH x TW, TW = 64 or 32) x 64 output channels, four waves as 2 (pixels) x 2
// (channels).  K runs over channel slices of 128 bytes per pixel (64 channels; split: 32 as [8 h][8 l] x 4) and, inside a slice, the nine
// taps: the (TH + 2) x (TW + 2) patch of a slice is staged ONCE (next slice's patch in registers meanwhile) and only the 8 KB weight stage
// of a tap moves per step (double-buffered through registers) - 11.8 KB of L2 -> LDS fill per 16 MFMAs and wave against 16 KB for the
// igemm's 128x128 tile and 24 KB for its 128x64 tile, at 50 KB of LDS (three work-groups per CU).  The filter is the MFMA's row operand and
// rows are assigned to channels as in patch3x3_ci8_kernel, so a lane owns one 16-byte chunk of a pixel's output row; rows are assembled in
// LDS and stored as whole 128-byte (256-byte) segments.
struct PgParams {
    const void* x;
    const void* w;
    const float* bias;
    void* y;
    int N, Hi, Wi, Ho, Wo, Ci, Co, up, relu;
};

template <bool SP, int TW, int JB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JB > 2 ? 2 : 3))) void patch3x3_kernel(const PgParams p) {
    constexpr int TH = 128 / TW, PW = TW + 2, PR = TH + 2, NPX = PR * PW;
    constexpr int NCHA = NPX * 8, ITA = (NCHA + 255) / 256;
    constexpr int BN = 32 * JB;                    // output channels per work-group: two wave columns of JB 16-channel blocks
    constexpr int ESZ = SP ? 4 : 2, RC = BN * ESZ / 16;      // RC: 16-byte chunks per output row of the tile
    constexpr int BW = BN * 8 / 256, BSTAGE = BN * 128;      // weight stage: chunks per thread, bytes
    constexpr int KK = SP ? 1 : 2;                 // 32-deep MFMA steps per slice
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch = smem;
    char* const bst = smem + NPX * 128;             // two weight stages of 64 rows x 128 bytes
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4, wm = wid >> 1, wn = wid & 1;
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH, n_tiles = p.Co / BN;
    const uint32_t total = (uint32_t)tiles_x * tiles_y * p.N * n_tiles;
    uint32_t t0 = xcd_remap(blockIdx.x, total);
    const int n_tile = t0 % n_tiles; t0 /= n_tiles;
    const int tx = t0 % tiles_x; t0 /= tiles_x;
    const int ty = t0 % tiles_y;
    const int n = t0 / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = n_tile * BN;
    const int S = p.Ci * ESZ / 128, G = S * 9;
    const size_t rowb = (size_t)p.Ci * ESZ;          // bytes per input pixel / per (co, tap) weight row

    // ---- loaders.  Patch chunk e of this thread: pixel e >> 3 of the patch (reflection at the output resolution, then the folded upsample),
    // chunk e & 7 of the slice; its offset is recomputed at every fetch (once per nine steps) rather than held in ITA registers.  Weight
    // chunk: row n0 + (e >> 3) of tap tp, as a 32-bit offset from the (scalar) tile base.
    const char* xin = (const char*)p.x + (size_t)n * p.Hi * p.Wi * rowb;
    const char* wbase = (const char*)p.w + (size_t)n0 * 9 * rowb;
    unsigned wrow[BW];
#pragma unroll
    for (int i = 0; i < BW; ++i) {
        const int e = tid + i * 256;
        wrow[i] = (unsigned)((size_t)(e >> 3) * 9 * rowb + (e & 7) * 16);
    }
    // Chunk swizzle of a weight stage.  A fragment read touches the 16 rows (qq, r) -> qq * 4 JB + jb * 4 + r, not 16 consecutive ones: the
    // eight rows of one parity differ in bit 1 and in qq, so THOSE bits select the chunk permutation (the igemm's (row >> 1) & 7 maps rows
    // r and r + 16 to the same bank group: 2-way conflicts on every filter read, a third of the kernel's LDS cycles by SQ_LDS_BANK_CONFLICT)
    auto wswz = [](int row) __attribute__((always_inline)) { return ((row >> 1) & 1) | (((row / (4 * JB)) & 3) << 1); };
    u32x4 ra[ITA], rb[BW];
    auto fetchA = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ITA; ++i) {
            const int e = tid + i * 256;
            if (e < NCHA) {
                const int px = e >> 3, ch = e & 7;
                const int r = px / PW, c = px - r * PW;
                const int yy = reflect1(y0 - 1 + r, p.Ho) >> p.up, xx = reflect1(x0 - 1 + c, p.Wo) >> p.up;
                ra[i] = *(const u32x4*)(xin + (unsigned)(((size_t)yy * p.Wi + xx) * rowb + ch * 16) + sl * 128);
            }
        }
    };
    auto storeA = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ITA; ++i) {
            const int e = tid + i * 256;
            if (e < NCHA) {
                const int px = e >> 3, ch = e & 7;
                const int col = px % PW;
                *(u32x4*)(patch + px * 128 + ((ch ^ ((col >> 1) & 7)) << 4)) = ra[i];
            }
        }
    };
    auto fetchB = [&](int sl, int tp) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < BW; ++i) rb[i] = *(const u32x4*)(wbase + (size_t)tp * rowb + sl * 128 + wrow[i]);
    };
    auto storeB = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < BW; ++i) {
            const int e = tid + i * 256;
            const int row = e >> 3, ch = e & 7;
            *(u32x4*)(bst + buf * BSTAGE + row * 128 + ((ch ^ wswz(row)) << 4)) = rb[i];
        }
    };

    // ---- fragment addressing.  Filter rows: MFMA row i = (qq, r) of block jb <-> channel wn * 32 + qq * 8 + jb * 4 + r (patch3x3_ci8_kernel)
    // woff[jb][kk]: byte offset of this lane's filter fragment inside a weight stage (split: the h chunk; l = h ^ 16)
    int woff[JB][KK];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int row = wn * (16 * JB) + (l15 >> 2) * (4 * JB) + jb * 4 + (l15 & 3);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) woff[jb][kk] = row * 128 + (((SP ? 2 * q : kk * 4 + q) ^ wswz(row)) << 4);
    }
    // Patch fragments.  The chunk swizzle of the patch goes by the pixel's COLUMN, (col >> 1) & 7 (PW is even: column parity = pixel
    // parity, so the 16 consecutive columns of a fragment read still fall into 16 bank groups), and a lane's column is l15 + 16 k + dx:
    // its swizzle is (l15 >> 1) & 7 (variant A) or that + 1 (variant B: dx == 2, or dx == 1 on an odd l15) - two precomputed chunk
    // offsets per K sub-step and ONE select per step, instead of a shift / and / xor chain per fragment read.
    int pbase[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int m = wm * 64 + mb * 16 + l15;
        pbase[mb] = ((m / TW) * PW + (m % TW)) * 128;
    }
    int cA[KK], cB[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int c = SP ? 2 * q : kk * 4 + q;
        cA[kk] = (c ^ ((l15 >> 1) & 7)) << 4;
        cB[kk] = (c ^ (((l15 >> 1) + 1) & 7)) << 4;
    }
    const bool odd = (l15 & 1) != 0;
    f32x4 acc[JB][4], acc2[SP ? JB : 1][SP ? 4 : 1];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            acc[jb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SP) acc2[jb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // ---- prologue: patch of slice 0 and weight stage 0 in LDS, stage 1 and the patch of slice 1 in registers (G = 9 S >= 9)
    fetchA(0);
    fetchB(0, 0);
    storeA();
    storeB(0);
    fetchB(0, 1);
    if (S > 1) fetchA(1);
    __syncthreads();
    int sl = 0, tp = 0;
    // (a second register set with stage g + 2 in flight as well changed nothing in the 16-bit form and cost the split form its third
    // work-group per CU: the weight stages come from the L2 well within one step)
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) storeB((g + 1) & 1);         // (every wave left stage (g + 1) & 1 at the barrier that ended step g - 1)
        if (g + 2 < G) {
            int s2 = sl, t2 = tp + 2;
            if (t2 >= 9) { t2 -= 9; ++s2; }
            fetchB(s2, t2);
        }
        const char* B = bst + (g & 1) * BSTAGE;
        const int dx = tp % 3;
        const char* ptap = patch + ((tp / 3) * PW + dx) * 128;       // (scalar)
        const bool useB = dx == 2 || (dx == 1 && odd);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            u32x4 wa[JB], wb[SP ? JB : 1];
#pragma unroll
            for (int jb = 0; jb < JB; ++jb) {
                wa[jb] = *(const u32x4*)(B + woff[jb][kk]);
                if constexpr (SP) wb[jb] = *(const u32x4*)(B + (woff[jb][kk] ^ 16));
            }
            const int csel = useB ? cB[kk] : cA[kk];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const char* prow = ptap + pbase[mb];
                if constexpr (SP) {
                    const half8 xh = *(const half8*)(prow + csel);
                    const half8 xl = *(const half8*)(prow + (csel ^ 16));
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) {
                        const half8 ah = __builtin_bit_cast(half8, wa[jb]), al = __builtin_bit_cast(half8, wb[jb]);
                        acc[jb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh, acc[jb][mb], 0, 0, 0);
                        acc2[jb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl, acc2[jb][mb], 0, 0, 0);
                        acc2[jb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh, acc2[jb][mb], 0, 0, 0);
                    }
                } else {
                    const elem8 xf = *(const elem8*)(prow + csel);
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) acc[jb][mb] = UDAPOSE_MFMA_16x16x32(__builtin_bit_cast(elem8, wa[jb]), xf, acc[jb][mb]);
                }
            }
        }
        __syncthreads();
        if (++tp == 9) {
            tp = 0; ++sl;
            if (sl < S) {                            // slice boundary: the next slice's patch (in registers since the last boundary) replaces this one
                storeA();
                if (sl + 1 < S) fetchA(sl + 1);
                __syncthreads();
            }
        }
    }
    // ---- epilogue: lane (pixel l15 of block mb, q) owns channels wn * 16 JB + q * 4 JB .. + 4 JB of its pixel = JB / 2 chunks (split: chunk pairs)
    // of the pixel's BN-channel output row, assembled in LDS (the staging buffers are free: the loop ended with a barrier), then whole rows out
    float bias_r[4 * JB];
#pragma unroll
    for (int e = 0; e < 4 * JB; ++e) bias_r[e] = p.bias ? p.bias[n0 + wn * (16 * JB) + q * (4 * JB) + e] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int m = wm * 64 + mb * 16 + l15;
        char* row = smem + m * (RC * 16);
#pragma unroll
        for (int c2 = 0; c2 < JB / 2; ++c2) {
            float v[8];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jb = c2 * 2 + j2;
                    float t2 = acc[jb][mb][r];
                    if constexpr (SP) t2 += acc2[jb][mb][r] * UDAPOSE_SP_INV;
                    t2 += bias_r[jb * 4 + r];
                    v[j2 * 4 + r] = p.relu ? (t2 > 0.f ? t2 : 0.f) : t2;
                }
            const int ci = (wn * 4 + q) * (JB / 2) + c2;          // 8-channel group of the row
            if constexpr (SP) {
                half8 h8, l8;
                sp_split8(v, h8, l8);
                const int sw = m & 15;
                *(half8*)(row + (((2 * ci) ^ sw) << 4)) = h8;
                *(half8*)(row + (((2 * ci + 1) ^ sw) << 4)) = l8;
            } else {
                elem8 o8;
#pragma unroll
                for (int r = 0; r < 8; ++r) o8[r] = (elem_t)v[r];
                *(elem8*)(row + ((ci ^ (RC == 16 ? (m & 15) : ((m >> 1) & 7))) << 4)) = o8;
            }
        }
    }
    __syncthreads();
    char* yo = (char*)p.y + (((size_t)n * p.Ho + y0) * p.Wo + x0) * p.Co * ESZ + (size_t)n0 * ESZ;
#pragma unroll
    for (int it = 0; it < RC / 2; ++it) {
        const int L = it * 256 + tid;
        const int m = L / RC, c = L % RC;
        const int sw = RC == 16 ? (m & 15) : ((m >> 1) & 7);
        const int r = m / TW, cc = m % TW;
        *(u32x4*)(yo + ((size_t)r * p.Wo + cc) * p.Co * ESZ + c * 16) = *(const u32x4*)(smem + m * (RC * 16) + ((c ^ sw) << 4));
    }
}

template <bool SP, int TW, int JB>
int launch_pg(const PgParams& p, hipStream_t s) {
    constexpr int TH = 128 / TW, BN = 32 * JB, LDS = (TH + 2) * (TW + 2) * 128 + 2 * BN * 128;       // (the output rows reuse it)
    static_assert(LDS >= 128 * BN * (SP ? 4 : 2), "epilogue rows fit the staging buffers");
    static std::atomic<unsigned long long> attr_done{0};
    static std::mutex attr_mu;
    once_per_device(attr_done, attr_mu, [] {
        (void)hipFuncSetAttribute((const void*)patch3x3_kernel<SP, TW, JB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    const long long total = (long long)(p.Wo / TW) * (p.Ho / TH) * p.N * (p.Co / BN);
    if (total <= 0 || total > 0x7fffffffll) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL((patch3x3_kernel<SP, TW, JB>), dim3((unsigned)total), dim3(256), LDS, s, p);
    return udapose_check_launch();
}

}  // namespace

// 1 when conv_fprop hands this geometry / epilogue to a patch-staged kernel
static int patch_trunk_ok(const ConvGeom& g, const ConvEpilogue& e);
int patch_conv_ok(const ConvGeom& g, const ConvEpilogue& e) {
    if (e.scale) return 0;                 // (per-channel scale: the igemm epilogue only)
    if (patch_trunk_ok(g, e)) return 1;
    if (!g.policy().patch_conv || g.transposed || g.upsample || !g.reflect || g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return 0;
    if (e.f32 || e.res || e.stats || g.Wi % 64 != 0 || g.Hi < 2 || g.Wi < 2) return 0;
    if (g.Ci == 64 && g.Co <= 16 && e.out_f32 && g.Hi % 4 == 0) return 1;
    if (g.Ci == 8 && g.Co == 64 && !e.out_f32 && g.Hi % 4 == 0) return 1;
    return 0;
}

// 1 when the geometry takes the trunk form (patch3x3_kernel): policy patch_conv >= 2
static int patch_trunk_ok(const ConvGeom& g, const ConvEpilogue& e) {
    if (g.policy().patch_conv < 2 || g.transposed || !g.reflect || g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return 0;
    if (e.f32 || e.res || e.stats || e.out_f32 || g.Ci < 64 || g.Ci % 64 || g.Co % 64) return 0;
    const int Ho = g.Ho(), Wo = g.Wo();
    if (Wo % 32 || Ho < 2 || Wo < 2) return 0;
    // tile shape: 4 x 32 pixels (a 6 x 34 patch: 1.6 staged pixels per output pixel) where the height allows, else 2 x 64 (4 x 66: 2.1)
    // (measured on the style network, N = 32: 2.85 against 2.89 ms per pass in the 16-bit form, 6.26 against 6.66 in the split form, whose
    // 2 x 64 variant also spills)
    const int TW = (Ho % 4 == 0) ? 32 : 64;
    if (Wo % TW || Ho % (128 / TW)) return 0;
    if ((long long)g.Hi * g.Wi * g.Ci * (e.split ? 4 : 2) >= (1ll << 32)) return 0;       // (32-bit patch offsets per image)
    return TW;
}

int patch_conv_fprop(hipStream_t s, const ConvGeom& g, const void* x, const void* w_fwd, void* y, const ConvEpilogue& e) {
    PcParams p{x, w_fwd, e.bias, y, g.N, g.Hi, g.Wi, g.Co, e.relu ? 1 : 0, e.out_f32 ? 1 : 0};
    if (g.Ci == 64 && g.Co <= 16) return e.split ? launch_co16<true, 4>(p, s) : launch_co16<false, 4>(p, s);
    if (g.Ci == 8 && g.Co == 64) return e.split ? launch_ci8<true>(p, s) : launch_ci8<false>(p, s);
    if (const int TW = patch_trunk_ok(g, e)) {
        PgParams q{x, w_fwd, e.bias, y, g.N, g.Hi, g.Wi, g.Ho(), g.Wo(), g.Ci, g.Co, g.upsample ? 1 : 0, e.relu ? 1 : 0};
        // 128 output channels per work-group in the 16-bit form where Co allows (policy patch_conv >= 3; the split form's two accumulator
        // sets keep it at 64)
        const bool wide = !e.split && g.Co % 128 == 0 && g.policy().patch_conv >= 3;
        if (TW == 64) return e.split ? launch_pg<true, 64, 2>(q, s) : (wide ? launch_pg<false, 64, 4>(q, s) : launch_pg<false, 64, 2>(q, s));
        return e.split ? launch_pg<true, 32, 2>(q, s) : (wide ? launch_pg<false, 32, 4>(q, s) : launch_pg<false, 32, 2>(q, s));
    }
    return UDAPOSE_ERR_UNSUPPORTED;
}

UDAPOSE_SP_SAT_READER(sp_sat_read_patchconv)
