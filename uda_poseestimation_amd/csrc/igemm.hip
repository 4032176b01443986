// Implicit-GEMM convolution for gfx950 (MI355X): NHWC bf16 activations, [Co][taps][Ci] bf16 weights, fp32 MFMA accumulate.
//
// One kernel family covers every dense contraction of the pose network and the style network:
//   * fprop of 1x1 / 3x3 / 7x7 convolutions (stride 1 or 2, zero or reflection padding, optional nearest-x2 upsample
//     folded into the loader),
//   * the 4x4 stride-2 transposed convolution as four sub-pixel classes of 2x2 taps,
//   * dgrad of all of the above (a transposed stride-s convolution is again a set of sub-pixel classes).
// A launch is described by a tap table: rows m=(n,i,j) of an Hg x Wg grid, input pixel (i*s+dy, j*s+dx) per tap,
// output pixel (i*os+oa, j*os+ob) per class.
//
// Tiling: 256 threads = 4 waves (wave64).  Block tile BM pixels x BN channels, K step 32 (one MFMA 16x16x32 deep).
// Global -> register -> LDS staging with double buffering (loads for step t+1 issued before the MFMAs of step t),
// 64-byte LDS rows with a 16-byte-chunk XOR swizzle that makes the ds_read_b128 fragment reads conflict-free.
// Epilogue: accumulators go through LDS so that every global store is a full 16-byte (8-channel) vector on
// consecutive channels (128-byte lines per pixel), with bias / residual / ReLU and the BatchNorm batch-statistic
// partial sums (sum, sum of squares of the fp32 accumulators) fused in.  The partial sums are written to a slab
// (one row per wave-row of the grid), not added atomically: every block would hit the same few cache lines.
#include "igemm.h"

namespace {

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <int BM, int BN, int WM, int WN>
struct IgCfg {
    static constexpr int TM = BM / WM, TN = BN / WN;
    static constexpr int MT = TM / 16, NT = TN / 16;
    static constexpr int A_LD = BM / 64 > 0 ? BM / 64 : 1;
    static constexpr int B_LD = BN / 64 > 0 ? BN / 64 : 1;
    static constexpr int ER = TM < 32 ? TM : 32;          // epilogue rows per pass
    static constexpr int ELD = TN + 4;                    // fp32 row stride of the staging tile
    static constexpr int TAP_BYTES = 1024;                // 64 taps
    static constexpr int STAGE_BYTES = 2 * (BM + BN) * 64;
    static constexpr int EPI_BYTES = 4 * ER * ELD * 4;
    static constexpr int LDS_BYTES = TAP_BYTES + (STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES);
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgParams p) {
    using C = IgCfg<BM, BN, WM, WN>;
    constexpr int TM = C::TM, TN = C::TN, MT = C::MT, NT = C::NT, A_LD = C::A_LD, B_LD = C::B_LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    IgTap* taps_l = (IgTap*)smem;
    char* stage = smem + C::TAP_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const IgClass cls = p.cls[blockIdx.z];
    const int n_tile = blockIdx.x % p.n_tiles, m_tile = blockIdx.x / p.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const bool smallc = (p.flags & IG_FLAG_SMALLC) != 0;
    const bool reflect = (p.flags & IG_FLAG_REFLECT) != 0;
    const int up = (p.flags & IG_FLAG_UPSAMPLE) ? 1 : 0;
    const int Hl = p.Hi << up, Wl = p.Wi << up;

    if (tid < cls.ntaps && tid < 64) taps_l[tid] = p.taps[cls.tap_off + tid];

    // ---- per-thread loader state (rows are fixed for the whole K loop)
    const int chunk = tid & 3;
    int a_hi0[A_LD], a_wi0[A_LD], a_nb[A_LD];
    bool a_ok[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int r = (tid >> 2) + 64 * i;
        const int m = m0 + r;
        a_ok[i] = (r < BM) && (m < p.M);
        const uint32_t mm = a_ok[i] ? (uint32_t)m : 0u;
        const uint32_t n = fdiv(mm, p.div_hw);
        const uint32_t rem = mm - n * (uint32_t)(p.Hg * p.Wg);
        const uint32_t ii = fdiv(rem, p.div_w);
        const uint32_t jj = rem - ii * (uint32_t)p.Wg;
        a_hi0[i] = (int)ii * p.s;
        a_wi0[i] = (int)jj * p.s;
        a_nb[i] = (int)n * p.Hi * p.Wi;
    }
    bool b_ok[B_LD];
    size_t b_row[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int r = (tid >> 2) + 64 * i;
        const int co = n0 + r;
        b_ok[i] = (r < BN) && (co < p.Co);
        b_row[i] = (size_t)(b_ok[i] ? co : 0) * (size_t)(p.wtaps * p.Ci);
    }

    const int nsteps = smallc ? (cls.ntaps >> 2) : (cls.ntaps * p.Ci) >> 5;
    u32x4 ra[A_LD], rb[B_LD];
    int tap_cur = 0, c0_cur = 0;   // uniform K cursor (non-SMALLC)

    __syncthreads();   // tap table visible

    auto issue_loads = [&]() {
        const int tap = smallc ? (tap_cur + chunk) : tap_cur;
        const IgTap t = taps_l[tap];
        const int coff = smallc ? 0 : c0_cur + chunk * 8;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int hi = a_hi0[i] + t.dy, wi = a_wi0[i] + t.dx;
            if (reflect) {
                hi = hi < 0 ? -hi : (hi >= Hl ? 2 * Hl - 2 - hi : hi);
                wi = wi < 0 ? -wi : (wi >= Wl ? 2 * Wl - 2 - wi : wi);
            }
            const bool ok = a_ok[i] && (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
            hi >>= up; wi >>= up;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *(const u32x4*)(p.x + ((size_t)(a_nb[i] + hi * p.Wi + wi) * p.Ci + coff));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (b_ok[i]) v = *(const u32x4*)(p.w + (b_row[i] + (size_t)t.widx * p.Ci + coff));
            rb[i] = v;
        }
        if (smallc) tap_cur += 4;
        else { c0_cur += 32; if (c0_cur >= p.Ci) { c0_cur = 0; ++tap_cur; } }
    };
    auto store_lds = [&](int buf) {
        char* A = stage + buf * (BM + BN) * 64;
        char* B = A + BM * 64;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int r = (tid >> 2) + 64 * i;
            if (r < BM) *(u32x4*)(A + r * 64 + ((chunk ^ swz(r)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int r = (tid >> 2) + 64 * i;
            if (r < BN) *(u32x4*)(B + r * 64 + ((chunk ^ swz(r)) << 4)) = rb[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        issue_loads();
        store_lds(0);
    }
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) issue_loads();
        const char* A = stage + buf * (BM + BN) * 64;
        const char* B = A + BM * 64;
        bf16x8 af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * TM + i * 16 + frow;
            af[i] = *(const bf16x8*)(A + r * 64 + ((fchunk ^ swz(r)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int r = wn * TN + j * 16 + frow;
            bfr[j] = *(const bf16x8*)(B + r * 64 + ((fchunk ^ swz(r)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (st + 1 < nsteps) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS (per-wave region) -> 8-channel vectors -> global
    constexpr int ER = C::ER, ELD = C::ELD, LPR = TN / 8 /*lanes per row*/, RPP = 64 / LPR /*rows per pass*/;
    float* est = (float*)stage + wid * ER * ELD;
    const int cg = lane % LPR, rsub = lane / LPR;
    const int cbase = n0 + wn * TN + cg * 8;
    const bool relu = (p.flags & IG_FLAG_RELU) != 0;
    const bool outf32 = (p.flags & IG_FLAG_OUT_F32) != 0;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = (p.bias && cbase + e < p.Co) ? p.bias[cbase + e] : 0.f;

#pragma unroll
    for (int ch = 0; ch < TM / ER; ++ch) {
        // (the trailing __syncthreads of the K loop already ordered the last LDS reads before these writes)
#pragma unroll
        for (int i = 0; i < ER / 16; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    est[(i * 16 + (lane >> 4) * 4 + r) * ELD + j * 16 + (lane & 15)] = acc[ch * (ER / 16) + i][j][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): own wave's writes landed (region is wave-private)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < ER / RPP; ++ps) {
            const int row = ps * RPP + rsub;
            const f32x4 v0 = *(const f32x4*)(est + row * ELD + cg * 8);
            const f32x4 v1 = *(const f32x4*)(est + row * ELD + cg * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const int m = m0 + wm * TM + ch * ER + row;
            if (p.stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
            }
            if (m < p.M && cbase < p.Co) {
                const uint32_t n = fdiv((uint32_t)m, p.div_hw);
                const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Hg * p.Wg);
                const uint32_t ii = fdiv(rem, p.div_w);
                const uint32_t jj = rem - ii * (uint32_t)p.Wg;
                const uint32_t oh = ii * p.os + cls.oa, ow = jj * p.os + cls.ob;
                if (oh >= (uint32_t)p.Ho || ow >= (uint32_t)p.Wo) continue;
                const size_t opix = ((size_t)n * p.Ho + oh) * p.Wo + ow;
                const size_t off = opix * p.Co + cbase;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias[e];
                if (p.res) {
                    const bf16x8 rv = *(const bf16x8*)(p.res + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                if (outf32) {
                    float* yo = (float*)p.y + off;
                    if (cbase + 8 <= p.Co && (p.Co & 3) == 0) {
                        *(f32x4*)yo = (f32x4){v[0], v[1], v[2], v[3]};
                        *(f32x4*)(yo + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (cbase + e < p.Co) yo[e] = v[e];
                    }
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    *(bf16x8*)((bf16_t*)p.y + off) = o;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    if (p.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
            }
        }
        if (lane < LPR && cbase < p.Co) {
            const size_t srow = ((size_t)blockIdx.z * p.m_tiles + m_tile) * WM + wm;
            float* sp = p.stats + srow * 2 * p.Co;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sp[cbase + e] = s1[e]; sp[p.Co + cbase + e] = s2[e]; }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(IgParams& p, hipStream_t stream) {
    using C = IgCfg<BM, BN, WM, WN>;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Co + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    dim3 grid(p.m_tiles * p.n_tiles, 1, p.nclass);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN>), grid, dim3(256), C::LDS_BYTES, stream, p);
    return udapose_check_launch();
}

}  // namespace

// Tile selection: fill the 256 CUs first, then prefer the larger tile.
int igemm_pick_tile(int M, int Co, int nclass) {
    if (Co <= 32) return 3;                       // 128x32 (head / 3-channel outputs)
    if (Co <= 64) return 1;                       // 128x64
    const long t128 = (long)((M + 127) / 128) * ((Co + 127) / 128) * nclass;
    if (t128 >= 512) return 0;                    // 128x128
    return 2;                                     // 64x64
}

int igemm_stat_rows(int M, int Co, int nclass, int tile) {
    switch (tile) {
        case 0: return nclass * ((M + 127) / 128) * 2;
        case 1: return nclass * ((M + 127) / 128) * 2;
        case 2: return nclass * ((M + 63) / 64) * 2;
        default: return nclass * ((M + 127) / 128) * 4;
    }
}

int igemm_launch(IgParams& p, int tile, hipStream_t stream) {
    if (p.Ci % 8 != 0 || (!(p.flags & IG_FLAG_SMALLC) && p.Ci % 32 != 0)) return UDAPOSE_ERR_ARG;
    if ((p.flags & IG_FLAG_SMALLC) && p.Ci != 8) return UDAPOSE_ERR_ARG;
    if (!(p.flags & IG_FLAG_OUT_F32) && (p.Co % 8) != 0) return UDAPOSE_ERR_ARG;
    for (int c = 0; c < p.nclass; ++c) {
        if (p.cls[c].ntaps > 64 || p.cls[c].ntaps < 0) return UDAPOSE_ERR_ARG;
        if ((p.flags & IG_FLAG_SMALLC) && (p.cls[c].ntaps & 3)) return UDAPOSE_ERR_ARG;
    }
    p.div_hw = make_fastdiv((uint32_t)(p.Hg * p.Wg));
    p.div_w = make_fastdiv((uint32_t)p.Wg);
    switch (tile) {
        case 0: return launch_cfg<128, 128, 2, 2>(p, stream);
        case 1: return launch_cfg<128, 64, 2, 2>(p, stream);
        case 2: return launch_cfg<64, 64, 2, 2>(p, stream);
        case 3: return launch_cfg<128, 32, 4, 1>(p, stream);
        default: return UDAPOSE_ERR_ARG;
    }
}
