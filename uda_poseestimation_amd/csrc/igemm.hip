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
#include <stdlib.h>
#include <algorithm>
#include "conv_plan.h"

namespace {

// 8 KiB of zeros: the source of every padded / out-of-range lane of the LDS-DMA loads (large enough that a padded lane
// may add the per-stage channel offset, < Ci * sizeof(T) <= 8 KiB, and still read zeros)
__device__ u32x4 g_zero16[512];

template <int U> struct IC { static constexpr int value = U; };
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(IC<N - 1>{}); }
}

// XOR swizzle of the 16-byte chunk index inside a 128-byte LDS row: makes the ds_read_b128 fragment reads (lane l ->
// row l&15, chunk l>>4) conflict-free (2 rows per 256-byte bank row; derivation in DESIGN.md)
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int BM, int BN, int WM, int WN, int NS>
struct IgCfg {
    static constexpr int BK = 64;                          // K elements per stage (128-byte LDS rows)
    static constexpr int BNL = BN < 64 ? 64 : BN;          // B rows held in LDS (every wave issues the same number of DMAs)
    static constexpr int TM = BM / WM, TN = BN / WN;
    static constexpr int MT = TM / 16, NT = TN / 16;
    static constexpr int A_PW = BM / 32, B_PW = BNL / 32;  // LDS-DMA instructions per wave per stage (8 rows x 128 B each)
    static constexpr int ER = TM < 32 ? TM : 32;           // epilogue rows per pass
    static constexpr int ELD = TN + 4;                     // fp32 row stride of the staging tile
    static constexpr int TAP_BYTES = 1024;                 // 64 taps
    static constexpr int STAGE1 = (BM + BNL) * 128;
    static constexpr int STAGE_BYTES = NS * STAGE1;
    static constexpr int EPI_BYTES = 4 * ER * ELD * 4;
    static constexpr int LDS_BYTES = TAP_BYTES + (STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES);
    static_assert(EPI_BYTES + WM * 2 * BN * 4 <= STAGE_BYTES, "the wave-row exchange of the BN statistics sits behind the epilogue regions");
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 24, "vmcnt range");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

// Main loop: NS-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no staging registers), counted vmcnt waits and
// ONE raw s_barrier per K step, so NS-1 stages of loads stay in flight across barriers while the MFMAs of the current
// stage run (the loads are latency-bound otherwise: a 64x64 tile only has 64 MFMA cycles of work per 32-deep step).
template <typename T, int BM, int BN, int WM, int WN, int NS, bool BS = false, int H3 = 0, bool SP = false>
// amdgpu_waves_per_eu(4): a register budget of 128 per lane.  Left alone the compiler spends 168 + 24 AGPRs on the 128x64 tile
// (two resident work-groups per CU); with the hint it needs 110 and none of the configurations the heuristic picks spills
// (the 128x128 ones, reachable only through the tuning override, do).  Measured: -0.65 ms per step.
// The BS variants (dgrads with the consumer BatchNorm's backward reduction in the epilogue) take waves_per_eu(3) = 168 registers:
// their tiles are resident three per CU by LDS either way, and the epilogue keeps a whole chunk's y / z / skip loads in flight.
// SP (with T = float as the 4-byte stride type): the operands are f16x2 split tensors (common.h) - the loaders are the fp32 ones
// byte for byte, the fragments of a 32-channel stage are the row's chunk pairs (2q, 2q+1) = (h, l) of lane group q, and a stage is
// three fp16 MFMAs per fragment pair into two accumulator sets (h.h | h.l + l.h, the second scaled by 2^-11 at the end).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM * BN >= 128 * 128 ? 2 : ((BS || H3 == 1 || SP) ? 3 : 4)))) void igemm_kernel(const IgParams p) {
    using C = IgCfg<BM, BN, WM, WN, NS>;
    // T = bf16 (MFMA 16x16x32 bf16) or float (exact fp32 MFMA 16x16x4: the reference's own precision for the teacher and
    // validate(); 1/16 of the bf16 rate, used for strict-parity forward passes).  A stage is 128 bytes of K per row either way.
    constexpr bool F32 = sizeof(T) == 4;
    static_assert(!SP || (F32 && !BS && (H3 == 0 || H3 == 3)), "SP: fp32-shaped loaders, plain or lean form, forward epilogue");
    constexpr int EPC = 16 / (int)sizeof(T);      // elements per 16-byte chunk
    constexpr int BKE = 128 / (int)sizeof(T);     // K elements per stage
    constexpr int TPS = BKE / 8;                  // taps per stage on the Ci == 8 path
    const T* px = (const T*)p.x;
    const T* pw = (const T*)p.w;
    constexpr int TM = C::TM, TN = C::TN, MT = C::MT, NT = C::NT, A_PW = C::A_PW, B_PW = C::B_PW, BNL = C::BNL;
    constexpr int LPS = A_PW + B_PW;    // DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    IgTap* taps_l = (IgTap*)smem;
    char* stage = smem + C::TAP_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: LDS-DMA bases stay scalar
    unsigned long long* const dbg = p.dbg ? p.dbg + (size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 : nullptr;
    if (dbg && tid == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
    const int wm = wid / WN, wn = wid % WN;
    // XCD-aware order: each XCD owns a contiguous range of (class, m_tile) so its L2 holds a 1/8 slice of the activations
    const uint32_t tiles = (uint32_t)p.m_tiles * p.n_tiles;
    const uint32_t wid_lin = xcd_remap(blockIdx.x + gridDim.x * blockIdx.z, tiles * gridDim.z);
    const int cls_id = (int)(wid_lin / tiles);
    const uint32_t tile_id = wid_lin - cls_id * tiles;
    const IgClass cls = p.cls[cls_id];
    const int n_tile = tile_id % p.n_tiles, m_tile = tile_id / p.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const bool smallc = (p.flags & IG_FLAG_SMALLC) != 0;
    const bool reflect = (p.flags & IG_FLAG_REFLECT) != 0;
    const int up = (p.flags & IG_FLAG_UPSAMPLE) ? 1 : 0;
    const int Hl = p.Hi << up, Wl = p.Wi << up;

    // 1x1 convolutions (and the one non-empty class of their strided data gradients) have the single tap (0, 0) -> weight slab 0:
    // the table entry is written from registers, which takes a dependent global-memory round trip out of every such work-group's
    // prologue (two thirds of the launches of a step)
    if constexpr (H3) {
        // (the run-staged 3x3 form derives its taps from the compile-time tap index and IG_FLAG_MIRROR: no table)
    } else if (p.flags & IG_FLAG_TAP0) { if (tid == 0) taps_l[0] = IgTap{0, 0, 0, cls_id}; }
    else if (tid < cls.ntaps && tid < 64) taps_l[tid] = p.taps[cls.tap_off + tid];

    // ---- per-lane loader state: DMA instruction i of this wave fills LDS rows (i*4+wid)*8 .. +8, lane -> (row, chunk)
    const int lrow = lane >> 3, pchunk = lane & 7;
    int a_hi0[A_PW], a_wi0[A_PW], a_nb[A_PW], a_lc[A_PW];
    bool a_ok[A_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int r = (i * 4 + wid) * 8 + lrow;
        const int m = m0 + r;
        a_ok[i] = m < p.M;
        a_lc[i] = pchunk ^ swz(r);                     // logical chunk this lane fetches (swizzle on the SOURCE side)
        const uint32_t mm = a_ok[i] ? (uint32_t)m : 0u;
        const uint32_t n = fdiv(mm, p.div_hw);
        const uint32_t rem = mm - n * (uint32_t)(p.Hg * p.Wg);
        const uint32_t ii = fdiv(rem, p.div_w);
        const uint32_t jj = rem - ii * (uint32_t)p.Wg;
        a_hi0[i] = (int)ii * p.s;
        a_wi0[i] = (int)jj * p.s;
        a_nb[i] = (int)n * p.Hi * p.Wi;
    }
    bool b_ok[B_PW];
    int b_lc[B_PW];
    size_t b_row[B_PW];
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
        const int r = (i * 4 + wid) * 8 + lrow;
        const int co = n0 + r;
        b_ok[i] = (r < BN) && (co < p.Co);
        b_lc[i] = pchunk ^ swz(r);
        b_row[i] = (size_t)(b_ok[i] ? co : 0) * (size_t)(p.wtaps * p.Ci);
    }

    const int nsteps = smallc ? cls.ntaps / TPS : (cls.ntaps * p.Ci) / BKE;
    int tap_cur = 0, c0_cur = 0;   // uniform K cursor
    // H3 == 3, "lean 1x1": a stride-1 1x1 convolution (or its data gradient) with full tiles.  Row m of A IS pixel m, so every DMA
    // source is  uniform base (advanced 128 bytes per stage by the scalar unit) + a per-lane 32-bit offset that never changes:
    // no vector instruction per stage goes into addresses (the generic loop spends ~70 scalar / vector instructions per 8 MFMAs)
    unsigned a_off3[A_PW], b_off3[B_PW];
    if constexpr (H3 == 3) {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            const int r = (i * 4 + wid) * 8 + lrow;
            a_off3[i] = ((unsigned)(m0 + r) * (unsigned)p.Ci + (unsigned)((pchunk ^ swz(r)) * EPC)) * (unsigned)sizeof(T);
        }
#pragma unroll
        for (int i = 0; i < B_PW; ++i) {
            const int r = (i * 4 + wid) * 8 + lrow;
            b_off3[i] = ((unsigned)(n0 + (r < BN ? r : 0)) * (unsigned)(p.wtaps * p.Ci) + (unsigned)((pchunk ^ swz(r)) * EPC)) * (unsigned)sizeof(T);
        }
    }
    const char* zsrc = (const char*)g_zero16;

    __syncthreads();   // tap table visible

    // Fast path (zero padding, no upsample, Ci >= one stage): everything that depends on the row is hoisted out of the K
    // loop.  Per row: the byte pointer of its (tap 0,0) pixel and a bit mask of the taps that fall inside the image.  When
    // the tap changes (every Ci/BKE stages) the per-row source pointers are re-selected once (pointer or zero page); inside
    // a tap every stage only adds the wave-uniform channel offset.  The zero page is 8 KiB, so a padded lane may add the
    // channel offset as well and still read zeros.  (The K loop was issue-bound on bookkeeping: 69 VALU + 59 SALU
    // instructions per 8 MFMAs, profiles/r1_pmc_notes.md.)
    const bool fast = !smallc && !reflect && !up;
    const char* a_ptr0[A_PW];
    unsigned long long a_mask[A_PW];
    const char* b_ptr0[B_PW];
    const char* a_cur[A_PW];
    const char* b_cur[B_PW];
    if (fast) {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            a_ptr0[i] = (const char*)(px + (((long long)a_nb[i] + (long long)a_hi0[i] * p.Wi + a_wi0[i]) * p.Ci + a_lc[i] * EPC));
            a_mask[i] = 0ull;
            a_cur[i] = zsrc;
        }
        // tap-validity masks: ONE LDS read per tap shared by this lane's rows (the row-outer form paid an LDS round trip per
        // (row, tap): 36 dependent reads = 2-4 us of a 3x3 conv's prologue, measured with udapose_debug_set_timeline)
#pragma unroll 4
        for (int t = 0; t < cls.ntaps; ++t) {
            const IgTap tt = taps_l[t];
#pragma unroll
            for (int i = 0; i < A_PW; ++i) {
                const int hi = a_hi0[i] + tt.dy, wi = a_wi0[i] + tt.dx;
                if (a_ok[i] && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi) a_mask[i] |= 1ull << t;
            }
        }
#pragma unroll
        for (int i = 0; i < B_PW; ++i) { b_ptr0[i] = (const char*)(pw + ((long long)b_row[i] + b_lc[i] * EPC)); b_cur[i] = zsrc; }
    }
    auto select_tap = [&]() __attribute__((always_inline)) {      // per-row source pointers of tap `tap_cur` (wave-uniform tap)
        const IgTap t = taps_l[tap_cur];
        const long long ab = (((long long)t.dy * p.Wi + t.dx) * p.Ci) * (long long)sizeof(T);
        const long long bb = ((long long)t.widx * p.Ci) * (long long)sizeof(T);
#pragma unroll
        for (int i = 0; i < A_PW; ++i) a_cur[i] = ((a_mask[i] >> tap_cur) & 1ull) ? a_ptr0[i] + ab : zsrc;
#pragma unroll
        for (int i = 0; i < B_PW; ++i) b_cur[i] = b_ok[i] ? b_ptr0[i] + bb : zsrc;
    };

    // issue the LDS-DMA loads of the next K stage into ring buffer UB (compile-time: LDS addresses fold to constants)
    auto issue_stage = [&](auto ub) __attribute__((always_inline)) {
        constexpr int UB = decltype(ub)::value;
        char* A = stage + UB * C::STAGE1;
        char* B = A + BM * 128;
        if constexpr (H3 == 3) {
            // (scalar base = the kernel argument itself, 32-bit lane offsets advanced by one add each: the saddr form of the load)
#pragma unroll
            for (int i = 0; i < A_PW; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)px + a_off3[i]),
                                                 (__attribute__((address_space(3))) void*)(A + (i * 4 + wid) * 1024), 16, 0, 0);
                a_off3[i] += 128u;
            }
#pragma unroll
            for (int i = 0; i < B_PW; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)pw + b_off3[i]),
                                                 (__attribute__((address_space(3))) void*)(B + (i * 4 + wid) * 1024), 16, 0, 0);
                b_off3[i] += 128u;
            }
            return;
        }
        if (fast) {
            if (c0_cur == 0) select_tap();
            const long long cb = (long long)c0_cur * (long long)sizeof(T);
#pragma unroll
            for (int i = 0; i < A_PW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_cur[i] + cb),
                                                 (__attribute__((address_space(3))) void*)(A + (i * 4 + wid) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < B_PW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_cur[i] + cb),
                                                 (__attribute__((address_space(3))) void*)(B + (i * 4 + wid) * 1024), 16, 0, 0);
            c0_cur += BKE;
            if (c0_cur >= p.Ci) { c0_cur = 0; ++tap_cur; }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            const IgTap t = taps_l[smallc ? tap_cur + a_lc[i] * EPC / 8 : tap_cur];
            const int coff = smallc ? (a_lc[i] * EPC) & 7 : c0_cur + a_lc[i] * EPC;
            int hi = a_hi0[i] + t.dy, wi = a_wi0[i] + t.dx;
            if (reflect) {
                hi = hi < 0 ? -hi : (hi >= Hl ? 2 * Hl - 2 - hi : hi);
                wi = wi < 0 ? -wi : (wi >= Wl ? 2 * Wl - 2 - wi : wi);
            }
            const bool ok = a_ok[i] && (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
            hi >>= up; wi >>= up;
            const char* src = ok ? (const char*)(px + ((size_t)(a_nb[i] + hi * p.Wi + wi) * p.Ci + coff)) : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(A + (i * 4 + wid) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PW; ++i) {
            const IgTap t = taps_l[smallc ? tap_cur + b_lc[i] * EPC / 8 : tap_cur];
            const int coff = smallc ? (b_lc[i] * EPC) & 7 : c0_cur + b_lc[i] * EPC;
            const char* src = b_ok[i] ? (const char*)(pw + (b_row[i] + (size_t)t.widx * p.Ci + coff)) : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(B + (i * 4 + wid) * 1024), 16, 0, 0);
        }
        if (smallc) tap_cur += TPS;
        else { c0_cur += BKE; if (c0_cur >= p.Ci) { c0_cur = 0; ++tap_cur; } }
    };

    f32x4 acc[MT][NT];
    f32x4 acc2[SP ? MT : 1][SP ? NT : 1];       // SP: the cross terms h.l + l.h (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SP) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // per-lane fragment offsets inside a stage (row * 128 + swizzled chunk), computed once
    const int frow = lane & 15, fchunk = lane >> 4;
    int a_fo[MT][2], b_fo[NT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * TM + i * 16 + frow;
#pragma unroll
        for (int k = 0; k < 2; ++k) a_fo[i][k] = r * 128 + (((F32 ? 2 * fchunk + k : fchunk + 4 * k) ^ swz(r)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wn * TN + j * 16 + frow;
#pragma unroll
        for (int k = 0; k < 2; ++k) b_fo[j][k] = BM * 128 + r * 128 + (((F32 ? 2 * fchunk + k : fchunk + 4 * k) ^ swz(r)) << 4);
    }

    auto compute = [&](auto u) __attribute__((always_inline)) {
        constexpr int U = decltype(u)::value;
        const char* S = stage + U * C::STAGE1;
        if constexpr (SP) {
            // lane group q = lane>>4 owns k = 8q..8q+7 of the 32-deep stage: chunk 2q holds their h halves, chunk 2q+1 their l halves
            // (a_fo / b_fo [.][0] and [.][1] of the fp32 form are exactly these two chunks)
            half8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) { ah[i] = *(const half8*)(S + a_fo[i][0]); al[i] = *(const half8*)(S + a_fo[i][1]); }
#pragma unroll
            for (int j = 0; j < NT; ++j) { bh[j] = *(const half8*)(S + b_fo[j][0]); bl[j] = *(const half8*)(S + b_fo[j][1]); }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
                }
        } else if constexpr (!F32) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                elem8 af[MT], bfr[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *(const elem8*)(S + a_fo[i][kk]);
#pragma unroll
                for (int j = 0; j < NT; ++j) bfr[j] = *(const elem8*)(S + b_fo[j][kk]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = UDAPOSE_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
            }
        } else {
            // fp32: lane group q = lane>>4 owns k = 8q..8q+7 of the 32-deep stage (two 16-byte chunks of its row);
            // sub-step e feeds element e of both operands to one exact 16x16x4 MFMA (the k <-> lane-group
            // assignment is the same for A and B, so the sum over k is unchanged)
            f32x4 al[MT][2], bl[NT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) al[i][h] = *(const f32x4*)(S + a_fo[i][h]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) bl[j][h] = *(const f32x4*)(S + b_fo[j][h]);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[i][e >> 2][e & 3], bl[j][e >> 2][e & 3], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (H3 == 1) {
        // ---- 3x3, stride 1, pad 1 (fprop, or the data gradient of such a conv): the A operand is staged ONCE per 64-channel
        // chunk instead of once per tap.  Output rows m0 .. m0+BM-1 are consecutive pixels (n, i, j) and the input has the same
        // geometry, so tap (dy, dx) of row m reads input pixel m + dy*W + dx: the stage holds the run of BM + 2(W+1)
        // consecutive pixels m0-(W+1) .. m0+BM+W (rows outside the tensor come from the zero page) and every tap reads it at a
        // constant row offset.  Taps that fall outside the image (zero padding; the run's neighbours belong to other image rows
        // or images there) are zeroed per lane in the fragment registers.  L2->LDS bytes per 64 channels: (BM + 2W + 2) rows of A
        // + 9 weight tiles, instead of 9 x (BM rows + weight tile): -41 % at W = 16 - and the L2->LDS feed is what bounds these
        // launches.  Ring: two A buffers (chunk c+1 is loaded, one piece per wave per tap, under the taps of chunk c) and three
        // weight-tile slots (tile s+2 is issued at sub-stage s; 9 taps = 3 x 3 slots, so the slot of a tap is a constant).
        static_assert(!F32 && B_PW == 2 && BN == 64, "H3: bf16, 64 weight rows");
        const int Wd = p.Wi, G = Wd + 1;
        const int RA = (BM + 2 * G + 7) & ~7, NPc = RA >> 3;      // A rows per stage; 1 KiB pieces (8 rows) per stage, <= 28
        char* const Ab0 = stage;                                   // RA rows + one row of zeros (what the padded taps read)
        char* const Ab1 = stage + (RA + 1) * 128;
        char* const Bq = stage + 2 * (RA + 1) * 128;               // 3 slots of BNL rows
        constexpr int NBT = 3;                                     // weight-tile slots
        char* const dump = Bq + NBT * BNL * 128;                   // 1 KiB: where the pieces beyond NPc go (uniform DMA counts)
        const int nchunks = p.Ci / BKE, nsub = nchunks * 9;
        // A 3x3 pad-1 plan is t -> (dy, dx) = +-(t/3 - 1, t%3 - 1), weight slab t (build_direct; mirrored for the data gradient):
        // one sign flag from the host (IG_FLAG_MIRROR), everything else follows from the compile-time tap index (no table reads, no
        // arrays of scalars: the parameter block already fills most of the scalar registers)
        const int sgn = (p.flags & IG_FLAG_MIRROR) ? -1 : 1;
        const int ataps = (NPc + 3) >> 2;                          // taps of a chunk during which A pieces of the next chunk are issued
        const int rb8 = 8 * p.Ci * (int)sizeof(T);                 // bytes between two pieces in the source
        const int idx0 = m0 - G + lrow;                            // source pixel of this lane's row of piece 0
        const int lc0 = (pchunk ^ swz(lrow)) ^ ((wid & 1) << 2);   // piece k = 4t + wid: swz(8k + lrow) = swz(lrow) ^ ((k & 1) << 2)
        const char* const a_p0 = (const char*)px + ((long long)idx0 * p.Ci + lc0 * EPC) * (long long)sizeof(T);
        int bfo3[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) bfo3[j][k] = b_fo[j][k] - BM * 128;
        auto issue_a = [&](int k, int chunk, char* Ab) __attribute__((always_inline)) {        // k = 4j + wid: wave-uniform piece index
            const bool real = k < NPc;
            const int idx = idx0 + 8 * k;
            const char* src = (real && (unsigned)idx < (unsigned)p.M) ? a_p0 + ((long long)k * rb8 + (long long)chunk * 128) : zsrc;
            char* dst = real ? Ab + k * 1024 : dump;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        };
        auto issue_b = [&](int tapi, int chunk, int slot) __attribute__((always_inline)) {
            const long long bb = ((long long)tapi * p.Ci + (long long)chunk * BKE) * (long long)sizeof(T);
            char* Bs = Bq + slot * (BNL * 128);
#pragma unroll
            for (int i = 0; i < B_PW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ok[i] ? b_ptr0[i] + bb : zsrc),
                                                 (__attribute__((address_space(3))) void*)(Bs + (i * 4 + wid) * 1024), 16, 0, 0);
        };
        // prologue, FIRST (the loads fly while the fragment offsets below are computed): the whole A stage of chunk 0 (the pieces
        // beyond NPc go to the dump), and the first weight tiles
        if (dbg && tid == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
        for (int j = 0; j < ataps; ++j) issue_a(j * 4 + wid, 0, Ab0);
        issue_b(0, 0, 0);
        issue_b(1, 0, 1);
        // Per lane, per fragment row and tap: the LDS byte offset (inside an A buffer) of the 16-byte fragment piece - the run row
        // of the tap's pixel with its swizzle, or the zero row when the tap falls outside the image.  Computed once: the K loop
        // then spends ONE add per fragment read (the loop is VALU-issue bound: ~70 scalar / vector instructions per 8 MFMAs in
        // the tap-staged form), and zero padding costs nothing there.
        static_assert(MT % 2 == 0, "H3: fragment rows are packed in pairs");
        unsigned fo3[MT / 2][2][9];                                 // two 16-bit LDS offsets per register (rows 2h, 2h+1)
        if (tid < 16) {                                             // the two zero rows (visible after the first barrier)
            *(u32x4*)(Ab0 + RA * 128 + (tid & 7) * 16 + (tid >> 3) * ((RA + 1) * 128)) = (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int h = 0; h < MT / 2; ++h)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < 9; ++t) fo3[h][kk][t] = 0u;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * TM + i * 16 + frow, m = m0 + r;
            int ii = -4, jj = -4;                                   // rows beyond M: every tap reads zeros
            if (m < p.M) {
                const uint32_t n = fdiv((uint32_t)m, p.div_hw);
                const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Hg * p.Wg);
                ii = (int)fdiv(rem, p.div_w);
                jj = (int)rem - ii * p.Wg;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = sgn * (t / 3 - 1), dx = sgn * (t % 3 - 1);
                const bool ok = (unsigned)(ii + dy) < (unsigned)p.Hi && (unsigned)(jj + dx) < (unsigned)p.Wi;
                const int rr = r + G + dy * Wd + dx;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    fo3[i / 2][kk][t] |= (unsigned)(ok ? rr * 128 + (((fchunk + 4 * kk) ^ swz(rr)) << 4) : RA * 128 + ((fchunk + 4 * kk) << 4)) << (16 * (i & 1));
            }
        }
        auto compute3 = [&](auto tc, const char* Ab, const char* Bs) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                elem8 af[MT], bfr[NT];
#pragma unroll
                for (int h = 0; h < MT / 2; ++h) {
                    unsigned w = fo3[h][kk][t];
                    asm volatile("" : "+v"(w));       // (opaque: the unpacked offsets are loop-invariant, and hoisting 4 x 18 of them spills)
                    af[2 * h] = *(const elem8*)(Ab + (w & 0xFFFFu));
                    af[2 * h + 1] = *(const elem8*)(Ab + (w >> 16));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) bfr[j] = *(const elem8*)(Bs + bfo3[j][kk]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = UDAPOSE_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
            }
        };
        __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): the zero rows are written before the first barrier
        for (int c = 0; c < nchunks; ++c) {
            const bool has_next = c + 1 < nchunks;
            const char* Acur = (c & 1) ? Ab1 : Ab0;
            char* Anext = (c & 1) ? Ab0 : Ab1;
            static_for<9>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                const int sidx = c * 9 + t;
                // DMAs younger than weight tile s (which, with the A stage of this chunk, must have landed): tile s+1 (2 per wave)
                // and the A pieces issued at sub-stages s-2 (after tile s) and s-1, where those taps issue one
                int young = 0;
                if (sidx + 1 < nsub) {
                    young = 2;
                    if (has_next) young += (t >= 1 && t - 1 < ataps ? 1 : 0) + (t >= 2 && t - 2 < ataps ? 1 : 0);
                }
                if (young == 0) wait_vmcnt<0>();
                else if (young == 2) wait_vmcnt<2>();
                else if (young == 3) wait_vmcnt<3>();
                else wait_vmcnt<4>();
                __builtin_amdgcn_s_barrier();
                if (t == 0 && dbg && tid == 0 && c == 0) dbg[2] = __builtin_amdgcn_s_memrealtime();
                if (sidx + 2 < nsub) issue_b((t + 2) % 9, c + ((t + 2) >= 9 ? 1 : 0), (t + 2) % 3);
                if (t < ataps && has_next) issue_a(t * 4 + wid, c + 1, Anext);
                compute3(tc, Acur, Bq + (t % 3) * (BNL * 128));
            });
        }
    } else {
    // prologue: NS-1 stages in flight
    if (dbg && tid == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
    int issued = 0;
    static_for<NS - 1>([&](auto u) __attribute__((always_inline)) {
        if (decltype(u)::value < nsteps) { issue_stage(u); ++issued; }
    });

    for (int st0 = 0; st0 < nsteps; st0 += NS) {
        static_for<NS>([&](auto u) __attribute__((always_inline)) {
            constexpr int U = decltype(u)::value;
            const int st = st0 + U;
            if (st < nsteps) {
                // stage st must have landed: at most (issued - st - 1) younger stages may still be in flight
                if (issued - st - 1 >= NS - 2) wait_vmcnt<LPS*(NS - 2)>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();          // every wave's share of stage st is in LDS; buffer (st-1)%NS is free
                if (dbg && tid == 0 && st == 0) dbg[2] = __builtin_amdgcn_s_memrealtime();
                if (issued < nsteps) { issue_stage(IC<(U + NS - 1) % NS>{}); ++issued; }
                compute(u);
            }
        });
    }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // all LDS reads of the ring are done before the epilogue reuses it
    if constexpr (SP) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += acc2[i][j][r] * UDAPOSE_SP_INV;
    }
    if (dbg && tid == 0) dbg[3] = __builtin_amdgcn_s_memrealtime();

    // ---- epilogue: accumulators -> LDS (per-wave region) -> 8-channel vectors -> global
    constexpr int ER = C::ER, ELD = C::ELD, LPR = TN / 8 /*lanes per row*/, RPP = 64 / LPR /*rows per pass*/;
    float* est = (float*)stage + wid * ER * ELD;
    const int cg = lane % LPR, rsub = lane / LPR;
    const int cbase = n0 + wn * TN + cg * 8;
    const bool relu = (p.flags & IG_FLAG_RELU) != 0;
    const bool outf32 = (F32 && !SP) || (p.flags & IG_FLAG_OUT_F32) != 0;
    const bool lin_out = p.os == 1 && cls.oa == 0 && cls.ob == 0 && p.Hg == p.Ho && p.Wg == p.Wo;
    if (!BS && p.stats) {
        // BN partial statistics straight from the accumulators: a lane holds rows (lane>>4)*4+r of every 16-row tile for
        // column j*16 + (lane&15), so the column sum is lane-local over (tile, r) plus two xor steps over lane>>4; the WM wave
        // rows of the work-group are then added through LDS, so the slab has ONE row per m-tile (the BN kernels that re-reduce
        // it per work-group read half as much).  (rows m >= M are zero-filled operand rows: they add 0)
        float* xch = (float*)(stage + C::EPI_BYTES);           // [WM][2][BN], behind the per-wave epilogue regions
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; a += v; b += v * v; }
            a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            const int cl = wn * TN + j * 16 + lane;
            if (lane < 16) { xch[(wm * 2 + 0) * BN + cl] = a; xch[(wm * 2 + 1) * BN + cl] = b; }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid % BN, col = n0 + cl;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += xch[(w * 2 + st) * BN + cl];
            const size_t srow = (size_t)cls_id * p.m_tiles + m_tile;
            if (col < p.Co) p.stats[(srow * 2 + st) * p.Co + col] = t;
        }
    }
    float bias[8], scl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bias[e] = (p.bias && cbase + e < p.Co) ? p.bias[cbase + e] : 0.f;
        scl[e] = (p.scale && cbase + e < p.Co) ? p.scale[cbase + e] : 1.f;       // (x * 1 + b == x + b exactly: the unscaled form is unchanged)
    }
    // BS (dgrad feeding a BatchNorm backward): per-lane coefficients of the consumer BN's 8 channels and the running partial
    // sums of g and g * xhat over this lane's rows
    float bmu[8], bis[8], bsc[8], bsh[8], bs1[8], bs2[8];
    if constexpr (BS) {
        const bool cok = cbase < p.Co;        // (Co % 8 == 0 is a launch requirement of this mode)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bmu[e] = cok ? p.bs_mean[cbase + e] : 0.f;
            bis[e] = cok ? p.bs_invstd[cbase + e] : 0.f;
            bs1[e] = 0.f; bs2[e] = 0.f; bsc[e] = 0.f; bsh[e] = 0.f;
        }
        if (!p.bs_z && cok) {
            // the forward's own scale / shift expressions (bn_finalize_k): the mask is z > 0 without reading z
#pragma unroll
            for (int e = 0; e < 8; ++e) { bsc[e] = p.bs_gamma[cbase + e] * bis[e]; bsh[e] = p.bs_beta[cbase + e] - bmu[e] * bsc[e]; }
        }
    }

    if constexpr (BS) {
        // BS epilogue: the loads of a chunk's rows (consumer BN's y, its z when the mask is z > 0, the skip gradient) are all
        // issued BEFORE the accumulators go through LDS, so one memory latency is exposed per chunk instead of two per row
        // (the generic loop loads the residual, waits, loads y and z, waits: 8 exposed HBM latencies per 128-row tile, and these
        // launches are HBM-bound: layer1's 64->256 data gradient moves 285 MB)
        constexpr int NP = ER / RPP;
        const bool use_mask = (p.flags & IG_FLAG_BSMASK) != 0;      // the saved ReLU bit mask (1 byte per 8 channels) instead of z
        const bool use_z = p.bs_z != nullptr && !use_mask, use_res = p.res != nullptr;
        const unsigned char* const bs_mask = (const unsigned char*)p.bs_z;
#pragma unroll
        for (int ch = 0; ch < TM / ER; ++ch) {
            size_t offs[NP];
            bool oks[NP];
            elem8 yv[NP], zv[NP], rv[NP];
            unsigned mv[NP];
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int m = m0 + wm * TM + ch * ER + ps * RPP + rsub;
                bool ok = m < p.M && cbase < p.Co;
                size_t opix = (size_t)m;
                if (!lin_out && ok) {
                    const uint32_t n = fdiv((uint32_t)m, p.div_hw);
                    const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Hg * p.Wg);
                    const uint32_t ii = fdiv(rem, p.div_w);
                    const uint32_t jj = rem - ii * (uint32_t)p.Wg;
                    const uint32_t oh = ii * p.os + cls.oa, ow = jj * p.os + cls.ob;
                    ok = oh < (uint32_t)p.Ho && ow < (uint32_t)p.Wo;
                    opix = ((size_t)n * p.Ho + oh) * p.Wo + ow;
                }
                oks[ps] = ok;
                offs[ps] = ok ? opix * p.Co + cbase : 0;
                yv[ps] = elem8{}; zv[ps] = elem8{}; rv[ps] = elem8{};
                mv[ps] = 0u;
                if (ok) {
                    yv[ps] = *(const elem8*)(p.bs_y + offs[ps]);
                    if (use_mask) mv[ps] = bs_mask[offs[ps] >> 3];
                    if (use_z) zv[ps] = *(const elem8*)(p.bs_z + offs[ps]);
                    if (use_res) rv[ps] = *(const elem8*)(p.res + offs[ps]);
                }
            }
#pragma unroll
            for (int i = 0; i < ER / 16; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        est[(i * 16 + (lane >> 4) * 4 + r) * ELD + j * 16 + (lane & 15)] = acc[ch * (ER / 16) + i][j][r];
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int row = ps * RPP + rsub;
                const f32x4 v0 = *(const f32x4*)(est + row * ELD + cg * 8);
                const f32x4 v1 = *(const f32x4*)(est + row * ELD + cg * 8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (oks[ps]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float yf = (float)yv[ps][e];
                        const float val = v[e] + (float)rv[ps][e];             // (rv is zero without a skip gradient)
                        const float ty = yf * bsc[e] + bsh[e];
                        const float t = use_mask ? (float)((mv[ps] >> e) & 1u) : (use_z ? (float)zv[ps][e] : ty);
                        float gv = t > 0.f ? val : 0.f;
                        if (!outf32) gv = (float)(elem_t)gv;                    // the sums see exactly the value the BN apply kernel will read
                        bs1[e] += gv;
                        bs2[e] += gv * ((yf - bmu[e]) * bis[e]);
                        v[e] = gv;
                    }
                    if (outf32) {
                        float* yo = (float*)p.y + offs[ps];
                        *(f32x4*)yo = (f32x4){v[0], v[1], v[2], v[3]};
                        *(f32x4*)(yo + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    } else {
                        elem8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (elem_t)v[e];
                        *(elem8*)((elem_t*)p.y + offs[ps]) = o;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
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
            if (m < p.M && cbase < p.Co) {
                size_t opix = (size_t)m;                 // row grid == output grid (every conv but the sub-pixel classes)
                if (!lin_out) {
                    const uint32_t n = fdiv((uint32_t)m, p.div_hw);
                    const uint32_t rem = (uint32_t)m - n * (uint32_t)(p.Hg * p.Wg);
                    const uint32_t ii = fdiv(rem, p.div_w);
                    const uint32_t jj = rem - ii * (uint32_t)p.Wg;
                    const uint32_t oh = ii * p.os + cls.oa, ow = jj * p.os + cls.ob;
                    if (oh >= (uint32_t)p.Ho || ow >= (uint32_t)p.Wo) continue;
                    opix = ((size_t)n * p.Ho + oh) * p.Wo + ow;
                }
                const size_t off = opix * p.Co + cbase;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], scl[e], bias[e]);
                if (p.res) {
                    if constexpr (SP) {
                        const char* rp = (const char*)p.res + off * 4;
                        float rr[8];
                        sp_join8(*(const half8*)rp, *(const half8*)(rp + 16), rr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rr[e];
                    } else if constexpr (F32) {
                        const f32x4 r0 = *(const f32x4*)((const float*)p.res + off), r1 = *(const f32x4*)((const float*)p.res + off + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                    } else {
                        const elem8 rv = *(const elem8*)(p.res + off);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                    }
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
                } else if constexpr (SP) {
                    half8 oh, ol;
                    sp_split8(v, oh, ol);
                    char* yo = (char*)p.y + off * 4;
                    *(half8*)yo = oh;
                    *(half8*)(yo + 16) = ol;
                } else {
                    elem8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (elem_t)v[e];
                    *(elem8*)((elem_t*)p.y + off) = o;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    }
    if constexpr (BS) {
        // lane (rsub, cg) holds 16 partial sums (8 channels x {g, g*xhat}) over its rows: transposed through the wave-private
        // staging region ([RPP row lanes][LPR*16 columns] = 4 KiB), every lane then adds one column over the row lanes; the WM
        // wave rows meet in the exchange region behind the staging tiles and ONE slab row per m-tile is stored.
        constexpr int NC = LPR * 16;
        static_assert(RPP * NC * 4 <= ER * ELD * 4, "partial-sum transposition fits the wave's staging region");
        float* tp = est + rsub * NC + cg * 16;
        *(f32x4*)(tp + 0) = (f32x4){bs1[0], bs1[1], bs1[2], bs1[3]};
        *(f32x4*)(tp + 4) = (f32x4){bs1[4], bs1[5], bs1[6], bs1[7]};
        *(f32x4*)(tp + 8) = (f32x4){bs2[0], bs2[1], bs2[2], bs2[3]};
        *(f32x4*)(tp + 12) = (f32x4){bs2[4], bs2[5], bs2[6], bs2[7]};
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        float* xch = (float*)(stage + C::EPI_BYTES);           // [WM][2][BN]
#pragma unroll
        for (int col = lane; col < NC; col += 64) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RPP; ++r) t += est[r * NC + col];
            const int e = col & 15, cl = wn * TN + (col >> 4) * 8 + (e & 7);
            xch[(wm * 2 + (e >> 3)) * BN + cl] = t;
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid % BN, col = n0 + cl;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += xch[(w * 2 + st) * BN + cl];
            const size_t srow = (size_t)cls_id * p.m_tiles + m_tile;
            if (col < p.Co) p.stats[(srow * 2 + st) * p.Co + col] = t;
        }
    }
    if (dbg && tid == 0) {
        dbg[4] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[5] = __builtin_amdgcn_s_memrealtime();
    }
}

template <typename T, int BM, int BN, int WM, int WN, int NS, bool BS = false, int H3 = 0, bool SP = false>
int launch_cfg_t(IgParams& p, hipStream_t stream, const Policy& pol) {
    using C = IgCfg<BM, BN, WM, WN, NS>;
    p.m_tiles = (p.M + BM - 1) / BM;
    p.n_tiles = (p.Co + BN - 1) / BN;
    static std::atomic<unsigned long long> attr_done{0};
    static std::mutex attr_mu;
    once_per_device(attr_done, attr_mu, [] {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<T, BM, BN, WM, WN, NS, BS, H3, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, H3 == 1 ? 112 * 1024 : C::LDS_BYTES);
    });
    dim3 grid(p.m_tiles * p.n_tiles, 1, p.nclass);
    // A launch whose K loop is ONE stage (K <= 128 bytes per row: layer1's 64-channel 1x1 convs and their data gradients, the
    // head's data gradient) only ever touches ring buffer 0: it asks for one stage of LDS instead of NS, so four instead of
    // three work-groups are resident per CU.  These launches are HBM-bound (33-285 MB each) and a work-group's life is a
    // load - compute - store sequence with nothing to overlap inside it: residency is what hides the latency.
    int lds = C::LDS_BYTES;
    if (!(p.flags & IG_FLAG_SMALLC) && pol.igemm_short_lds) {
        constexpr int bke = 128 / (int)sizeof(T);
        int nst = 0;
        for (int c = 0; c < p.nclass; ++c) nst = std::max(nst, p.cls[c].ntaps * p.Ci / bke);
        constexpr int one = C::TAP_BYTES + (C::STAGE1 > C::EPI_BYTES + WM * 2 * BN * 4 ? C::STAGE1 : C::EPI_BYTES + WM * 2 * BN * 4);
        if (nst <= 1 && one < lds) lds = one;
    }
    if constexpr (H3 == 1) {
        // two A buffers of BM + 2(W+1) rows (rounded to 8), three weight slots, the dump piece; at least the epilogue regions
        const int ra = (BM + 2 * (p.Wi + 1) + 7) & ~7;
        lds = C::TAP_BYTES + std::max(2 * (ra + 1) * 128 + 3 * C::BNL * 128 + 1024, C::EPI_BYTES + WM * 2 * BN * 4);
    }
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, NS, BS, H3, SP>), grid, dim3(256), lds, stream, p);
    return udapose_check_launch();
}

template <int BM, int BN, int WM, int WN, int NS>
int launch_cfg(IgParams& p, hipStream_t stream, const Policy& pol) {
    // lean 1x1 form: stride-1 single-tap convolution whose tiles are all full (M % BM, Co % BN), offsets in 32 bits
    const bool split = (p.flags & IG_FLAG_SPLIT) != 0;
    const long long esz = split ? 4 : 2;
    const bool lean = pol.igemm_lean && BN >= 64 && (split || !(p.flags & IG_FLAG_F32)) && !(p.flags & (IG_FLAG_SMALLC | IG_FLAG_REFLECT | IG_FLAG_UPSAMPLE)) && p.tap0 &&
                      p.nclass == 1 && p.cls[0].ntaps == 1 && p.s == 1 && p.os == 1 && p.Hg == p.Hi && p.Wg == p.Wi && p.Hg == p.Ho &&
                      p.Wg == p.Wo && p.M % BM == 0 && p.Co % BN == 0 && p.Ci % 64 == 0 && p.wtaps == 1 &&
                      (long long)p.M * p.Ci * esz < (1ll << 32) && (long long)p.Co * p.Ci * esz < (1ll << 32);
    if (split) {
        // f16x2 operands (fp32-shaped loaders, three fp16 MFMAs per stage): forward epilogue only
        if (!(p.flags & IG_FLAG_F32) || p.bs_y || (!(p.flags & IG_FLAG_OUT_F32) && (p.Co % 8))) return UDAPOSE_ERR_ARG;
        if constexpr (BN >= 64) { if (lean) return launch_cfg_t<float, BM, BN, WM, WN, NS, false, 3, true>(p, stream, pol); }
        return launch_cfg_t<float, BM, BN, WM, WN, NS, false, 0, true>(p, stream, pol);
    }
    if (p.bs_y) {
        // dgrad with the consumer BatchNorm's backward reduction in the epilogue (bf16 operands; bf16 or fp32 output)
        if ((p.flags & (IG_FLAG_F32 | IG_FLAG_SMALLC | IG_FLAG_RELU)) || p.bias || !p.stats || (p.Co % 8)) return UDAPOSE_ERR_ARG;
        if (lean) return launch_cfg_t<elem_t, BM, BN, WM, WN, NS, true, 3>(p, stream, pol);
        return launch_cfg_t<elem_t, BM, BN, WM, WN, NS, true>(p, stream, pol);
    }
    if (lean) return launch_cfg_t<elem_t, BM, BN, WM, WN, NS, false, 3>(p, stream, pol);
    return (p.flags & IG_FLAG_F32) ? launch_cfg_t<float, BM, BN, WM, WN, NS>(p, stream, pol) : launch_cfg_t<elem_t, BM, BN, WM, WN, NS>(p, stream, pol);
}

}  // namespace

// Tile selection (measured on MI355X over every PoseResNet-101 layer shape at N=32, tools/tune_conv.py, then re-tuned under the
// three-stream step): 128x64 tiles with a 2-stage ring as soon as they give two work-groups per CU, else 64x64 (3-stage ring from
// K = 2048).  Tile ids (the numbering of rounds 1-5 is kept; the ids that lost every A/B - 128x128 / 128x64 / 64x64 with deeper rings,
// the register-staged 7 / 8, the three-taps-per-barrier 12, the 256x128 13 / 14 - are gone from the build, tools/experiments/):
// 3 = 128x32 NS3 (heads), 4 = 128x128 NS2 (style network's large maps), 5 = 64x64 NS2, 6 = 128x64 NS2, 9 = 64x64 NS3,
// 10 / 11 = run-staged 3x3 form with 64- / 128-row tiles.
int igemm_pick_tile(int M, int Co, int nclass, int K, int h3_ok, const Policy& pol) {
    if (Co <= 32) return 3;
    if (pol.igemm_tile >= 0) return pol.igemm_tile;
    const int h3 = pol.igemm_h3;
    const long b12864 = (long)((M + 127) / 128) * ((Co + 63) / 64) * nclass;
    // run-staged 3x3 form (measured per shape at N = 32): 64-row tiles where the tap-staged form would take 64x64 (layer3: 21.0 vs
    // 21.9 us, layer4: 27.3 vs 27.9), 128-row tiles where it would take 128x64 and the run fits (layer2: 20.7 vs 22.1); layer1
    // (W = 64: a 194-row run per 64 output rows) stays tap-staged (27.6 vs 35.0)
    if (h3_ok && h3 == 1) { if (b12864 < pol.igemm_wg_min) return 10; if (h3_ok >= 2) return 11; }
    if (h3_ok && h3 == 2) return 10;
    if (h3_ok >= 2 && h3 == 3) return 11;              // (h3_ok >= 2: W <= 32, the 128-row form fits; 3: W <= 16)
    if (h3_ok && h3 == 3) return 10;
    // Every choice lands on THREE resident work-groups per CU (48 KB of LDS each): deeper rings (4 stages = 2 per CU) and one deep
    // 128x64 work-group per CU both measured slower - overlapping the fixed phases of several work-groups beats prefetch depth.
    // single-stream launches with many rounds of work-groups (the style network's 32x32 .. 128x128 maps): 128x128 tiles halve the
    // L2 -> LDS bytes per FLOP of the B operand (655-700 against 555-616 TFLOP/s on its 256-channel layers); Co % 128 != 0 would idle half a tile
    if (pol.igemm_big_min > 0 && nclass == 1 && Co % 128 == 0 && b12864 >= pol.igemm_big_min) return 4;
    if (b12864 >= pol.igemm_wg_min) return 6;
    // (round 4: the 3-stage ring from K = 2048 on, not 1024 - layer3's c1 and the data gradient of its c3, K = 1024, replayed alone from a graph take 8.6 us
    //  with two stages against 9.8 with three, tools/time_l3_convs.py; whole step -0.03 .. -0.14 ms on two boxes, two stages for every K +0.07: r4_ab_runs.txt)
    return K >= (pol.igemm_ns3_k > 0 ? pol.igemm_ns3_k : 2048) ? 9 : 5;
}

int igemm_stat_rows(int M, int Co, int nclass, int tile) {
    switch (tile) {
        case 5: case 9: case 10: return nclass * ((M + 63) / 64);       // one row per m-tile (wave rows added in-kernel)
        default: return nclass * ((M + 127) / 128);
    }
}

int igemm_launch(IgParams& p, int tile, hipStream_t stream, const Policy& pol) {
    p.dbg = pol.timeline;
    if (p.flags & IG_FLAG_TAP0) p.flags &= ~IG_FLAG_TAP0;
    if (pol.igemm_tap0 && p.tap0) p.flags |= IG_FLAG_TAP0;
    const int bke = (p.flags & IG_FLAG_F32) ? 32 : 64;
    if (p.Ci % 8 != 0 || (!(p.flags & IG_FLAG_SMALLC) && p.Ci % bke != 0)) return UDAPOSE_ERR_ARG;
    if ((p.flags & IG_FLAG_SMALLC) && p.Ci != 8) return UDAPOSE_ERR_ARG;
    if (!(p.flags & (IG_FLAG_OUT_F32 | IG_FLAG_F32)) && (p.Co % 8) != 0) return UDAPOSE_ERR_ARG;
    for (int c = 0; c < p.nclass; ++c) {
        if (p.cls[c].ntaps > 64 || p.cls[c].ntaps < 0) return UDAPOSE_ERR_ARG;
        if ((p.flags & IG_FLAG_SMALLC) && (p.cls[c].ntaps % (bke / 8))) return UDAPOSE_ERR_ARG;
    }
    p.div_hw = make_fastdiv((uint32_t)(p.Hg * p.Wg));
    p.div_w = make_fastdiv((uint32_t)p.Wg);
    if (p.flags & IG_FLAG_SPLIT) {
        // f16x2 launches take the plain (or lean 1x1) form: the run-staged 3x3 tiles map to the tap-staged tile of their row count
        if (tile == 11) tile = 6;
        if (tile == 10) tile = 9;
    }
    // the 3x3 stride-1 same-size geometry of the run-staged form (tile ids 10 / 11 are handed out by igemm_pick_tile only when h3_ok): re-checked here
    const bool h3_geo = !(p.flags & (IG_FLAG_F32 | IG_FLAG_SMALLC | IG_FLAG_REFLECT | IG_FLAG_UPSAMPLE)) && p.nclass == 1 && p.cls[0].ntaps == 9 &&
                        p.s == 1 && p.os == 1 && p.Hg == p.Hi && p.Wg == p.Wi && p.Hi == p.Ho && p.Wi == p.Wo && p.Ci % 64 == 0 &&
                        p.cls[0].oa == 0 && p.cls[0].ob == 0;
    switch (tile) {
        case 3: return launch_cfg<128, 32, 4, 1, 3>(p, stream, pol);
        case 4: return launch_cfg<128, 128, 2, 2, 2>(p, stream, pol);
        case 5: return launch_cfg<64, 64, 2, 2, 2>(p, stream, pol);
        case 6: return launch_cfg<128, 64, 2, 2, 2>(p, stream, pol);
        case 9: return launch_cfg<64, 64, 2, 2, 3>(p, stream, pol);
        case 10: {
            if (!(h3_geo && p.Wi <= 64)) return launch_cfg<64, 64, 2, 2, 3>(p, stream, pol);
            if (p.bs_y) {
                if ((p.flags & IG_FLAG_RELU) || p.bias || !p.stats || (p.Co % 8)) return UDAPOSE_ERR_ARG;
                return launch_cfg_t<elem_t, 64, 64, 2, 2, 3, true, 1>(p, stream, pol);
            }
            return launch_cfg_t<elem_t, 64, 64, 2, 2, 3, false, 1>(p, stream, pol);
        }
        case 11: {
            // the same with 128-row tiles (the 9 weight tiles of a chunk serve twice the rows); run of 128 + 2(W+1) rows <= 28 pieces
            if (!(h3_geo && p.Wi <= 32)) return launch_cfg<128, 64, 2, 2, 2>(p, stream, pol);
            if (p.bs_y) {
                if ((p.flags & IG_FLAG_RELU) || p.bias || !p.stats || (p.Co % 8)) return UDAPOSE_ERR_ARG;
                return launch_cfg_t<elem_t, 128, 64, 2, 2, 2, true, 1>(p, stream, pol);
            }
            return launch_cfg_t<elem_t, 128, 64, 2, 2, 2, false, 1>(p, stream, pol);
        }
        default: return UDAPOSE_ERR_ARG;
    }
}

UDAPOSE_SP_SAT_READER(sp_sat_read_igemm)
