// Dev probe (round 4): what does the conv -> train-mode BatchNorm seam cost on this chip, kept inside a launch (grid barrier, apply from
// the accumulators in registers) against cut into launches (conv with partial statistics | one-launch finalize + apply), on the geometry
// where the step is latency-bound: layer3 of PoseResNet-101 at N = 32 (M = 8192 output pixels of 16 x 16 maps; bottleneck c1 1024 -> 256,
// c2 3x3 256 -> 256, c3 256 -> 1024; 23 blocks = 69 conv + BN).  VERDICT r3 next #3 asked for this accounting with measurements.
// Three forms of the SAME arithmetic on the same tiles (bf16 MFMA 16x16x32, LDS-DMA ring, XOR-swizzled 128-byte rows, XCD-aware tile order):
//   U  two launches per layer: gemm_k (y in bf16 + one slab row of column sums per m-tile) | bn_k (column sums of the slab in fp64 ->
//      scale / shift -> z = relu(y * scale + shift)): the production structure (igemm_kernel + bn_apply_chunk_k)
//   F  one launch per layer: tile -> slab row -> XCD-hierarchical grid barrier -> column sums -> apply from the accumulators -> z
//   P  one launch for the whole chain: F's body per layer + a second grid barrier per layer (z must be visible to the next layer's loads)
//   N  "normalize on load": bn1 / bn2 have no launch - the consumer convolution applies the input's BatchNorm + ReLU while staging its A operand through
//      registers, scale / shift computed by the producer's last work-group to arrive (3 conv + 1 BatchNorm launch per bottleneck instead of 6 launches)
// Every spin is bounded (give-up flag, results then wrong and reported).  The 3x3 layer reads its nine taps as row shifts of the same
// tensor modulo M (borders ignored: a probe of time, not a convolution of images).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/conv_bn_seam tools/probe/conv_bn_seam.hip
//   run:   timeout -k 10 250 tools/probe/conv_bn_seam [blocks = 23] [cold weights 0/1] [weight prefetch in U's BatchNorm launch 0/1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define M_ROWS 8192          // N = 32 images x 16 x 16
#define MAP_W 16

struct Layer {
    const bf16* x;  // [M][Kc]
    const bf16* w;  // [N][taps * Kc]
    void* y;        // [M][N]  (U only)   bf16, or 4-byte elements in the W4 forms
    void* z;        // [M][N]
    float* slab;    // [m_tiles][2][N]
    const float* gamma; const float* beta;
    int Kc, N, taps;
    const void* next_w; int next_w_bytes;      // (U, prefetch experiment) the NEXT layer's weights: the BatchNorm launch pulls them into every XCD's L2
};
struct GridBar { unsigned int cnt[8][32]; unsigned int top[32]; unsigned int gen[8][32]; unsigned int err[32]; };

__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t total) {
    const uint32_t q = total >> 3, r = total & 7u, xcd = bid & 7u, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// XCD-hierarchical barrier: arrivals counted per XCD, the XCD's last arriver adds to the top counter, the chip's last arriver publishes the
// generation word of every XCD, everybody polls the word of the XCD it runs on.  Counters are monotonic (epoch = barriers passed so far).
__device__ __forceinline__ void grid_barrier(GridBar* b, unsigned int epoch, unsigned int per_xcc) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int xcc = (unsigned int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
        // (the work-group's stores have reached its XCD's L2: __syncthreads waited for them.  Work-groups of one XCD share that L2, so only
        //  the XCD's last arriver writes it back - one release fence per XCD instead of one per work-group)
        const unsigned int a = __hip_atomic_fetch_add(&b->cnt[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (a == (epoch + 1u) * per_xcc) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned int t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (t == (epoch + 1u) * 8u)
                for (int x = 0; x < 8; ++x) __hip_atomic_store(&b->gen[x][0], epoch + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned int spins = 0;
        while (__hip_atomic_load(&b->gen[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1u) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023u) == 1023u && __hip_atomic_load(&b->err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;    // somebody gave up already
            if (++spins > (1u << 18)) { __hip_atomic_store(&b->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // give up: results wrong, reported
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N == 0, "vmcnt value");
}

template <int BM, int BN, int NS, bool W4 = false>
struct Cfg {
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int PIECES = (BM + BN) / 8, PPW = PIECES / 4;
    static constexpr int ES = W4 ? 4 : 2, OUT_STRIDE = BN * ES + 16;                   // staged output tile: padded rows
    static constexpr int RING = NS * STAGE, OUT = BM * OUT_STRIDE;
    static constexpr int LDS = (RING > OUT ? RING : OUT) + 4096;      // + coefficients / wave-row scratch
    static constexpr int MI = BM / 32, NJ = BN / 32;                  // 16x16 blocks per wave (waves 2 x 2)
};

// the K loop of one tile: acc = X[m0.., taps] * W[n0..]^T
template <int BM, int BN, int NS>
__device__ __forceinline__ void conv_tile(const Layer& L, int m0, int n0, char* smem, f32x4 (&acc)[BM / 32][BN / 32]) {
    using C = Cfg<BM, BN, NS>;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1, l15 = lane & 15, g = lane >> 4;
    const int kpt = L.Kc / 64, nks = kpt * L.taps, Ktot = L.Kc * L.taps;
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int ks, int buf) {
        const int tap = ks / kpt, kc = (ks - tap * kpt) * 64;
        const int shift = L.taps == 1 ? 0 : ((tap / 3 - 1) * MAP_W + (tap % 3 - 1));
        char* sbase = smem + buf * C::STAGE;
#pragma unroll
        for (int p = 0; p < C::PPW; ++p) {
            const int piece = p * 4 + wid, r8 = lane >> 3, ch = (lane & 7) ^ r8;
            const bf16* src;
            if (piece < BM / 8) {
                const int m = (m0 + piece * 8 + r8 + shift) & (M_ROWS - 1);
                src = L.x + (size_t)m * L.Kc + kc + ch * 8;
            } else {
                const int n = n0 + (piece - BM / 8) * 8 + r8;
                src = L.w + (size_t)n * Ktot + ks * 64 + ch * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sbase + piece * 1024), 16, 0, 0);
        }
    };
    for (int b = 0; b < NS - 1; ++b) if (b < nks) issue(b, b);
    for (int ks = 0; ks < nks; ++ks) {
        if (ks + NS - 1 <= nks) wait_vmcnt<(NS - 2) * C::PPW>(); else wait_vmcnt<0>();     // (the tail drains)
        __builtin_amdgcn_s_barrier();
        if (ks + NS - 1 < nks) issue(ks + NS - 1, (ks + NS - 1) % NS);
        const char* sA = smem + (ks % NS) * C::STAGE;
        const char* sB = sA + C::A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int coff = ((s * 4 + g) ^ (l15 & 7)) * 16;
            bf16x8 a[C::MI], b[C::NJ];
#pragma unroll
            for (int i = 0; i < C::MI; ++i) a[i] = *(const bf16x8*)(sA + (wm * (BM / 2) + i * 16 + l15) * 128 + coff);
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) b[j] = *(const bf16x8*)(sB + (wn * (BN / 2) + j * 16 + l15) * 128 + coff);
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
#pragma unroll
                for (int j = 0; j < C::NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();       // the ring is free for the epilogue
}

// one slab row per m-tile: column sums and sums of squares of the fp32 accumulators
template <int BM, int BN, int NS, bool W4>
__device__ __forceinline__ void tile_stats(const Layer& L, int m_tile, int n0, char* smem, const f32x4 (&acc)[BM / 32][BN / 32], bool coherent = false) {
    using C = Cfg<BM, BN, NS, W4>;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1, l15 = lane & 15;
    float* red = (float*)(smem + C::LDS - 4096);       // [2 wm][2][BN] floats <= 4 KB for BN <= 256
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (lane < 16) {
            const int col = wn * (BN / 2) + j * 16 + l15;
            red[(wm * 2 + 0) * BN + col] = s;
            red[(wm * 2 + 1) * BN + col] = q;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * BN; t += 256) {
        const int which = t / BN, col = t - which * BN;
        const float v = red[(0 * 2 + which) * BN + col] + red[(1 * 2 + which) * BN + col];
        float* dst = L.slab + ((size_t)m_tile * 2 + which) * L.N + n0 + col;
        // (the last-arriver forms publish the row with device-coherent write-through stores: visible to another XCD without writing the L2 back)
        if (coherent) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = v;
    }
}

// column sums of the slab (fp64) -> scale / shift of this tile's BN channels, left in LDS (sc[BN], sh[BN])
template <int BN>
__device__ __forceinline__ void col_coeffs(const Layer& L, int m_tiles, int n0, char* scratch /* >= 16 * BN + 8 * BN bytes */) {
    constexpr int QUADS = BN / 4, RG = 256 / (2 * QUADS);     // row groups
    double* part = (double*)scratch;                         // [RG][2][BN]  (RG * 2 * BN * 8 = 4096 * ... bytes: 16 KB for any BN here)
    float* sc = (float*)(scratch + (size_t)RG * 2 * BN * 8);
    float* sh = sc + BN;
    const int t = threadIdx.x, quad = t % QUADS, which = (t / QUADS) & 1, rg = t / (2 * QUADS);
    const float* base = L.slab + (size_t)which * L.N + n0 + quad * 4;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int r = rg; r < m_tiles; r += RG * 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (r + k * RG < m_tiles) ? *(const f32x4*)(base + (size_t)(r + k * RG) * 2 * L.N) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) { a0 += v[k][0]; a1 += v[k][1]; a2 += v[k][2]; a3 += v[k][3]; }
    }
    part[(rg * 2 + which) * BN + quad * 4 + 0] = a0; part[(rg * 2 + which) * BN + quad * 4 + 1] = a1;
    part[(rg * 2 + which) * BN + quad * 4 + 2] = a2; part[(rg * 2 + which) * BN + quad * 4 + 3] = a3;
    __syncthreads();
    for (int c = t; c < BN; c += 256) {
        double s = 0, q = 0;
        for (int r = 0; r < RG; ++r) { s += part[(r * 2 + 0) * BN + c]; q += part[(r * 2 + 1) * BN + c]; }
        const double mean = s / M_ROWS, var = q / M_ROWS - mean * mean;
        const float invstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + 1e-5));
        const float scale = L.gamma[n0 + c] * invstd;
        sc[c] = scale; sh[c] = L.beta[n0 + c] - (float)mean * scale;
    }
    __syncthreads();
}

// accumulators (optionally BN + ReLU applied) -> bf16 tile through LDS -> 16-byte stores
template <int BM, int BN, int NS, bool APPLY, bool W4>
__device__ __forceinline__ void store_tile(void* out_, int ldn, int m0, int n0, char* smem, const f32x4 (&acc)[BM / 32][BN / 32], const float* sc, const float* sh) {
    using C = Cfg<BM, BN, NS, W4>;
    char* out = (char*)out_;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1, l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
        const int col = wn * (BN / 2) + j * 16 + l15;
        float a = 1.f, b = 0.f;
        if constexpr (APPLY) { a = sc[col]; b = sh[col]; }
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if constexpr (APPLY) v = fmaxf(__builtin_fmaf(v, a, b), 0.f);
                const int row = wm * (BM / 2) + i * 16 + g * 4 + r;
                if constexpr (W4) *(float*)(smem + row * C::OUT_STRIDE + col * 4) = v;
                else *(bf16*)(smem + row * C::OUT_STRIDE + col * 2) = (bf16)v;
            }
    }
    __syncthreads();
    constexpr int CPR = BN * C::ES / 16, CHUNKS = BM * CPR;
    for (int q = threadIdx.x; q < CHUNKS; q += 256) {
        const int row = q / CPR, cc = q - row * CPR;
        *(uint4*)(out + ((size_t)(m0 + row) * ldn + n0) * C::ES + cc * 16) = *(const uint4*)(smem + row * C::OUT_STRIDE + cc * 16);
    }
}

template <int BM, int BN>
__device__ __forceinline__ void tile_of(int total, int n_tiles, int& m_tile, int& n_tile) {
    const uint32_t w = xcd_remap(blockIdx.x, total);
    m_tile = (int)(w / n_tiles); n_tile = (int)(w - m_tile * n_tiles);
}

// U, first launch: conv + statistics + y
template <int BM, int BN, int NS, bool W4>
__global__ __launch_bounds__(256) void gemm_k(const Layer L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = Cfg<BM, BN, NS, W4>;
    int m_tile, n_tile;
    tile_of<BM, BN>(gridDim.x, L.N / BN, m_tile, n_tile);
    f32x4 acc[C::MI][C::NJ];
    conv_tile<BM, BN, NS>(L, m_tile * BM, n_tile * BN, smem, acc);
    tile_stats<BM, BN, NS, W4>(L, m_tile, n_tile * BN, smem, acc);
    store_tile<BM, BN, NS, false, W4>(L.y, L.N, m_tile * BM, n_tile * BN, smem, acc, nullptr, nullptr);
}
// U, second launch: slab -> coefficients -> z = relu(y * scale + shift); one work-group per 128 x 64 piece (bn_apply_chunk_k's shape)
template <bool W4>
__global__ __launch_bounds__(256) void bn_k(const Layer L, int m_tiles_slab) {
    __shared__ __attribute__((aligned(16))) char scratch[16 * 1024 + 512];
    const int n_chunks = L.N / 64;
    const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = (int)(w / n_chunks), n0 = (int)(w - mt * n_chunks) * 64, m0 = mt * 128;
    // prefetch experiment: the work-groups of each XCD (blockIdx & 7 up to a per-launch rotation) together touch the whole next weight tensor once,
    // requested before the statistics prelude and consumed after the last store
    uint4 pf = {0u, 0u, 0u, 0u};
    if (L.next_w) {
        const int per_xcd = (int)(gridDim.x >> 3), local = (int)(blockIdx.x >> 3), total16 = L.next_w_bytes >> 4;
        const int chunk = (total16 + per_xcd - 1) / per_xcd;
        const uint4* src = (const uint4*)L.next_w + (size_t)local * chunk;
        for (int i = threadIdx.x; i < chunk && local * chunk + i < total16; i += 256) { const uint4 v = src[i]; pf.x ^= v.x; pf.y ^= v.y; pf.z ^= v.z; pf.w ^= v.w; }
    }
    col_coeffs<64>(L, m_tiles_slab, n0, scratch);
    const float* sc = (const float*)(scratch + 8 * 2 * 64 * 8);
    const float* sh = sc + 64;
    if constexpr (W4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = threadIdx.x + k * 256, row = q >> 4, cc = q & 15;
            const size_t off = (size_t)(m0 + row) * L.N + n0 + cc * 4;
            const f32x4 v = *(const f32x4*)((const float*)L.y + off);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(__builtin_fmaf(v[e], sc[cc * 4 + e], sh[cc * 4 + e]), 0.f);
            *(f32x4*)((float*)L.z + off) = o;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = threadIdx.x + k * 256, row = q >> 3, cc = q & 7;
            const size_t off = (size_t)(m0 + row) * L.N + n0 + cc * 8;
            const bf16x8 v = *(const bf16x8*)((const bf16*)L.y + off);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(__builtin_fmaf((float)v[e], sc[cc * 8 + e], sh[cc * 8 + e]), 0.f);
            *(bf16x8*)((bf16*)L.z + off) = o;
        }
    }
    if ((pf.x ^ pf.y ^ pf.z ^ pf.w) == 0x9e3779b9u) ((unsigned int*)L.slab)[0] = 1u;     // (keeps the prefetch loads alive; practically never true)
}
// F: conv + statistics + grid barrier + apply from the accumulators
template <int BM, int BN, int NS, bool W4>
__device__ __forceinline__ void fused_body(const Layer& L, int total, char* smem, GridBar* bar, unsigned int epoch) {
    using C = Cfg<BM, BN, NS, W4>;
    int m_tile, n_tile;
    tile_of<BM, BN>(total, L.N / BN, m_tile, n_tile);
    f32x4 acc[C::MI][C::NJ];
    conv_tile<BM, BN, NS>(L, m_tile * BM, n_tile * BN, smem, acc);
    tile_stats<BM, BN, NS, W4>(L, m_tile, n_tile * BN, smem, acc);
    grid_barrier(bar, epoch, (unsigned int)total / 8u);
    char* scratch = smem;                            // column sums at the head of the ring (free after the K loop)
    col_coeffs<BN>(L, M_ROWS / BM, n_tile * BN, scratch);
    constexpr int RG = 256 / (2 * (BN / 4));
    // copy the coefficients out of the region the staged tile is about to overwrite
    float* keep = (float*)(smem + C::LDS - 4096);
    const float* sc = (const float*)(scratch + (size_t)RG * 2 * BN * 8);
    for (int c = threadIdx.x; c < 2 * BN; c += 256) keep[c] = sc[c];
    __syncthreads();
    store_tile<BM, BN, NS, true, W4>(L.z, L.N, m_tile * BM, n_tile * BN, smem, acc, keep, keep + BN);
}
template <int BM, int BN, int NS, bool W4>
__global__ __launch_bounds__(256) void fused_k(const Layer L, GridBar* bar, unsigned int epoch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fused_body<BM, BN, NS, W4>(L, gridDim.x, smem, bar, epoch);
}
// P: the whole chain in one launch (256 work-groups; c1 / c2 as 128 x 64 tiles, c3 as 128 x 256), two barriers per layer
template <bool W4>
__global__ __launch_bounds__(256) void persist_k(const Layer* layers, int n_layers, GridBar* bar, unsigned int epoch0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int epoch = epoch0;
    for (int l = 0; l < n_layers; ++l) {
        const Layer L = layers[l];
        if (L.N == 1024) fused_body<128, 256, 2, W4>(L, gridDim.x, smem, bar, epoch);
        else fused_body<128, 64, 2, W4>(L, gridDim.x, smem, bar, epoch);
        ++epoch;
        grid_barrier(bar, epoch, gridDim.x / 8u);      // z of this layer visible before anybody stages it
        ++epoch;
    }
}

// the barrier alone: n barriers back to back (what the guide's price list calls barrier-xcd, host-paired)
__global__ __launch_bounds__(256) void bar_only_k(GridBar* bar, int n, unsigned int epoch0) {
    for (int i = 0; i < n; ++i) grid_barrier(bar, epoch0 + (unsigned int)i, gridDim.x / 8u);
}

// ---- N: "normalize on load" - the BatchNorm + ReLU of a convolution's INPUT applied while the consumer stages its A operand (registers -> transform -> LDS
// instead of LDS-DMA), the scale / shift computed by the LAST work-group of the producer's n-tile to arrive (no finalize launch): bn1 and bn2 of a bottleneck
// have no launch of their own; bn3 (residual in the real network) keeps its launch.  What a backward pass needs - z1, z2 - is written as a side output by the
// consumer's n-tile-0 work-groups (optional: a teacher pass keeps nothing).
struct LayerN {
    Layer L;                                  // L.x = the RAW bf16 conv output of the previous layer when sc_in is set, a finished activation otherwise
    const float* sc_in; const float* sh_in;   // [Kc] scale / shift of the input's BatchNorm
    float* sc_out; float* sh_out;             // [N] written by the last-arriving work-group of each n-tile (null: this layer's BatchNorm is a launch)
    unsigned int* ticket;                     // [n_tiles][32] arrival counters, monotonic (m_tiles arrivals per launch and n-tile)
    bf16* zside;                              // [M][Kc] the transformed input kept for a backward pass (null: not kept)
    int fence_release;                        // 1: iteration-1 form (plain slab stores + a release fence per work-group); 0: write-through slab stores, no fence
};

template <int BM, int BN>
__device__ __forceinline__ void conv_tile_rs(const LayerN& P, int m0, int n0, int n_tile, char* smem, f32x4 (&acc)[BM / 32][BN / 32]) {
    using C = Cfg<BM, BN, 2>;
    const Layer& L = P.L;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1, l15 = lane & 15, g = lane >> 4;
    const int kpt = L.Kc / 64, nks = kpt * L.taps, Ktot = L.Kc * L.taps;
    constexpr int ACH = BM * 8 / 256;          // 16-byte A chunks per thread and stage
    constexpr int BPW = BN / 8 / 4;            // B pieces per wave and stage
    float* csc = (float*)(smem + C::LDS);      // the input BatchNorm's coefficients, all Kc channels
    float* csh = csc + L.Kc;
    for (int i = threadIdx.x; i < L.Kc; i += 256) { csc[i] = P.sc_in[i]; csh[i] = P.sh_in[i]; }
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ra[ACH];
    auto tap_of = [&](int ks, int& kc, int& shift, bool& centre) {
        const int tap = ks / kpt;
        kc = (ks - tap * kpt) * 64;
        shift = L.taps == 1 ? 0 : ((tap / 3 - 1) * MAP_W + (tap % 3 - 1));
        centre = L.taps == 1 || tap == 4;
    };
    auto loadA = [&](int ks) {
        int kc, shift; bool centre;
        tap_of(ks, kc, shift, centre);
#pragma unroll
        for (int a = 0; a < ACH; ++a) {
            const int q = threadIdx.x + 256 * a, row = q >> 3, ch = q & 7;
            const int m = (m0 + row + shift) & (M_ROWS - 1);
            ra[a] = *(const bf16x8*)(L.x + (size_t)m * L.Kc + kc + ch * 8);
        }
    };
    auto issueB = [&](int ks, int buf) {
        char* sB = smem + buf * C::STAGE + C::A_BYTES;
#pragma unroll
        for (int p = 0; p < BPW; ++p) {
            const int piece = p * 4 + wid, r8 = lane >> 3, ch = (lane & 7) ^ r8;
            const int n = n0 + piece * 8 + r8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(L.w + (size_t)n * Ktot + ks * 64 + ch * 8),
                                             (__attribute__((address_space(3))) void*)(sB + piece * 1024), 16, 0, 0);
        }
    };
    auto xformA = [&](int ks, int buf) {
        int kc, shift; bool centre;
        tap_of(ks, kc, shift, centre);
        char* sA = smem + buf * C::STAGE;
#pragma unroll
        for (int a = 0; a < ACH; ++a) {
            const int q = threadIdx.x + 256 * a, row = q >> 3, ch = q & 7;
            const f32x4 s0 = *(const f32x4*)(csc + kc + ch * 8), s1 = *(const f32x4*)(csc + kc + ch * 8 + 4);
            const f32x4 h0 = *(const f32x4*)(csh + kc + ch * 8), h1 = *(const f32x4*)(csh + kc + ch * 8 + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (bf16)fmaxf(__builtin_fmaf((float)ra[a][e], s0[e], h0[e]), 0.f);
                o[e + 4] = (bf16)fmaxf(__builtin_fmaf((float)ra[a][e + 4], s1[e], h1[e]), 0.f);
            }
            *(bf16x8*)(sA + row * 128 + ((ch ^ (row & 7)) * 16)) = o;
            if (P.zside && centre && n_tile == 0) *(bf16x8*)(P.zside + (size_t)(m0 + row) * L.Kc + kc + ch * 8) = o;
        }
    };
    loadA(0); issueB(0, 0);
    __syncthreads();                     // (the coefficient table)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xformA(0, 0);
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        if (ks + 1 < nks) { loadA(ks + 1); issueB(ks + 1, (ks + 1) & 1); }
        const char* sA = smem + (ks & 1) * C::STAGE;
        const char* sB = sA + C::A_BYTES;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int coff = ((s2 * 4 + g) ^ (l15 & 7)) * 16;
            bf16x8 a[C::MI], b[C::NJ];
#pragma unroll
            for (int i = 0; i < C::MI; ++i) a[i] = *(const bf16x8*)(sA + (wm * (BM / 2) + i * 16 + l15) * 128 + coff);
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) b[j] = *(const bf16x8*)(sB + (wn * (BN / 2) + j * 16 + l15) * 128 + coff);
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
#pragma unroll
                for (int j = 0; j < C::NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            xformA(ks + 1, (ks + 1) & 1);
        }
        __syncthreads();
    }
}

// the last work-group of an n-tile to arrive turns the slab's columns into that n-tile's scale / shift
template <int BM, int BN>
__device__ __forceinline__ void finalize_last(const LayerN& P, int n_tile, int n0, char* smem) {
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (design iteration 1 ran an agent-scope release fence here in every work-group - an L2 write-back each: +13 us per launch at 512 work-groups, +26 at
        //  1024.  Iteration 2: the slab row was stored write-through (tile_stats, coherent) and __syncthreads has waited for those stores: no fence.)
        if (P.fence_release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned int t = __hip_atomic_fetch_add(&P.ticket[n_tile * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ((t + 1u) % (unsigned int)(M_ROWS / BM)) == 0u;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
    col_coeffs<BN>(P.L, M_ROWS / BM, n0, smem);
    constexpr int RG = 256 / (2 * (BN / 4));
    const float* sc = (const float*)(smem + (size_t)RG * 2 * BN * 8);
    for (int c = threadIdx.x; c < BN; c += 256) { P.sc_out[n0 + c] = sc[c]; P.sh_out[n0 + c] = sc[BN + c]; }
}

template <int BM, int BN, int NS, bool XF>
__global__ __launch_bounds__(256) void gemm_n_k(const LayerN P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = Cfg<BM, BN, NS>;
    int m_tile, n_tile;
    tile_of<BM, BN>(gridDim.x, P.L.N / BN, m_tile, n_tile);
    f32x4 acc[C::MI][C::NJ];
    if constexpr (XF) conv_tile_rs<BM, BN>(P, m_tile * BM, n_tile * BN, n_tile, smem, acc);
    else conv_tile<BM, BN, NS>(P.L, m_tile * BM, n_tile * BN, smem, acc);
    tile_stats<BM, BN, NS, false>(P.L, m_tile, n_tile * BN, smem, acc, P.sc_out != nullptr && !P.fence_release);
    store_tile<BM, BN, NS, false, false>(P.L.y, P.L.N, m_tile * BM, n_tile * BN, smem, acc, nullptr, nullptr);
    if (P.sc_out) finalize_last<BM, BN>(P, n_tile, n_tile * BN, smem);
}
template <int BM, int BN, int NS, bool XF>
static void launch_gemm_n(const LayerN& P, hipStream_t st) {
    constexpr int lds = Cfg<BM, BN, NS>::LDS + (XF ? 8192 : 0);
    static bool once = false;
    if (!once) { hipFuncSetAttribute((const void*)gemm_n_k<BM, BN, NS, XF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); once = true; }
    hipLaunchKernelGGL((gemm_n_k<BM, BN, NS, XF>), dim3((M_ROWS / BM) * (P.L.N / BN)), dim3(256), lds, st, P);
}

static void fill_bf16(std::vector<bf16>& v, unsigned seed, float scale) {
    unsigned s = seed;
    for (auto& e : v) { s = s * 1664525u + 1013904223u; e = (bf16)(((int)(s >> 9) % 2001 - 1000) * 0.001f * scale); }
}

template <int BM, int BN, int NS, bool W4>
static void launch_gemm(const Layer& L, hipStream_t st) {
    using C = Cfg<BM, BN, NS, W4>;
    static bool once = false;
    if (!once) { hipFuncSetAttribute((const void*)gemm_k<BM, BN, NS, W4>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); once = true; }
    hipLaunchKernelGGL((gemm_k<BM, BN, NS, W4>), dim3((M_ROWS / BM) * (L.N / BN)), dim3(256), C::LDS, st, L);
}
template <int BM, int BN, int NS, bool W4>
static void launch_fused(const Layer& L, hipStream_t st, GridBar* bar, unsigned epoch) {
    using C = Cfg<BM, BN, NS, W4>;
    static bool once = false;
    if (!once) { hipFuncSetAttribute((const void*)fused_k<BM, BN, NS, W4>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); once = true; }
    hipLaunchKernelGGL((fused_k<BM, BN, NS, W4>), dim3((M_ROWS / BM) * (L.N / BN)), dim3(256), C::LDS, st, L, bar, epoch);
}

struct Ctx {
    hipStream_t st;
    std::vector<Layer> hl; Layer* d_layers; int n_layers;
    bf16* zA; GridBar* bar;
    std::vector<bf16> input;
    char* evict = nullptr;       // cold-weights mode: 1 GiB written between repetitions (nothing of the previous repetition left in the Infinity Cache)
    void reset_input() {
        if (evict) hipMemset(evict, 1, (size_t)1 << 30);
        hipMemcpy(zA, input.data(), input.size() * 2, hipMemcpyHostToDevice);
    }
};

template <typename F>
static void time_it(Ctx& c, const char* name, F&& enqueue, int reps, GridBar* errs, int n_errs) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double best = 1e30;
    for (int r = 0; r < reps + 2; ++r) {
        c.reset_input();
        hipDeviceSynchronize();
        hipEventRecord(a, c.st);
        enqueue();
        hipEventRecord(b, c.st);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (r >= 2 && ms < best) best = ms;
    }
    unsigned err = 0;
    for (int k = 0; k < n_errs; ++k) { unsigned e = 0; hipMemcpy(&e, &errs[k].err[0], 4, hipMemcpyDeviceToHost); err |= e; }
    printf("%-78s | %8.1f us per chain | %6.2f us per conv+BN%s\n", name, best * 1e3, best * 1e3 / c.n_layers, err ? "  ** BARRIER GAVE UP: invalid **" : "");
    fflush(stdout);
    hipEventDestroy(a); hipEventDestroy(b);
}
template <typename F>
static hipGraphExec_t graph_of(hipStream_t st, F&& enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    enqueue();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    return ge;
}

template <bool W4>
static void run_forms(Ctx& c) {
    hipStream_t st = c.st;
    constexpr int kLds = Cfg<128, 256, 2, W4>::LDS;
    hipFuncSetAttribute((const void*)persist_k<W4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    auto bn = [&](const Layer& L, int m_tiles_slab) { hipLaunchKernelGGL((bn_k<W4>), dim3((M_ROWS / 128) * (L.N / 64)), dim3(256), 0, st, L, m_tiles_slab); };
    auto chain_U = [&](int small, int c3wide) {
        for (const Layer& L : c.hl) {
            if (L.N == 1024) { if (c3wide) launch_gemm<128, 256, 2, W4>(L, st); else launch_gemm<128, 64, 2, W4>(L, st); }
            else if (small) launch_gemm<64, 64, 3, W4>(L, st); else launch_gemm<128, 64, 2, W4>(L, st);
            bn(L, M_ROWS / ((L.N != 1024 && small) ? 64 : 128));
        }
    };
    chain_U(0, 0); chain_U(1, 1); hipStreamSynchronize(st);       // (function attributes set outside any capture)
    printf("## %s\n", W4 ? "4-byte y and z (the byte traffic of the f16x2 / fp32 forms: y fp32, z split; the K loop stays the bf16 one)" : "bf16 y and z (the benched configuration)");
    // ---- the barrier forms compute what the two-launch form computes: one block (3 layers), z of c3
    if (!W4) {
        std::vector<bf16> ref((size_t)M_ROWS * 1024), got((size_t)M_ROWS * 1024);
        auto run3 = [&](int form) {
            c.reset_input();
            hipMemset(c.bar, 0, sizeof(GridBar));
            for (int l = 0; l < 3; ++l) {
                const Layer& L = c.hl[l];
                if (form == 0) {
                    if (L.N == 1024) launch_gemm<128, 256, 2, W4>(L, st); else launch_gemm<128, 64, 2, W4>(L, st);
                    bn(L, M_ROWS / 128);
                } else if (form == 1) {
                    if (L.N == 1024) launch_fused<128, 256, 2, W4>(L, st, c.bar, (unsigned)l); else launch_fused<128, 64, 2, W4>(L, st, c.bar, (unsigned)l);
                }
            }
            if (form == 2) hipLaunchKernelGGL((persist_k<W4>), dim3(256), dim3(256), kLds, st, c.d_layers, 3, c.bar, 0u);
            hipStreamSynchronize(st);
        };
        run3(0); hipMemcpy(ref.data(), c.zA, ref.size() * 2, hipMemcpyDeviceToHost);
        for (int form = 1; form <= 2; ++form) {
            run3(form); hipMemcpy(got.data(), c.zA, got.size() * 2, hipMemcpyDeviceToHost);
            double md = 0, mx = 0; size_t nz = 0;
            for (size_t i = 0; i < ref.size(); ++i) { md = fmax(md, fabs((double)(float)ref[i] - (double)(float)got[i])); mx = fmax(mx, fabs((double)(float)ref[i])); nz += (float)got[i] != 0.f; }
            unsigned err = 0; hipMemcpy(&err, &c.bar->err[0], 4, hipMemcpyDeviceToHost);
            printf("# check %s vs two launches after one block: max |dz| %.4f of max |z| %.3f (y rounded to bf16 on one side only), %zu non-zero of %zu%s\n",
                   form == 1 ? "F (one launch per layer)" : "P (one launch)", md, mx, nz, ref.size(), err ? "  ** BARRIER GAVE UP **" : "");
        }
    }
    // ---- timing
    for (int small = 0; small < 2; ++small)
        for (int c3wide = 0; c3wide < 2; ++c3wide) {
            char name[200];
            snprintf(name, sizeof name, "U two launches per layer; c1/c2 %s, c3 %s (graph replay)", small ? "64x64 x512" : "128x64 x256", c3wide ? "128x256 x256" : "128x64 x1024");
            hipGraphExec_t ge = graph_of(st, [&]() { chain_U(small, c3wide); });
            time_it(c, name, [&]() { hipGraphLaunch(ge, st); }, 6, c.bar, 0);
            hipGraphExecDestroy(ge);
        }
    for (int small = 0; small < 2; ++small) {
        // F: each launch passes one barrier on counters of its own grid size: a counter set per layer kind, epochs counted per set (eager launches:
        // the epoch is a kernel argument)
        GridBar* bars; hipMalloc((void**)&bars, sizeof(GridBar) * 3); hipMemset(bars, 0, sizeof(GridBar) * 3);
        unsigned ep[3] = {0, 0, 0};
        char name[200];
        snprintf(name, sizeof name, "F one launch per layer (barrier inside); c1/c2 %s, c3 128x256 x256", small ? "64x64 x512" : "128x64 x256");
        time_it(c, name, [&]() {
            int l = 0;
            for (const Layer& L : c.hl) {
                const int kind = l % 3; ++l;
                if (L.N == 1024) launch_fused<128, 256, 2, W4>(L, st, bars + kind, ep[kind]);
                else if (small) launch_fused<64, 64, 3, W4>(L, st, bars + kind, ep[kind]); else launch_fused<128, 64, 2, W4>(L, st, bars + kind, ep[kind]);
                ++ep[kind];
            }
        }, 6, bars, 3);
        hipFree(bars);
    }
    if (!W4) {
        // N: normalize on load (bn1 / bn2 without a launch of their own)
        float* coef; unsigned int* tick;
        hipMalloc((void**)&coef, 4 * 1024 * sizeof(float)); hipMalloc((void**)&tick, 2 * 16 * 32 * sizeof(unsigned int));
        hipMemset(tick, 0, 2 * 16 * 32 * sizeof(unsigned int));
        float *sc1 = coef, *sh1 = coef + 1024, *sc2 = coef + 2048, *sh2 = coef + 3072;
        auto chain_N = [&](int nblocks, bool keep) {
            for (int b = 0; b < nblocks; ++b) {
                const Layer &l1 = c.hl[3 * b], &l2 = c.hl[3 * b + 1], &l3 = c.hl[3 * b + 2];
                LayerN p1{l1, nullptr, nullptr, sc1, sh1, tick, nullptr, 0};
                LayerN p2{l2, sc1, sh1, sc2, sh2, tick + 16 * 32, keep ? (bf16*)l1.z : nullptr, 0};
                p2.L.x = (const bf16*)l1.y;                   // the raw output of c1
                LayerN p3{l3, sc2, sh2, nullptr, nullptr, nullptr, keep ? (bf16*)l2.z : nullptr, 0};
                p3.L.x = (const bf16*)l2.y;
                launch_gemm_n<64, 64, 3, false>(p1, st);
                launch_gemm_n<64, 64, 2, true>(p2, st);
                launch_gemm_n<128, 64, 2, true>(p3, st);
                bn(l3, M_ROWS / 128);
            }
        };
        {   // one block against the two-launch form (same formulas, same order of sums: expected identical)
            std::vector<bf16> ref((size_t)M_ROWS * 1024), got((size_t)M_ROWS * 1024);
            c.reset_input();
            for (int l = 0; l < 3; ++l) { const Layer& L = c.hl[l]; if (L.N == 1024) launch_gemm<128, 64, 2, W4>(L, st); else launch_gemm<64, 64, 3, W4>(L, st); bn(L, M_ROWS / (L.N == 1024 ? 128 : 64)); }
            hipStreamSynchronize(st); hipMemcpy(ref.data(), c.zA, ref.size() * 2, hipMemcpyDeviceToHost);
            c.reset_input();
            chain_N(1, true);
            hipStreamSynchronize(st); hipMemcpy(got.data(), c.zA, got.size() * 2, hipMemcpyDeviceToHost);
            double md = 0; size_t nz = 0;
            for (size_t i = 0; i < ref.size(); ++i) { md = fmax(md, fabs((double)(float)ref[i] - (double)(float)got[i])); nz += (float)got[i] != 0.f; }
            printf("# check N (normalize on load) vs two launches after one block: max |dz| %.5f, %zu non-zero of %zu\n", md, nz, ref.size());
        }
        for (int fin = 0; fin < 3; ++fin) {
            // what the arrival ticket + last-arriver finalize costs a plain conv launch (c1, 64x64 x512 and c3-shaped 128x64 x1024, repeated back to back)
            for (int kind = 0; kind < 3; kind += 2) {
                hipGraphExec_t ge = graph_of(st, [&]() {
                    for (int r = 0; r < c.n_layers; ++r) {
                        LayerN p{c.hl[kind], nullptr, nullptr, fin ? sc1 : nullptr, fin ? sh1 : nullptr, tick, nullptr, fin == 2};
                        if (kind == 0) launch_gemm_n<64, 64, 3, false>(p, st); else launch_gemm_n<128, 64, 2, false>(p, st);
                    }
                });
                char name[160];
                snprintf(name, sizeof name, "  conv %s alone, %s", kind == 0 ? "c1 1024->256 64x64 x512" : "c3 256->1024 128x64 x1024", fin == 0 ? "plain (slab only)" : fin == 1 ? "ticket + last-arriver finalize, write-through slab row, no fence" : "ticket + last-arriver finalize, release fence per work-group");
                time_it(c, name, [&]() { hipGraphLaunch(ge, st); }, 3, c.bar, 0);
                hipGraphExecDestroy(ge);
            }
        }
        for (int keep = 1; keep >= 0; --keep) {
            hipGraphExec_t ge = graph_of(st, [&]() { chain_N(c.n_layers / 3, keep != 0); });
            time_it(c, keep ? "N normalize on load: 3 conv + 1 BatchNorm launch per block; z1, z2 kept (student)" : "N normalize on load: the same, z1 / z2 not kept (teacher)",
                    [&]() { hipGraphLaunch(ge, st); }, 6, c.bar, 0);
            hipGraphExecDestroy(ge);
        }
        hipFree(coef); hipFree(tick);
    }
    {
        hipMemset(c.bar, 0, sizeof(GridBar));
        unsigned ep = 0;
        time_it(c, "P one launch for the chain, 256 work-groups, two barriers per layer", [&]() {
            hipLaunchKernelGGL((persist_k<W4>), dim3(256), dim3(256), kLds, st, c.d_layers, c.n_layers, c.bar, ep);
            ep += 2u * c.n_layers;
        }, 6, c.bar, 1);
    }
    {
        hipGraphExec_t ge = graph_of(st, [&]() { for (const Layer& L : c.hl) { if (L.N == 1024) launch_gemm<128, 64, 2, W4>(L, st); else launch_gemm<64, 64, 3, W4>(L, st); } });
        time_it(c, "  conv launches alone (c1/c2 64x64 x512, c3 128x64 x1024)", [&]() { hipGraphLaunch(ge, st); }, 4, c.bar, 0);
        hipGraphExecDestroy(ge);
        for (int kind = 0; kind < 3; ++kind)
            for (int form = 0; form < 2; ++form) {
                // one layer kind repeated (the methodology of tools/tune_conv.py: the same launch back to back)
                ge = graph_of(st, [&]() {
                    for (int r = 0; r < c.n_layers; ++r) {
                        const Layer& L = c.hl[kind];
                        if (kind == 2) { if (form) launch_gemm<128, 256, 2, W4>(L, st); else launch_gemm<128, 64, 2, W4>(L, st); }
                        else if (form) launch_gemm<128, 64, 2, W4>(L, st); else launch_gemm<64, 64, 3, W4>(L, st);
                    }
                });
                char name[160];
                snprintf(name, sizeof name, "  conv %s alone, %s", kind == 0 ? "c1 1024->256" : kind == 1 ? "c2 3x3 256" : "c3 256->1024",
                         kind == 2 ? (form ? "128x256 x256" : "128x64 x1024") : (form ? "128x64 x256" : "64x64 x512"));
                time_it(c, name, [&]() { hipGraphLaunch(ge, st); }, 3, c.bar, 0);
                hipGraphExecDestroy(ge);
            }
        ge = graph_of(st, [&]() { for (const Layer& L : c.hl) bn(L, M_ROWS / 128); });
        time_it(c, "  BatchNorm launches alone", [&]() { hipGraphLaunch(ge, st); }, 4, c.bar, 0);
        hipGraphExecDestroy(ge);
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 23;
    Ctx c;
    c.n_layers = blocks * 3;
    hipStreamCreate(&c.st);
    // activations: A (1024 ch) -> c1 -> B (256) -> c2 -> Cc (256) -> c3 -> A   (buffers sized for 4-byte elements)
    bf16 *zA, *zB, *zC, *yA, *yB, *yC, *w1, *w2, *w3; float *slab, *gamma, *beta;
    hipMalloc((void**)&zA, (size_t)M_ROWS * 1024 * 4); hipMalloc((void**)&zB, (size_t)M_ROWS * 256 * 4); hipMalloc((void**)&zC, (size_t)M_ROWS * 256 * 4);
    hipMalloc((void**)&yA, (size_t)M_ROWS * 1024 * 4); hipMalloc((void**)&yB, (size_t)M_ROWS * 256 * 4); hipMalloc((void**)&yC, (size_t)M_ROWS * 256 * 4);
    hipMalloc((void**)&w1, (size_t)256 * 1024 * 2); hipMalloc((void**)&w2, (size_t)256 * 2304 * 2); hipMalloc((void**)&w3, (size_t)1024 * 256 * 2);
    hipMalloc((void**)&slab, (size_t)3 * 128 * 2 * 1024 * 4); hipMalloc((void**)&gamma, 1024 * 4); hipMalloc((void**)&beta, 1024 * 4);
    hipMalloc((void**)&c.bar, sizeof(GridBar)); hipMalloc((void**)&c.d_layers, sizeof(Layer) * c.n_layers);
    hipMemset(zA, 0, (size_t)M_ROWS * 1024 * 4); hipMemset(zB, 0, (size_t)M_ROWS * 256 * 4); hipMemset(zC, 0, (size_t)M_ROWS * 256 * 4);
    c.zA = zA;
    c.input.resize((size_t)M_ROWS * 1024); fill_bf16(c.input, 1u, 1.f);
    {
        std::vector<bf16> a((size_t)256 * 1024); fill_bf16(a, 2u, 0.05f); hipMemcpy(w1, a.data(), a.size() * 2, hipMemcpyHostToDevice);
        std::vector<bf16> b((size_t)256 * 2304); fill_bf16(b, 3u, 0.05f); hipMemcpy(w2, b.data(), b.size() * 2, hipMemcpyHostToDevice);
        std::vector<bf16> d((size_t)1024 * 256); fill_bf16(d, 4u, 0.05f); hipMemcpy(w3, d.data(), d.size() * 2, hipMemcpyHostToDevice);
        std::vector<float> g(1024, 1.f), z(1024, 0.25f); hipMemcpy(gamma, g.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(beta, z.data(), 4096, hipMemcpyHostToDevice);
    }
    float* slab1 = slab, *slab2 = slab + (size_t)128 * 2 * 1024, *slab3 = slab2 + (size_t)128 * 2 * 1024;
    const Layer c1{zA, w1, yB, zB, slab1, gamma, beta, 1024, 256, 1, nullptr, 0}, c2{zB, w2, yC, zC, slab2, gamma, beta, 256, 256, 9, nullptr, 0}, c3{zC, w3, yA, zA, slab3, gamma, beta, 256, 1024, 1, nullptr, 0};
    const bool cold = argc > 2 && atoi(argv[2]) != 0;
    for (int b = 0; b < blocks; ++b) {
        Layer a = c1, m = c2, e = c3;
        if (cold) {       // every layer its own weights (copies of the three tensors): the network's case - each weight tensor is read once per pass
            bf16 *x1, *x2, *x3;
            hipMalloc((void**)&x1, (size_t)256 * 1024 * 2); hipMalloc((void**)&x2, (size_t)256 * 2304 * 2); hipMalloc((void**)&x3, (size_t)1024 * 256 * 2);
            hipMemcpy(x1, w1, (size_t)256 * 1024 * 2, hipMemcpyDeviceToDevice); hipMemcpy(x2, w2, (size_t)256 * 2304 * 2, hipMemcpyDeviceToDevice);
            hipMemcpy(x3, w3, (size_t)1024 * 256 * 2, hipMemcpyDeviceToDevice);
            a.w = x1; m.w = x2; e.w = x3;
        }
        c.hl.push_back(a); c.hl.push_back(m); c.hl.push_back(e);
    }
    if (argc > 3 && atoi(argv[3]) != 0) {
        for (size_t l = 0; l + 1 < c.hl.size(); ++l) { c.hl[l].next_w = c.hl[l + 1].w; c.hl[l].next_w_bytes = c.hl[l + 1].N * c.hl[l + 1].Kc * c.hl[l + 1].taps * 2; }
        printf("# PREFETCH: every BatchNorm launch of U pulls the next layer's weights into every XCD's L2\n");
    }
    if (cold) { hipMalloc((void**)&c.evict, (size_t)1 << 30); printf("# COLD WEIGHTS: every layer has its own weight tensors and 1 GiB is written between repetitions\n"); }
    hipMemcpy(c.d_layers, c.hl.data(), sizeof(Layer) * c.n_layers, hipMemcpyHostToDevice);
    hipMemset(c.bar, 0, sizeof(GridBar));
    printf("# conv -> BatchNorm(train) -> ReLU chain, layer3 geometry (M = 8192; c1 1024->256, c2 3x3 256->256, c3 256->1024), %d blocks = %d conv+BN, alone on the chip\n", blocks, c.n_layers);
    run_forms<false>(c);
    for (int wgs : {256, 512}) {
        GridBar* b2; hipMalloc((void**)&b2, sizeof(GridBar)); hipMemset(b2, 0, sizeof(GridBar));
        unsigned ep = 0;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        double best = 1e30;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(a, c.st);
            hipLaunchKernelGGL(bar_only_k, dim3(wgs), dim3(256), 0, c.st, b2, 200, ep);
            hipEventRecord(b, c.st); hipEventSynchronize(b);
            ep += 200u;
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (r >= 1 && ms < best) best = ms;
        }
        unsigned err = 0; hipMemcpy(&err, &b2->err[0], 4, hipMemcpyDeviceToHost);
        printf("  the grid barrier alone, %d work-groups, 200 back to back: %.2f us each%s\n", wgs, best * 1e3 / 200, err ? "  ** GAVE UP **" : "");
        hipFree(b2);
    }
    run_forms<true>(c);       // (values are meaningless here - the next layer reads the 4-byte z as bf16 - only the time is)
    return 0;
}
