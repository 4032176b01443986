// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the UDA pose hot path.
// wave = 64 lanes; MFMA 16x16x32 bf16; LDS 160 KiB/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UDAPOSE_OK 0
#define UDAPOSE_ERR_ARG (-1)
#define UDAPOSE_ERR_LAUNCH (-2)
#define UDAPOSE_ERR_UNSUPPORTED (-3)
#define UDAPOSE_ERR_NOT_PREPARED (-4)   // a device table this call needs was not built (udapose_net_bind / udapose_conv_prepare)

// Storage / MFMA-operand element type of this build of the library.  The same sources are compiled twice:
//   libudapose_hip.so      elem_t = bf16  (v_mfma_f32_16x16x32_bf16)   - BASELINE.json's benched precision
//   libudapose_hip_f16.so  elem_t = fp16  (v_mfma_f32_16x16x32_f16)    - the reference's autocast dtype (train_human.py:280,414)
// Accumulation, BatchNorm statistics, master weights, gradients of weights and every loss are fp32 in both.
#if defined(UDAPOSE_ELEM_F16)
typedef _Float16 elem_t;
typedef __attribute__((ext_vector_type(8))) _Float16 elem8;
typedef __attribute__((ext_vector_type(4))) _Float16 elem4;
#define UDAPOSE_ELEM_KIND 1
#define UDAPOSE_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#else
typedef __bf16 elem_t;
typedef __attribute__((ext_vector_type(8))) __bf16 elem8;
typedef __attribute__((ext_vector_type(4))) __bf16 elem4;
#define UDAPOSE_ELEM_KIND 0
#define UDAPOSE_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ---- "f16x2" storage: the fp32-grade precision mode (both builds of the library carry it; it does not depend on elem_t).
// A value v is kept as two fp16 numbers (h, l) with v = h + l * 2^-11 to ~2^-22 relative: h = fp16(v), l = fp16((v - h) * 2^11)
// (the remainder is scaled so that it stays in fp16's normal range whenever h does).  A product of two such values is
// h_a h_b + (h_a l_b + l_a h_b) 2^-11 + O(2^-22): THREE v_mfma_f32_16x16x32_f16 per 32-deep K step with two fp32 accumulators,
// against eight v_mfma_f32_16x16x4_f32 at a quarter of the rate each for the exact fp32 form (gfx950 has no xf32).
// Memory layout of a split tensor: every group of 8 consecutive channels is 32 bytes, [8 x h][8 x l] - byte for byte the
// footprint and the addressing of an fp32 NHWC tensor (element stride 4 bytes, 16-byte chunks), so the fp32 loaders of the
// igemm kernel and the 8-channel pointwise kernels serve it unchanged, and one 128-byte LDS row of 32 channels holds the
// MFMA fragments of lane group q as chunk 2q (h) and chunk 2q+1 (l).  Range: |v| <= 65504 (saturating).
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
struct sp32 { unsigned int bits; };      // 4-byte stride type of a split tensor (never read as a scalar: groups of 8 only)
#define UDAPOSE_SP_SCALE 2048.f
#define UDAPOSE_SP_INV (1.f / 2048.f)
// Saturation counter of the f16x2 stores (round 4; VERDICT r3: the mode saturated silently): every store of a value outside fp16's
// range (|v| > 65504, or NaN, which the clamp below would turn into -65504) adds one to a device counter.  Device code is linked per
// source file, so every translation unit that stores split values owns a counter (UDAPOSE_SP_SAT_READER defines its host-side reader)
// and udapose_split_saturations sums them.  A saturating network is the exception: the atomic is behind a wave-uniform-in-practice branch.
static __device__ unsigned int g_sp_sat_count = 0u;
#define UDAPOSE_SP_SAT_READER(fn)                                                                          \
    unsigned long long fn(int reset) {                                                                     \
        unsigned int v = 0u, z = 0u;                                                                       \
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sp_sat_count), sizeof(v)) != hipSuccess) return ~0ull;   \
        if (reset && v) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sp_sat_count), &z, sizeof(z));              \
        return v;                                                                                          \
    }
__device__ __forceinline__ void sp_split8(const float (&v)[8], half8& h, half8& l) {
    bool sat = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sat |= !(fabsf(v[e]) <= 65504.f);
        const float c = fminf(fmaxf(v[e], -65504.f), 65504.f);
        const _Float16 hh = (_Float16)c;
        h[e] = hh;
        l[e] = (_Float16)((c - (float)hh) * UDAPOSE_SP_SCALE);
    }
    if (sat) atomicAdd(&g_sp_sat_count, 1u);
}
__device__ __forceinline__ void sp_join8(const half8& h, const half8& l, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)h[e] + (float)l[e] * UDAPOSE_SP_INV;
}
// element e (any index) of a split tensor
__device__ __forceinline__ float sp_load1(const void* base, size_t e) {
    const _Float16* g = (const _Float16*)((const char*)base + (e >> 3) * 32);
    return (float)g[e & 7] + (float)g[8 + (e & 7)] * UDAPOSE_SP_INV;
}
__device__ __forceinline__ void sp_store1(void* base, size_t e, float v) {
    _Float16* g = (_Float16*)((char*)base + (e >> 3) * 32);
    if (!(fabsf(v) <= 65504.f)) atomicAdd(&g_sp_sat_count, 1u);
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    const _Float16 hh = (_Float16)c;
    g[e & 7] = hh;
    g[8 + (e & 7)] = (_Float16)((c - (float)hh) * UDAPOSE_SP_SCALE);
}

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

static inline int udapose_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? UDAPOSE_OK : UDAPOSE_ERR_LAUNCH;
}

// Exact unsigned division by a runtime constant (mul-hi + shift), host-prepared.
struct FastDiv {
    uint32_t d, magic, shift;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d == 1) { f.magic = 0; f.shift = 0; return f; }
    uint32_t s = 0;
    while ((1u << s) < d) ++s;
    uint64_t m = ((((uint64_t)1 << 32) * (((uint64_t)1 << s) - d)) / d) + 1;
    f.magic = (uint32_t)m;
    f.shift = s;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    if (f.d == 1) return n;
    uint32_t t = __umulhi(n, f.magic);
    return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

// one BN layer of the batched running-statistics update (pw_bn_running_update_multi)
struct BnRunJob { size_t save_off; float* rm; float* rv; long long* nbt; int C, pad; };

// Clears `bytes` (a multiple of 4, 4-byte aligned) at p with a KERNEL.  Every clear on a capturable path goes through this instead of
// hipMemsetAsync: on ROCm 7.2 a hipGraph memset node can be replayed out of order with the kernel node that depends on it (from the
// third replay of a small captured graph on, with the runtime's default DEBUG_CLR_GRAPH_PACKET_CAPTURE=1: the clear lands AFTER the
// scatter that follows it; tools/probe/graph_memset_order.py reproduces it with torch alone).  pointwise.hip.
int pw_zero(hipStream_t s, void* p, size_t bytes);

// one range of the multi-range clear (pw_zero_multi): byte offset from a base pointer, length in 16-byte units
struct ZeroJob { long long off; long long n16; };
// one layer of the split-sum launch (pw_split_sum): `ks` partial tiles of `n` floats, `stride` floats apart, at byte offset part_off of the pass's
// workspace, are added in split order into the tensor at byte offset dst_off of the gradient base (dst_ws: of the workspace); beta 1 accumulates
#define UDAPOSE_SPLIT_SUM_CHUNK 1024u      // floats of a job one work-group of pw_split_sum adds (one 16-byte column per thread)
struct SumJob { long long part_off; long long dst_off; unsigned n; unsigned stride; int ks; int dst_ws; float beta; int pad; };
int pw_split_sum(hipStream_t s, const SumJob* d_jobs, const int* d_blk, int nblk, void* ws, void* grad_base, void* ws2 = nullptr, void* grad_base2 = nullptr);

// XCD-aware work-group remap (MI355X: 8 XCDs, each with a private 4 MiB L2; work-groups are dealt round-robin, so b and
// b+8 share an L2).  Returns a bijective permutation of the linear block id that gives every XCD one CONTIGUOUS range of
// work ids, so that blocks which re-read the same operand panels hit the same L2 instead of each XCD streaming the whole
// working set from the Infinity Cache.  Placement is a speed matter only; results never depend on it.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t total) {
    const uint32_t q = total >> 3, r = total & 7u, xcd = bid & 7u, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
