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

// one range of the multi-range clear (pw_zero_multi): byte offset from a base pointer, length in 16-byte units
struct ZeroJob { long long off; long long n16; };

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
