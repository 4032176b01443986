// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the UDA pose hot path.
// wave = 64 lanes; MFMA 16x16x32 bf16; LDS 160 KiB/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UDAPOSE_OK 0
#define UDAPOSE_ERR_ARG (-1)
#define UDAPOSE_ERR_LAUNCH (-2)
#define UDAPOSE_ERR_UNSUPPORTED (-3)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
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

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
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
