// AdaIN feature re-normalisation (lib/models/Style_net.py:4-29,167-168) on NHWC features (element type or fp32):
// out = alpha*((c-mu_c)/sd_c*sd_s+mu_s) + (1-alpha)*c with per-(n,c) statistics over H*W.  HBM-bound: the algorithmic minimum
// is content + style in, result out (201 MB for [32,512,32,32] fp32, 100 MB in bf16).
#include "common.h"

namespace {
constexpr int TPB = 256;

// 8 consecutive channel values <-> fp32 registers, bf16 or fp32 storage (fp32: the reference's own precision for the style
// path, train_human.py:347-356 runs it outside autocast)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8<elem_t>(const elem_t* p, float (&o)[8]) {
    const elem8 v = *(const elem8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void ld8<sp32>(const sp32* p, float (&o)[8]) {      // f16x2 split storage (common.h)
    const char* c = (const char*)p;
    sp_join8(*(const half8*)c, *(const half8*)(c + 16), o);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<sp32>(sp32* p, const float (&v)[8]) {
    half8 h, l;
    sp_split8(v, h, l);
    char* c = (char*)p;
    *(half8*)c = h;
    *(half8*)(c + 16) = l;
}
template <> __device__ __forceinline__ void st8<elem_t>(elem_t* p, const float (&v)[8]) {
    elem8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (elem_t)v[e];
    *(elem8*)p = o;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}

// One work-group of 1024 threads (16 waves: the CU's memory pipeline needs that many loads in flight - the 256-thread form
// of round 1 moved 2.1 TB/s) per (image, 64-channel slab): 8 channel groups x 128 pixel lanes, every lane moves 16 (bf16) or 32
// (fp32) contiguous bytes per pixel, a wave covers whole 128- / 256-byte pixel rows of the slab.  Pass 1 sweeps content and
// style once for the per-(n,c) sums (lane partials -> three xor steps inside the wave -> one LDS row per wave); the first
// CACHE pixels of each lane's content stay in registers, so for feature maps up to 128 * CACHE = 1024 pixels (relu4_1 of a
// 256x256 image: the training shape) pass 2 re-reads nothing: the kernel moves the algorithmic minimum (content + style in,
// result out).  Larger maps re-read the uncached remainder.  Unbiased variance + eps like the reference.
constexpr int ATPB = 1024, APL = ATPB / 8, CACHE = 8;

// the cached pixels stay in their STORAGE form (4 registers per 8 bf16 values, 8 per 8 floats)
template <typename T> struct Raw8;
template <> struct Raw8<elem_t> { elem8 v; };
template <> struct Raw8<float> { f32x4 a, b; };
template <> struct Raw8<sp32> { half8 h, l; };
__device__ __forceinline__ void ldraw(const sp32* p, Raw8<sp32>& r) { const char* c = (const char*)p; r.h = *(const half8*)c; r.l = *(const half8*)(c + 16); }
__device__ __forceinline__ void unraw(const Raw8<sp32>& r, float (&o)[8]) { sp_join8(r.h, r.l, o); }
__device__ __forceinline__ void ldraw(const elem_t* p, Raw8<elem_t>& r) { r.v = *(const elem8*)p; }
__device__ __forceinline__ void ldraw(const float* p, Raw8<float>& r) { r.a = *(const f32x4*)p; r.b = *(const f32x4*)(p + 4); }
__device__ __forceinline__ void unraw(const Raw8<elem_t>& r, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)r.v[e];
}
__device__ __forceinline__ void unraw(const Raw8<float>& r, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = r.a[e]; o[4 + e] = r.b[e]; }
}

template <typename T>
__global__ __launch_bounds__(ATPB) void adain_k(const T* __restrict__ content, const T* __restrict__ style, T* __restrict__ out, int HWc, int HWs,
                                                int C, float eps, float alpha, const float* __restrict__ alpha_dev, float* __restrict__ stats_out) {
    if (alpha_dev) alpha = *alpha_dev;      // (a captured launch reads the step's blend factor from device memory)
    __shared__ float red[ATPB / 64][8][4][8];   // [wave][cg][stat][e]  16 KiB
    __shared__ float coef[64][2];
    const int slabs = C / 64;
    const int n = blockIdx.x / slabs, sl = blockIdx.x % slabs;
    const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c0 = sl * 64 + cg * 8;
    const T* cp = content + (size_t)n * HWc * C + c0;
    const T* sp = style + (size_t)n * HWs * C + c0;
    const int wave = threadIdx.x >> 6;
    Raw8<T> keep[CACHE];
    {   // content: sums + register cache
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int k = 0; k < CACHE; ++k) {
            const int p = pl + k * APL;
            if (p < HWc) {
                float v[8];
                ldraw(cp + (size_t)p * C, keep[k]);
                unraw(keep[k], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
            }
        }
        for (int p = pl + CACHE * APL; p < HWc; p += APL) {
            float v[8];
            ld8<T>(cp + (size_t)p * C, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
        }
        // lanes of a wave with equal cg (lane & 7) hold partials of the same channels: xor over lane bits 3..5
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        if ((threadIdx.x & 63) < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wave][cg][0][e] = s1[e]; red[wave][cg][1][e] = s2[e]; }
        }
    }
    {   // style: sums only
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        for (int p = pl; p < HWs; p += APL) {
            float v[8];
            ld8<T>(sp + (size_t)p * C, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        if ((threadIdx.x & 63) < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wave][cg][2][e] = s1[e]; red[wave][cg][3][e] = s2[e]; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int g = threadIdx.x >> 3, e = threadIdx.x & 7;
        double a[4] = {0, 0, 0, 0};
        for (int k = 0; k < ATPB / 64; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += (double)red[k][g][q][e];
        const double mc = a[0] / HWc, ms = a[2] / HWs;
        const double vc = (a[1] - a[0] * mc) / (HWc - 1) + eps, vs = (a[3] - a[2] * ms) / (HWs - 1) + eps;
        const float sdc = (float)sqrt(vc > 0 ? vc : 0), sds = (float)sqrt(vs > 0 ? vs : 0);
        // out = alpha*((x-mc)/sdc*sds+ms) + (1-alpha)*x = x*(alpha*sds/sdc + 1-alpha) + alpha*(ms - mc*sds/sdc)
        coef[threadIdx.x][0] = alpha * (sds / sdc) + (1.f - alpha);
        coef[threadIdx.x][1] = alpha * ((float)ms - (float)mc * (sds / sdc));
        if (stats_out) {
            float* so = stats_out + ((size_t)n * C + sl * 64 + threadIdx.x) * 4;
            so[0] = (float)mc; so[1] = sdc; so[2] = (float)ms; so[3] = sds;
        }
    }
    __syncthreads();
    if (!out) return;                     // statistics only (calc_mean_std)
    float k0[8], k1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { k0[e] = coef[cg * 8 + e][0]; k1[e] = coef[cg * 8 + e][1]; }
    T* op = out + (size_t)n * HWc * C + c0;
#pragma unroll
    for (int k = 0; k < CACHE; ++k) {
        const int p = pl + k * APL;
        if (p < HWc) {
            float v[8], o[8];
            unraw(keep[k], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[e] * k0[e] + k1[e];
            st8<T>(op + (size_t)p * C, o);
        }
    }
    for (int p = pl + CACHE * APL; p < HWc; p += APL) {
        float v[8], o[8];
        ld8<T>(cp + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] * k0[e] + k1[e];
        st8<T>(op + (size_t)p * C, o);
    }
}
}  // namespace

int adain_launch(hipStream_t s, const elem_t* content, const elem_t* style, elem_t* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                 const float* alpha_dev, float* stats_out) {
    if (C % 64 || HWc < 2 || HWs < 2) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(adain_k<elem_t>, dim3(N * (C / 64)), dim3(ATPB), 0, s, content, style, out, HWc, HWs, C, eps, alpha, alpha_dev, stats_out);
    return udapose_check_launch();
}
int adain_launch_f32(hipStream_t s, const float* content, const float* style, float* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                     const float* alpha_dev, float* stats_out) {
    if (C % 64 || HWc < 2 || HWs < 2) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(adain_k<float>, dim3(N * (C / 64)), dim3(ATPB), 0, s, content, style, out, HWc, HWs, C, eps, alpha, alpha_dev, stats_out);
    return udapose_check_launch();
}
int adain_launch_split(hipStream_t s, const void* content, const void* style, void* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                       const float* alpha_dev, float* stats_out) {
    if (C % 64 || HWc < 2 || HWs < 2) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(adain_k<sp32>, dim3(N * (C / 64)), dim3(ATPB), 0, s, (const sp32*)content, (const sp32*)style, (sp32*)out, HWc, HWs, C, eps, alpha,
                       alpha_dev, stats_out);
    return udapose_check_launch();
}

UDAPOSE_SP_SAT_READER(sp_sat_read_adain)
