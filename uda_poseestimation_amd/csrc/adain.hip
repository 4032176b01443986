// AdaIN feature re-normalisation (lib/models/Style_net.py:4-29,167-168) on NHWC bf16 features.
// One block per (image, 64-channel slab): 8 channel groups x 32 pixel lanes.  Pass 1 sweeps content and style once for
// per-(n,c) sum / sum-of-squares (fp32, wave + LDS reduced), pass 2 re-reads content (L2/MALL resident: 128 KiB slab) and
// writes alpha*((c-mu_c)/sd_c*sd_s+mu_s) + (1-alpha)*c.  Unbiased variance + eps like the reference.
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
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
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

template <typename T>
__global__ void adain_k(const T* __restrict__ content, const T* __restrict__ style, T* __restrict__ out, int HWc, int HWs, int C,
                        float eps, float alpha, float* __restrict__ stats_out) {
    __shared__ float red[32][8][4][8];   // [pixel lane][cg][stat][e]  32 KiB
    __shared__ float coef[64][2];
    const int slabs = C / 64;
    const int n = blockIdx.x / slabs, sl = blockIdx.x % slabs;
    const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c0 = sl * 64 + cg * 8;
    float cs[8], cq[8], ss[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cq[e] = ss[e] = sq[e] = 0.f;
    const T* cp = content + (size_t)n * HWc * C + c0;
    const T* sp = style + (size_t)n * HWs * C + c0;
    for (int p = pl; p < HWc; p += 32) {
        float v[8];
        ld8<T>(cp + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = v[e]; cs[e] += f; cq[e] += f * f; }
    }
    for (int p = pl; p < HWs; p += 32) {
        float v[8];
        ld8<T>(sp + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = v[e]; ss[e] += f; sq[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[pl][cg][0][e] = cs[e]; red[pl][cg][1][e] = cq[e]; red[pl][cg][2][e] = ss[e]; red[pl][cg][3][e] = sq[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int g = threadIdx.x >> 3, e = threadIdx.x & 7;
        double a[4] = {0, 0, 0, 0};
        for (int k = 0; k < 32; ++k)
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
    float k0[8], k1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { k0[e] = coef[cg * 8 + e][0]; k1[e] = coef[cg * 8 + e][1]; }
    if (!out) return;                     // statistics only (calc_mean_std)
    T* op = out + (size_t)n * HWc * C + c0;
    for (int p = pl; p < HWc; p += 32) {
        float v[8], o[8];
        ld8<T>(cp + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] * k0[e] + k1[e];
        st8<T>(op + (size_t)p * C, o);
    }
}
}  // namespace

int adain_launch(hipStream_t s, const elem_t* content, const elem_t* style, elem_t* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                 float* stats_out) {
    if (C % 64 || HWc < 2 || HWs < 2) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(adain_k<elem_t>, dim3(N * (C / 64)), dim3(TPB), 0, s, content, style, out, HWc, HWs, C, eps, alpha, stats_out);
    return udapose_check_launch();
}
int adain_launch_f32(hipStream_t s, const float* content, const float* style, float* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                     float* stats_out) {
    if (C % 64 || HWc < 2 || HWs < 2) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(adain_k<float>, dim3(N * (C / 64)), dim3(TPB), 0, s, content, style, out, HWc, HWs, C, eps, alpha, stats_out);
    return udapose_check_launch();
}
