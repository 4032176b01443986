// AdaIN feature re-normalisation (lib/models/Style_net.py:4-29,167-168) on NHWC bf16 features.
// One block per (image, 64-channel slab): 8 channel groups x 32 pixel lanes.  Pass 1 sweeps content and style once for
// per-(n,c) sum / sum-of-squares (fp32, wave + LDS reduced), pass 2 re-reads content (L2/MALL resident: 128 KiB slab) and
// writes alpha*((c-mu_c)/sd_c*sd_s+mu_s) + (1-alpha)*c.  Unbiased variance + eps like the reference.
#include "common.h"

namespace {
constexpr int TPB = 256;

__global__ void adain_k(const bf16_t* __restrict__ content, const bf16_t* __restrict__ style, bf16_t* __restrict__ out, int HWc, int HWs, int C,
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
    const bf16_t* cp = content + (size_t)n * HWc * C + c0;
    const bf16_t* sp = style + (size_t)n * HWs * C + c0;
    for (int p = pl; p < HWc; p += 32) {
        const bf16x8 v = *(const bf16x8*)(cp + (size_t)p * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; cs[e] += f; cq[e] += f * f; }
    }
    for (int p = pl; p < HWs; p += 32) {
        const bf16x8 v = *(const bf16x8*)(sp + (size_t)p * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; ss[e] += f; sq[e] += f * f; }
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
    bf16_t* op = out + (size_t)n * HWc * C + c0;
    for (int p = pl; p < HWc; p += 32) {
        const bf16x8 v = *(const bf16x8*)(cp + (size_t)p * C);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)v[e] * k0[e] + k1[e]);
        *(bf16x8*)(op + (size_t)p * C) = o;
    }
}
}  // namespace

int adain_launch(hipStream_t s, const bf16_t* content, const bf16_t* style, bf16_t* out, int N, int HWc, int HWs, int C, float eps, float alpha,
                 float* stats_out) {
    if (C % 64 || HWc < 2 || HWs < 2) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(adain_k, dim3(N * (C / 64)), dim3(TPB), 0, s, content, style, out, HWc, HWs, C, eps, alpha, stats_out);
    return udapose_check_launch();
}
