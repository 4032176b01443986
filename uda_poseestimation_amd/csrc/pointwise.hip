// Memory-bound kernels of the pose network for gfx950: layout conversion, weight packing, training-mode BatchNorm
// (finalize / apply / backward), 3x3 s2 and 2x2 ceil max-pooling.  All activation tensors are NHWC bf16 and every
// thread moves 16 bytes (8 channels) per access.  These kernels are HBM-roofline bound (8 TB/s peak).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TPB = 256;
inline int nblk(size_t n, int per = TPB) { return (int)((n + per - 1) / per); }

// ------------------------------------------------------------------ layout conversion
// 8 consecutive activation values <-> fp32 registers, for bf16 or fp32 storage
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
template <> __device__ __forceinline__ void ld8<sp32>(const sp32* p, float (&o)[8]) {      // f16x2 split storage (common.h): 32 bytes = [8 h][8 l]
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
template <typename T> __device__ __forceinline__ float ld1(const T* p, size_t i) { return (float)p[i]; }
template <> __device__ __forceinline__ float ld1<sp32>(const sp32* p, size_t i) { return sp_load1(p, i); }
template <typename T> __device__ __forceinline__ void st1(T* p, size_t i, float v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void st1<sp32>(sp32* p, size_t i, float v) { sp_store1(p, i, v); }
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

// src NCHW fp32 [N,C,HW] -> dst NHWC [N,HW,Cp] (channels >= C zero-filled); one thread per (pixel, 8-channel group)
template <typename T>
__global__ void nchw_f32_to_nhwc_k(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int HW, int Cp) {
    const int G = Cp >> 3;
    const size_t total = (size_t)N * HW * G;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const size_t pix = i % ((size_t)N * HW);      // pixel fastest: coalesced plane reads
        const int g = (int)(i / ((size_t)N * HW));
        const size_t n = pix / HW, hw = pix % HW;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = g * 8 + e;
            o[e] = c < C ? src[(n * C + c) * HW + hw] : 0.f;
        }
        st8<T>(dst + pix * Cp + g * 8, o);
    }
}

// src NHWC (fp32 or bf16) [N,HW,Cs] -> dst NCHW fp32 [N,C,HW], optional per-channel clamp (style-net output)
template <typename T>
__global__ void nhwc_to_nchw_f32_k(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int HW, int Cs,
                                   const float* __restrict__ lo, const float* __restrict__ hi) {
    const size_t total = (size_t)N * HW;
    for (size_t pix = (size_t)blockIdx.x * TPB + threadIdx.x; pix < total; pix += (size_t)gridDim.x * TPB) {
        const size_t n = pix / HW, hw = pix % HW;
        for (int c = 0; c < C; ++c) {
            float v = ld1<T>(src, pix * Cs + c);
            if (lo) v = fmaxf(fminf(v, hi[c]), lo[c]);
            dst[(n * C + c) * HW + hw] = v;
        }
    }
}

// ------------------------------------------------------------------ weight packing (fp32 master -> bf16 GEMM layouts)
__global__ void cast_f32_bf16_k(const float* __restrict__ src, elem_t* __restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (size_t)gridDim.x * TPB) {
        const f32x4 a = *(const f32x4*)(src + i * 8), b = *(const f32x4*)(src + i * 8 + 4);
        elem8 o = {(elem_t)a[0], (elem_t)a[1], (elem_t)a[2], (elem_t)a[3], (elem_t)b[0], (elem_t)b[1], (elem_t)b[2], (elem_t)b[3]};
        *(elem8*)(dst + i * 8) = o;
    }
}
// src [A][T][B] fp32 -> dst [B][T][A] (bf16 or fp32) (32x32 LDS-tiled transpose per tap); grid = (B/32, A/32, T)
template <typename D>
__global__ void transpose_cast_k(const float* __restrict__ src, D* __restrict__ dst, int A, int T, int B) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, b0 = blockIdx.x * 32, a0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, b = b0 + tx;
        tile[r][tx] = (a < A && b < B) ? src[((size_t)a * T + t) * B + b] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int b = b0 + r, a = a0 + tx;
        if (a < A && b < B) st1<D>(dst, ((size_t)b * T + t) * A + a, tile[tx][r]);
    }
}
// One launch packs every weight of a network: a job is either a contiguous cast (T == 0: n elements) or a per-tap
// transpose [A][T][B] fp32 -> [B][T][A] bf16; block b works on job blk_job[b], sub-block blk_sub[b].
// kind: bit 0 = transpose, bit 1 = f16x2 split output (the fp32-grade plans) instead of the element type.
struct PackJob { const float* src; elem_t* dst; int A, T, B, kind; long long n; };
__global__ void pack_multi_k(const PackJob* __restrict__ jobs, const int* __restrict__ blk_job, const int* __restrict__ blk_sub) {
    __shared__ float tile[32][33];
    const PackJob j = jobs[blk_job[blockIdx.x]];
    const int sub = blk_sub[blockIdx.x];
    if ((j.kind & 1) == 0) {
        const long long base = (long long)sub * 8192;           // 8192 elements per block
        for (long long i = base + threadIdx.x * 8; i < base + 8192 && i < j.n; i += TPB * 8) {
            const f32x4 a = *(const f32x4*)(j.src + i), b = *(const f32x4*)(j.src + i + 4);
            if (j.kind & 2) {
                const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                st8<sp32>((sp32*)j.dst + i, v);
            } else {
                elem8 o = {(elem_t)a[0], (elem_t)a[1], (elem_t)a[2], (elem_t)a[3], (elem_t)b[0], (elem_t)b[1], (elem_t)b[2], (elem_t)b[3]};
                *(elem8*)(j.dst + i) = o;
            }
        }
        return;
    }
    const int tb = (j.B + 31) / 32, ta = (j.A + 31) / 32;
    const int t = sub / (ta * tb), rem = sub % (ta * tb);
    const int a0 = (rem / tb) * 32, b0 = (rem % tb) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, b = b0 + tx;
        tile[r][tx] = (a < j.A && b < j.B) ? j.src[((size_t)a * j.T + t) * j.B + b] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int b = b0 + r, a = a0 + tx;
        if (a < j.A && b < j.B) {
            if (j.kind & 2) sp_store1(j.dst, ((size_t)b * j.T + t) * j.A + a, tile[tx][r]);
            else j.dst[((size_t)b * j.T + t) * j.A + a] = (elem_t)tile[tx][r];
        }
    }
}

// generic strided gather with zero padding: dst[a][kh][kwp][bp] bf16 <- src[a*sa + kh*skh + kw*skw + b*sb] (kw<KW, b<B)
template <typename D>
__global__ void pack_strided_k(const float* __restrict__ src, D* __restrict__ dst, int A, int KH, int KWp, int KW, int Bp, int B,
                               long sa, long skh, long skw, long sb) {
    const size_t total = (size_t)A * KH * KWp * Bp;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int b = (int)(i % Bp);
        size_t r = i / Bp;
        const int kw = (int)(r % KWp); r /= KWp;
        const int kh = (int)(r % KH);
        const int a = (int)(r / KH);
        float v = 0.f;
        if (kw < KW && b < B) v = src[a * sa + kh * skh + kw * skw + b * sb];
        st1<D>(dst, i, v);
    }
}
// inverse of pack_strided for gradients: dst[a*sa + kh*skh + kw*skw + b*sb] (beta*dst +) = src[a][kh][kwp][bp] (fp32)
__global__ void unpack_strided_k(const float* __restrict__ src, float* __restrict__ dst, int A, int KH, int KWp, int KW, int Bp, int B,
                                 long sa, long skh, long skw, long sb, float beta) {
    const size_t total = (size_t)A * KH * KW * B;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int b = (int)(i % B);
        size_t r = i / B;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH);
        const int a = (int)(r / KH);
        const float v = src[(((size_t)a * KH + kh) * KWp + kw) * Bp + b];
        float* d = dst + (a * sa + kh * skh + kw * skw + b * sb);
        *d = (beta != 0.f ? beta * *d : 0.f) + v;
    }
}

// ------------------------------------------------------------------ BatchNorm (training mode)
// Column sums of a partial-sum slab [rows][2][C]: block = 32 channels x 32 row lanes (1024 threads), 128-byte coalesced
// row segments, fp64 accumulation, LDS tree over the row lanes.  Result for channel c0+cl in (s1, s2) of lanes rl == 0.
#ifndef UDAPOSE_FIN_C
#define UDAPOSE_FIN_C 8
#endif
constexpr int FIN_C = UDAPOSE_FIN_C;    // channels per block.  8 (32-byte row segments, C/8 blocks of 256 threads): the finalize kernels sit in
                                        // every layer's dependency chain and are pure latency - 4x the blocks of the 128-byte-segment form (32)
                                        // measured -0.1 ms per step (profiles/r2_ab_runs.txt); 4 / 16 channels and more row lanes measured slower
#ifndef UDAPOSE_FIN_RL
#define UDAPOSE_FIN_RL 32
#endif
constexpr int FIN_RL = UDAPOSE_FIN_RL;  // row lanes per block
constexpr int FIN_T = FIN_C * FIN_RL;
__device__ __forceinline__ void slab_colsum(const float* __restrict__ slab, int rows, int C, int c, bool cvalid, double& s1, double& s2,
                                            double (*red)[FIN_C][2]) {
    const int cl = threadIdx.x % FIN_C, rl = threadIdx.x / FIN_C;
    double a = 0.0, b = 0.0;
    if (cvalid) {
        // 8, then 4 independent row loads in flight per thread (the loop is pure latency: 1024 slab rows = 32 per lane); the
        // additions keep the order of the 4-row form, so the sums are bit-identical to it
        int r = rl;
        for (; r + 7 * FIN_RL < rows; r += 8 * FIN_RL) {
            const float* q = slab + (size_t)r * 2 * C + c;
            const size_t st = (size_t)FIN_RL * 2 * C;
            const float a0 = q[0], b0 = q[C], a1 = q[st], b1 = q[st + C];
            const float a2 = q[2 * st], b2 = q[2 * st + C], a3 = q[3 * st], b3 = q[3 * st + C];
            const float a4 = q[4 * st], b4 = q[4 * st + C], a5 = q[5 * st], b5 = q[5 * st + C];
            const float a6 = q[6 * st], b6 = q[6 * st + C], a7 = q[7 * st], b7 = q[7 * st + C];
            a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
            a += ((double)a4 + (double)a5) + ((double)a6 + (double)a7);
            b += ((double)b4 + (double)b5) + ((double)b6 + (double)b7);
        }
        for (; r + 3 * FIN_RL < rows; r += 4 * FIN_RL) {
            const float* q = slab + (size_t)r * 2 * C + c;
            const size_t st = (size_t)FIN_RL * 2 * C;
            const float a0 = q[0], b0 = q[C], a1 = q[st], b1 = q[st + C];
            const float a2 = q[2 * st], b2 = q[2 * st + C], a3 = q[3 * st], b3 = q[3 * st + C];
            a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; r < rows; r += FIN_RL) {
            a += (double)slab[(size_t)r * 2 * C + c];
            b += (double)slab[(size_t)r * 2 * C + C + c];
        }
    }
    red[rl][cl][0] = a; red[rl][cl][1] = b;
    __syncthreads();
    for (int st = FIN_RL / 2; st > 0; st >>= 1) {
        if (rl < st) { red[rl][cl][0] += red[rl + st][cl][0]; red[rl][cl][1] += red[rl + st][cl][1]; }
        __syncthreads();
    }
    s1 = red[0][cl][0]; s2 = red[0][cl][1];
}
// Finalize: reduce the conv epilogue's partial sums in fp64, produce scale/shift and saved mean/invstd, update running
// stats (momentum, unbiased variance) exactly like torch.nn.BatchNorm2d in train().
__global__ __launch_bounds__(FIN_T) void bn_finalize_k(const float* __restrict__ slab, int rows, int C, double count, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                              long long* __restrict__ nbt, float momentum, float eps, float* __restrict__ scale,
                              float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                              const float* __restrict__ pre_bias = nullptr) {
    __shared__ double red[FIN_RL][FIN_C][2];
    const int c = blockIdx.x * FIN_C + (threadIdx.x % FIN_C);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    double s1, s2;
    slab_colsum(slab, rows, C, c, c < C, s1, s2, red);
    if ((threadIdx.x / FIN_C) != 0 || c >= C) return;
    double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    // pre_bias: the producing convolution adds a per-channel bias AFTER its statistics epilogue (the sums are those of the raw
    // accumulators): the mean of what it stored is shifted by it, the variance is not (Upsampling(bias=True), pose_resnet.py:15,41)
    if (pre_bias) mean += (double)pre_bias[c];
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    save_invstd[C + c] = (float)unb;          // third saved vector: unbiased variance (deferred running-stat update)
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
// deferred running-statistics update of one BN layer from its saved batch statistics [3][C] (mean, invstd, unbiased var):
// used when a forward ran concurrently with another forward of the same module and must not race on the buffers
__global__ void bn_running_update_k(const float* __restrict__ save, int C, float* __restrict__ rm, float* __restrict__ rv,
                                    long long* __restrict__ nbt, float momentum) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    rm[c] = (1.f - momentum) * rm[c] + momentum * save[c];
    rv[c] = (1.f - momentum) * rv[c] + momentum * save[2 * C + c];
}
// y += x over n floats (n % 4 == 0): sum of the two per-pass flat gradient buffers
__global__ void axpy_k(float* __restrict__ y, const float* __restrict__ x, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (size_t)gridDim.x * TPB) {
        f32x4 a = *(f32x4*)(y + i * 4);
        const f32x4 b = *(const f32x4*)(x + i * 4);
        a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        *(f32x4*)(y + i * 4) = a;
    }
}
__global__ void axpy_tail_k(float* __restrict__ y, const float* __restrict__ x, int n) {
    if ((int)threadIdx.x < n) y[threadIdx.x] += x[threadIdx.x];
}
// eval mode: scale/shift from running statistics
__global__ void bn_eval_coeff_k(int C, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                const float* __restrict__ rv, float eps, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// z = [relu]( y*scale[c] + shift[c] [+ res] ), NHWC (bf16 or fp32 storage), 8 channels per thread
// (TY: storage of the conv output y; T: storage of the residual and of z - the f16x2 mode keeps y in fp32 and z split)
template <typename T, typename TY = T>
__global__ void bn_apply_k(const TY* __restrict__ y, const T* __restrict__ res, T* __restrict__ z, size_t n8, int C,
                           const float* __restrict__ scale, const float* __restrict__ shift, int relu, unsigned char* __restrict__ mask = nullptr,
                           int xcd = 0) {
    const int G = C >> 3;
    // the launcher keeps gridDim.x * TPB a multiple of G (G is a power of two <= TPB, or the grid is one block per G-aligned
    // stride), so a thread's channel group never changes: its coefficients are loaded once
    const int c0 = (int)(((size_t)blockIdx.x * TPB + threadIdx.x) % G) * 8;
    const f32x4 sa = *(const f32x4*)(scale + c0), sb = *(const f32x4*)(scale + c0 + 4);
    const f32x4 ha = *(const f32x4*)(shift + c0), hb = *(const f32x4*)(shift + c0 + 4);
    // xcd (launcher: TPB % G == 0, gridDim.x % 8 == 0): work-group b runs on XCD b mod 8 (round-robin dealing), and XCD k takes the
    // k-th EIGHTH of the pixel rows - the rows the implicit GEMM's work-groups on XCD k wrote just before and will read next (each
    // XCD owns a contiguous range of m-tiles there): y comes from, and z stays in, that XCD's L2 (tools/probe/l2_handoff.hip)
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x, iend = n8, istep = (size_t)gridDim.x * TPB;
    if (xcd) {
        const size_t rows = n8 / G, per = (rows + 7) / 8;
        const unsigned x = blockIdx.x & 7u, local = blockIdx.x >> 3, nb = gridDim.x >> 3;
        const unsigned RB = TPB / G;
        size_t rend = (size_t)(x + 1) * per;
        if (rend > rows) rend = rows;
        i = ((size_t)x * per + (size_t)local * RB + threadIdx.x / G) * G + threadIdx.x % G;
        iend = rend * G;
        istep = (size_t)nb * RB * G;
    }
    for (; i < iend; i += istep) {
        float v[8];
        ld8<TY>(y + i * 8, v);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] * (e < 4 ? sa[e] : sb[e - 4]) + (e < 4 ? ha[e] : hb[e - 4]);
        if (res) {
            float r[8];
            ld8<T>(res + i * 8, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += r[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (relu && o[e] < 0.f) ? 0.f : o[e];
        st8<T>(z + i * 8, o);
        if constexpr (sizeof(T) == 2) {
        if (mask) {     // bit e: the STORED value of channel e is > 0 (what a reader of z would see)
            unsigned mb = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) mb |= ((float)(T)o[e] > 0.f ? 1u : 0u) << e;
            mask[i] = (unsigned char)mb;
        }
        }
    }
}

// 8 consecutive gradient values as fp32 (the gradient entering a BN backward may be kept in fp32 near the loss, where
// g - mean(g) - xhat*mean(g*xhat) cancels most of g and bf16 rounding of g would dominate the result)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<elem_t>(const elem_t* p, float (&o)[8]) {
    const elem8 v = *(const elem8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}

// The gradient entering the stem's BatchNorm, gathered on the fly from the gradient of the 3x3 stride-2 max-pool's OUTPUT and the
// saved winning taps (what maxpool3x3s2_bwd_k would have written: the <= 4 windows that contain the pixel, fp32 sum, rounded to
// the storage type exactly as that kernel stores it) - the full-resolution gradient tensor (67 MB per pass at 256x256, b=32) is
// never written or re-read.  pix = flat NHW index of the pool INPUT.
struct PoolSrc { const elem_t* dy; const unsigned char* idx; int H, W; };
__device__ __forceinline__ void pool_grad8(const PoolSrc& ps, size_t pix, int C, int c0, float (&d)[8]) {
    const int w = (int)(pix % ps.W);
    const size_t r = pix / ps.W;
    const int h = (int)(r % ps.H);
    const size_t n = r / ps.H;
    const int Ho = (ps.H - 1) / 2 + 1, Wo = (ps.W - 1) / 2 + 1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ho = h >> 1; ho <= ((h + 1) >> 1); ++ho)
        for (int wo = w >> 1; wo <= ((w + 1) >> 1); ++wo) {
            if (ho >= Ho || wo >= Wo) continue;
            const int kh = h - (ho * 2 - 1), kw = w - (wo * 2 - 1);
            if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;
            const size_t o = ((n * Ho + ho) * Wo + wo) * C + c0;
            const unsigned long long pk = *(const unsigned long long*)(ps.idx + o);
            const elem8 dd = *(const elem8*)(ps.dy + o);
            const unsigned tap = (unsigned)(kh * 3 + kw);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (((pk >> (8 * e)) & 0xFF) == tap) acc[e] += (float)dd[e];
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = (float)(elem_t)acc[e];
}

// Backward reduce: per-channel sum(g) and sum(g*xhat), g = dz * (z>0) when relu.  Partial slab [blocks][2][C].
// Requires C/8 to be a power of two <= 256 (thread's channel group is loop-invariant).
template <typename DZ, bool POOL = false>
__global__ void bn_bwd_reduce_k(const DZ* __restrict__ dz, const elem_t* __restrict__ z, const elem_t* __restrict__ y,
                                size_t npix, int C, const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                float* __restrict__ slab, int pix_per_block, const float* __restrict__ gamma, const float* __restrict__ beta,
                                PoolSrc ps = PoolSrc{nullptr, nullptr, 0, 0}) {
    __shared__ float red[TPB][17];
    const int G = C >> 3;
    const int g = threadIdx.x % G, prow = threadIdx.x / G, pstep = TPB / G;
    const int c0 = g * 8;
    float mu[8], is[8], s1[8], s2[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; s1[e] = 0.f; s2[e] = 0.f; sc[e] = 0.f; sh[e] = 0.f; }
    if (relu == 2) {   // ReLU mask recomputed from y with the forward's own scale / shift expressions (z is not read)
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = gamma[c0 + e] * is[e]; sh[e] = beta[c0 + e] - mu[e] * sc[e]; }
    }
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    size_t p1 = p0 + pix_per_block;
    if (p1 > npix) p1 = npix;
    for (size_t p = p0 + prow; p < p1; p += pstep) {
        const size_t off = p * C + c0;
        float d[8];
        if constexpr (POOL) pool_grad8(ps, p, C, c0, d);
        else load8<DZ>(dz + off, d);
        const elem8 yy = *(const elem8*)(y + off);
        elem8 zz = {};
        if (relu == 1) zz = *(const elem8*)(z + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float gv = d[e];
            if (relu == 1 && !((float)zz[e] > 0.f)) gv = 0.f;
            if (relu == 2 && !((float)yy[e] * sc[e] + sh[e] > 0.f)) gv = 0.f;
            s1[e] += gv;
            s2[e] += gv * (((float)yy[e] - mu[e]) * is[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < G) {
        float a[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
        for (int k = 0; k < pstep; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] += red[threadIdx.x + k * G][e];
        float* sp = slab + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sp[c0 + e] = a[e]; sp[C + c0 + e] = a[8 + e]; }
    }
}
// Finalize backward: dgamma, dbeta (beta_acc*old + new) and the apply coefficients ca = gamma*invstd, cb = sum(g)/M,
// cc = sum(g*xhat)/M.
__global__ __launch_bounds__(FIN_T) void bn_bwd_finalize_k(const float* __restrict__ slab, int rows, int C, double count, const float* __restrict__ gamma,
                                  const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  float beta_acc, float* __restrict__ coef) {
    __shared__ double red[FIN_RL][FIN_C][2];
    const int c = blockIdx.x * FIN_C + (threadIdx.x % FIN_C);
    double s1, s2;
    slab_colsum(slab, rows, C, c, c < C, s1, s2, red);
    if ((threadIdx.x / FIN_C) != 0 || c >= C) return;
    if (dgamma) {
        dgamma[c] = (beta_acc != 0.f ? beta_acc * dgamma[c] : 0.f) + (float)s2;
        dbeta[c] = (beta_acc != 0.f ? beta_acc * dbeta[c] : 0.f) + (float)s1;
    }
    coef[c] = gamma[c] * invstd[c];
    coef[C + c] = (float)(s1 / count);
    coef[2 * C + c] = (float)(s2 / count);
}
// dy = ca*(g - cb - xhat*cc); optionally also write g (masked dz) for the skip branch
template <typename DZ, bool POOL = false>
__global__ void bn_bwd_apply_k(const DZ* __restrict__ dz, const elem_t* __restrict__ z, const elem_t* __restrict__ y,
                               elem_t* __restrict__ dy, elem_t* __restrict__ gout, size_t n8, int C, const float* __restrict__ mean,
                               const float* __restrict__ invstd, const float* __restrict__ coef, int relu, const float* __restrict__ gamma,
                               const float* __restrict__ beta, PoolSrc ps = PoolSrc{nullptr, nullptr, 0, 0}) {
    const int G = C >> 3;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (size_t)gridDim.x * TPB) {
        const int c0 = (int)(i % G) * 8;
        float d[8];
        if constexpr (POOL) pool_grad8(ps, i / G, C, c0, d);
        else load8<DZ>(dz + i * 8, d);
        const elem8 yy = *(const elem8*)(y + i * 8);
        elem8 zz = {};
        if (relu == 1) zz = *(const elem8*)(z + i * 8);
        // per-channel coefficients as 16-byte loads (8 consecutive channels)
        float mu[8], is[8], ca[8], cb[8], cc[8], sc[8], sh[8];
        load8<float>(mean + c0, mu);
        load8<float>(invstd + c0, is);
        load8<float>(coef + c0, ca);
        load8<float>(coef + C + c0, cb);
        load8<float>(coef + 2 * C + c0, cc);
        if (relu == 2) {
            float ga[8], be[8];
            load8<float>(gamma + c0, ga);
            load8<float>(beta + c0, be);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = ga[e] * is[e]; sh[e] = be[e] - mu[e] * sc[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
        }
        elem8 o, go;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float gv = d[e];
            if (relu == 1 && !((float)zz[e] > 0.f)) gv = 0.f;
            if (relu == 2 && !((float)yy[e] * sc[e] + sh[e] > 0.f)) gv = 0.f;
            const float xh = ((float)yy[e] - mu[e]) * is[e];
            o[e] = (elem_t)(ca[e] * (gv - cb[e] - xh * cc[e]));
            go[e] = (elem_t)gv;
        }
        *(elem8*)(dy + i * 8) = o;
        if (gout) *(elem8*)(gout + i * 8) = go;
    }
}

// ---- channel-chunked BN forward: finalize + apply in one launch (wide, small-spatial layers) ---------------------------
// Work-group = 64 channels x one pixel range, grid (C/64, S).  Prelude: the work-group column-sums its 64 channels of the conv
// epilogue's partial-statistics slab ([rows][2][C], rows <= 128 since the igemm adds its wave rows itself: <= 64 KB of
// L2-resident data, read as 16-byte vectors) in fp64 with a fixed order and derives scale / shift exactly as bn_finalize_k
// does; the sp == 0 work-groups also write the saved statistics and the running-statistics update.  Then it streams its pixels.
// (TY / TZ: element types of the pre-BN conv output and of the residual / output: elem_t, elem_t in the 16-bit modes; float, sp32 in the
// f16x2 mode, whose convolutions leave y in fp32 and whose activations are split tensors)
template <typename TY, typename TZ>
__global__ __launch_bounds__(TPB) void bn_apply_chunk_k(const TY* __restrict__ y, const TZ* __restrict__ res, TZ* __restrict__ z,
                                                        size_t npix, int C, const float* __restrict__ slab, int rows, double count,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        long long* __restrict__ nbt, float* __restrict__ save, int relu, int P,
                                                        unsigned char* __restrict__ mask, int xcd) {
    __shared__ double part[8][128];
    __shared__ float scs[64], shs[64];
    // xcd: the (chunk, pixel range) pairs are dealt so that XCD k gets the k-th EIGHTH of the pixel rows (all chunks of it) - the rows
    // the implicit GEMMs' work-groups on XCD k produced and will consume (igemm.hip: each XCD owns a contiguous range of m-tiles):
    // y is then read from, and z left in, that XCD's L2 (tools/probe/l2_handoff.hip: 17.9 against 6.7 TB/s across a kernel boundary)
    int chunk = blockIdx.x, sp = blockIdx.y;
    if (xcd) {
        const uint32_t w = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        sp = (int)(w / gridDim.x);
        chunk = (int)(w - (uint32_t)sp * gridDim.x);
    }
    {
        const int q = threadIdx.x & 31, rg = threadIdx.x >> 5;          // 16-byte column quad (16 per statistic), row group
        const float* base = slab + (size_t)(q >> 4) * C + chunk * 64 + (q & 15) * 4;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = rg;
        for (; r + 56 < rows; r += 64) {      // eight independent 16-byte loads in flight (same order of additions as the four-load form)
            const f32x4 u = *(const f32x4*)(base + (size_t)r * 2 * C), v = *(const f32x4*)(base + (size_t)(r + 8) * 2 * C);
            const f32x4 w = *(const f32x4*)(base + (size_t)(r + 16) * 2 * C), x = *(const f32x4*)(base + (size_t)(r + 24) * 2 * C);
            const f32x4 u2 = *(const f32x4*)(base + (size_t)(r + 32) * 2 * C), v2 = *(const f32x4*)(base + (size_t)(r + 40) * 2 * C);
            const f32x4 w2 = *(const f32x4*)(base + (size_t)(r + 48) * 2 * C), x2 = *(const f32x4*)(base + (size_t)(r + 56) * 2 * C);
            a0 += ((double)u[0] + (double)v[0]) + ((double)w[0] + (double)x[0]);
            a1 += ((double)u[1] + (double)v[1]) + ((double)w[1] + (double)x[1]);
            a2 += ((double)u[2] + (double)v[2]) + ((double)w[2] + (double)x[2]);
            a3 += ((double)u[3] + (double)v[3]) + ((double)w[3] + (double)x[3]);
            a0 += ((double)u2[0] + (double)v2[0]) + ((double)w2[0] + (double)x2[0]);
            a1 += ((double)u2[1] + (double)v2[1]) + ((double)w2[1] + (double)x2[1]);
            a2 += ((double)u2[2] + (double)v2[2]) + ((double)w2[2] + (double)x2[2]);
            a3 += ((double)u2[3] + (double)v2[3]) + ((double)w2[3] + (double)x2[3]);
        }
        for (; r + 24 < rows; r += 32) {      // four independent 16-byte loads in flight
            const f32x4 u = *(const f32x4*)(base + (size_t)r * 2 * C), v = *(const f32x4*)(base + (size_t)(r + 8) * 2 * C);
            const f32x4 w = *(const f32x4*)(base + (size_t)(r + 16) * 2 * C), x = *(const f32x4*)(base + (size_t)(r + 24) * 2 * C);
            a0 += ((double)u[0] + (double)v[0]) + ((double)w[0] + (double)x[0]);
            a1 += ((double)u[1] + (double)v[1]) + ((double)w[1] + (double)x[1]);
            a2 += ((double)u[2] + (double)v[2]) + ((double)w[2] + (double)x[2]);
            a3 += ((double)u[3] + (double)v[3]) + ((double)w[3] + (double)x[3]);
        }
        for (; r < rows; r += 8) {
            const f32x4 u = *(const f32x4*)(base + (size_t)r * 2 * C);
            a0 += (double)u[0]; a1 += (double)u[1]; a2 += (double)u[2]; a3 += (double)u[3];
        }
        part[rg][q * 4 + 0] = a0; part[rg][q * 4 + 1] = a1; part[rg][q * 4 + 2] = a2; part[rg][q * 4 + 3] = a3;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int col = threadIdx.x, c = chunk * 64 + col;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += part[k][col]; s2 += part[k][64 + col]; }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scs[col] = sc;
        shs[col] = beta[c] - (float)mean * sc;
        if (sp == 0) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            save[c] = (float)mean;
            save[C + c] = invstd;
            save[2 * C + c] = (float)unb;
            if (running_mean) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
            if (c == 0 && nbt) *nbt += 1;
        }
    }
    __syncthreads();
    const int cg = threadIdx.x & 7, prow = threadIdx.x >> 3;
    const int c0 = chunk * 64 + cg * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scs[cg * 8 + e]; sh[e] = shs[cg * 8 + e]; }
    const size_t p0 = (size_t)sp * P;
    size_t p1 = p0 + P;
    if (p1 > npix) p1 = npix;
    for (size_t p = p0 + prow; p < p1; p += 32) {
        const size_t off = p * C + c0;
        float v[8], o[8];
        ld8<TY>(y + off, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] * sc[e] + sh[e];
        if (res) {
            float r8[8];
            ld8<TZ>(res + off, r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += r8[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (relu && o[e] < 0.f) ? 0.f : o[e];
        st8<TZ>(z + off, o);
        if constexpr (sizeof(TZ) == sizeof(elem_t)) {
            if (mask) {
                unsigned mb = 0u;
#pragma unroll
                for (int e = 0; e < 8; ++e) mb |= ((float)(elem_t)o[e] > 0.f ? 1u : 0u) << e;
                mask[off >> 3] = (unsigned char)mb;
            }
        }
    }
}

// ---- channel-chunked BN backward (wide, small-spatial layers: C >= 256, <= 32 K pixels) ------------------------------
// Work-group = 64 channels (one 128-byte segment of every pixel row) x one pixel range; grid (C/64, S).  The reduce writes
// S partial rows per chunk ([chunk][S][2][64] floats, <= 32 KB per chunk), and every apply work-group sums its chunk's rows
// itself (fp64, fixed order) before streaming: the separate finalize launch (6-7 us on the critical path of every layer)
// is gone, at the price of a <= 32 KB L2-resident prelude per work-group.
template <typename DZ>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_chunk_k(const DZ* __restrict__ dz, const elem_t* __restrict__ z, const elem_t* __restrict__ y,
                                                             size_t npix, int C, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int relu, float* __restrict__ slab, int P,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta) {
    __shared__ float red[32][8][17];
    const int chunk = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int cg = threadIdx.x & 7, prow = threadIdx.x >> 3;
    const int c0 = chunk * 64 + cg * 8;
    float mu[8], is[8], s1[8], s2[8], sc[8], sh[8];
    load8<float>(mean + c0, mu);
    load8<float>(invstd + c0, is);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; sc[e] = 0.f; sh[e] = 0.f; }
    if (relu == 2) {
        float ga[8], be[8];
        load8<float>(gamma + c0, ga);
        load8<float>(beta + c0, be);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = ga[e] * is[e]; sh[e] = be[e] - mu[e] * sc[e]; }
    }
    const size_t p0 = (size_t)sp * P;
    size_t p1 = p0 + P;
    if (p1 > npix) p1 = npix;
    for (size_t p = p0 + prow; p < p1; p += 32) {
        const size_t off = p * C + c0;
        float d[8];
        load8<DZ>(dz + off, d);
        const elem8 yy = *(const elem8*)(y + off);
        elem8 zz = {};
        if (relu == 1) zz = *(const elem8*)(z + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float gv = d[e];
            if (relu == 1 && !((float)zz[e] > 0.f)) gv = 0.f;
            if (relu == 2 && !((float)yy[e] * sc[e] + sh[e] > 0.f)) gv = 0.f;
            s1[e] += gv;
            s2[e] += gv * (((float)yy[e] - mu[e]) * is[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[prow][cg][e] = s1[e]; red[prow][cg][8 + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
        float a = 0.f;
        for (int k = 0; k < 32; ++k) a += red[k][g][e];
        slab[(((size_t)chunk * S + sp) * 2 + (e >> 3)) * 64 + g * 8 + (e & 7)] = a;
    }
}
template <typename DZ>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_chunk_k(const DZ* __restrict__ dz, const elem_t* __restrict__ z, const elem_t* __restrict__ y,
                                                            elem_t* __restrict__ dy, elem_t* __restrict__ gout, size_t npix, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ slab, int P, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float beta_acc) {
    __shared__ double tot[2][64];
    const int chunk = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    if (threadIdx.x < 128) {
        const int st = threadIdx.x >> 6, col = threadIdx.x & 63;
        const float* q = slab + ((size_t)chunk * S * 2 + st) * 64 + col;
        double a = 0.0;
        int k = 0;
        for (; k + 4 <= S; k += 4)        // four independent loads in flight; the order of the fp64 additions is fixed
            a += ((double)q[(size_t)k * 128] + (double)q[(size_t)(k + 1) * 128]) + ((double)q[(size_t)(k + 2) * 128] + (double)q[(size_t)(k + 3) * 128]);
        for (; k < S; ++k) a += (double)q[(size_t)k * 128];
        tot[st][col] = a;
    }
    __syncthreads();
    if (sp == 0 && threadIdx.x < 64 && dgamma) {
        const int c = chunk * 64 + threadIdx.x;
        dgamma[c] = (beta_acc != 0.f ? beta_acc * dgamma[c] : 0.f) + (float)tot[1][threadIdx.x];
        dbeta[c] = (beta_acc != 0.f ? beta_acc * dbeta[c] : 0.f) + (float)tot[0][threadIdx.x];
    }
    const int cg = threadIdx.x & 7, prow = threadIdx.x >> 3;
    const int c0 = chunk * 64 + cg * 8;
    const double count = (double)npix;
    float mu[8], is[8], ca[8], cb[8], cc[8], sc[8], sh[8], ga[8];
    load8<float>(mean + c0, mu);
    load8<float>(invstd + c0, is);
    load8<float>(gamma + c0, ga);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ca[e] = ga[e] * is[e];
        cb[e] = (float)(tot[0][cg * 8 + e] / count);
        cc[e] = (float)(tot[1][cg * 8 + e] / count);
        sc[e] = 0.f; sh[e] = 0.f;
    }
    if (relu == 2) {
        float be[8];
        load8<float>(beta + c0, be);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = ga[e] * is[e]; sh[e] = be[e] - mu[e] * sc[e]; }
    }
    const size_t p0 = (size_t)sp * P;
    size_t p1 = p0 + P;
    if (p1 > npix) p1 = npix;
    for (size_t p = p0 + prow; p < p1; p += 32) {
        const size_t off = p * C + c0;
        float d[8];
        load8<DZ>(dz + off, d);
        const elem8 yy = *(const elem8*)(y + off);
        elem8 zz = {};
        if (relu == 1) zz = *(const elem8*)(z + off);
        elem8 o, go;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float gv = d[e];
            if (relu == 1 && !((float)zz[e] > 0.f)) gv = 0.f;
            if (relu == 2 && !((float)yy[e] * sc[e] + sh[e] > 0.f)) gv = 0.f;
            const float xh = ((float)yy[e] - mu[e]) * is[e];
            o[e] = (elem_t)(ca[e] * (gv - cb[e] - xh * cc[e]));
            go[e] = (elem_t)gv;
        }
        *(elem8*)(dy + off) = o;
        if (gout) *(elem8*)(gout + off) = go;
    }
}

// dy = ca*(g - cb - xhat*cc) for an already masked g (pre-reduced form, layers the chunked kernel does not take): the grid
// keeps gridDim.x * TPB a multiple of C/8 (bn_apply_grid), so a thread's 8 channels never change and its 40 coefficients are
// loaded once instead of per 16-byte access; no integer division in the loop.
template <typename DZ>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_pre_k(const DZ* __restrict__ g, const elem_t* __restrict__ y, elem_t* __restrict__ dy, size_t n8,
                                                          int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ coef, int xcd = 0) {
    const int G = C >> 3;
    const int c0 = (int)(((size_t)blockIdx.x * TPB + threadIdx.x) % G) * 8;
    float mu[8], is[8], ca[8], cb[8], cc[8];
    load8<float>(mean + c0, mu);
    load8<float>(invstd + c0, is);
    load8<float>(coef + c0, ca);
    load8<float>(coef + C + c0, cb);
    load8<float>(coef + 2 * C + c0, cc);
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x, iend = n8, istep = (size_t)gridDim.x * TPB;
    if (xcd) {      // (XCD k <- the k-th eighth of the pixel rows, as in bn_apply_k)
        const size_t rows = n8 / G, per = (rows + 7) / 8;
        const unsigned x = blockIdx.x & 7u, local = blockIdx.x >> 3, nb = gridDim.x >> 3;
        const unsigned RB = TPB / G;
        size_t rend = (size_t)(x + 1) * per;
        if (rend > rows) rend = rows;
        i = ((size_t)x * per + (size_t)local * RB + threadIdx.x / G) * G + threadIdx.x % G;
        iend = rend * G;
        istep = (size_t)nb * RB * G;
    }
    for (; i < iend; i += istep) {
        float d[8];
        load8<DZ>(g + i * 8, d);
        const elem8 yy = *(const elem8*)(y + i * 8);
        elem8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)yy[e] - mu[e]) * is[e];
            o[e] = (elem_t)(ca[e] * (d[e] - cb[e] - xh * cc[e]));
        }
        *(elem8*)(dy + i * 8) = o;
    }
}

// ---- BN backward whose reduction was done by the producing dgrad's epilogue (igemm.hip, BS mode) -----------------------
// dz already carries the ReLU mask (g), and slab[rows][2][C] holds one partial row of (sum g, sum g*xhat) per m-tile of that
// dgrad launch.  Work-group = 64 channels x one pixel range, grid (C/64, S): prelude = fp64 column sums of the work-group's 64
// channels (rows <= 128: <= 64 KB of L2-resident data read as 16-byte vectors, fixed order), then dy = ca*(g - cb - xhat*cc).
template <typename DZ>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_pre_chunk_k(const DZ* __restrict__ g, const elem_t* __restrict__ y, elem_t* __restrict__ dy,
                                                                size_t npix, int C, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ slab, int rows, int P, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float beta_acc, int xcd) {
    __shared__ double part[8][128];
    __shared__ double tot[2][64];
    int chunk = blockIdx.x, sp = blockIdx.y;
    if (xcd) {      // (XCD k <- the k-th eighth of the pixel rows, as in bn_apply_chunk_k)
        const uint32_t w = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        sp = (int)(w / gridDim.x);
        chunk = (int)(w - (uint32_t)sp * gridDim.x);
    }
    {
        const int q = threadIdx.x & 31, rg = threadIdx.x >> 5;          // 16-byte column quad (16 per statistic), row group
        const float* base = slab + (size_t)(q >> 4) * C + chunk * 64 + (q & 15) * 4;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = rg;
        for (; r + 56 < rows; r += 64) {      // eight independent 16-byte loads in flight (same order of additions as the four-load form)
            const f32x4 u = *(const f32x4*)(base + (size_t)r * 2 * C), v = *(const f32x4*)(base + (size_t)(r + 8) * 2 * C);
            const f32x4 w = *(const f32x4*)(base + (size_t)(r + 16) * 2 * C), x = *(const f32x4*)(base + (size_t)(r + 24) * 2 * C);
            const f32x4 u2 = *(const f32x4*)(base + (size_t)(r + 32) * 2 * C), v2 = *(const f32x4*)(base + (size_t)(r + 40) * 2 * C);
            const f32x4 w2 = *(const f32x4*)(base + (size_t)(r + 48) * 2 * C), x2 = *(const f32x4*)(base + (size_t)(r + 56) * 2 * C);
            a0 += ((double)u[0] + (double)v[0]) + ((double)w[0] + (double)x[0]);
            a1 += ((double)u[1] + (double)v[1]) + ((double)w[1] + (double)x[1]);
            a2 += ((double)u[2] + (double)v[2]) + ((double)w[2] + (double)x[2]);
            a3 += ((double)u[3] + (double)v[3]) + ((double)w[3] + (double)x[3]);
            a0 += ((double)u2[0] + (double)v2[0]) + ((double)w2[0] + (double)x2[0]);
            a1 += ((double)u2[1] + (double)v2[1]) + ((double)w2[1] + (double)x2[1]);
            a2 += ((double)u2[2] + (double)v2[2]) + ((double)w2[2] + (double)x2[2]);
            a3 += ((double)u2[3] + (double)v2[3]) + ((double)w2[3] + (double)x2[3]);
        }
        for (; r + 24 < rows; r += 32) {      // four independent 16-byte loads in flight
            const f32x4 u = *(const f32x4*)(base + (size_t)r * 2 * C), v = *(const f32x4*)(base + (size_t)(r + 8) * 2 * C);
            const f32x4 w = *(const f32x4*)(base + (size_t)(r + 16) * 2 * C), x = *(const f32x4*)(base + (size_t)(r + 24) * 2 * C);
            a0 += ((double)u[0] + (double)v[0]) + ((double)w[0] + (double)x[0]);
            a1 += ((double)u[1] + (double)v[1]) + ((double)w[1] + (double)x[1]);
            a2 += ((double)u[2] + (double)v[2]) + ((double)w[2] + (double)x[2]);
            a3 += ((double)u[3] + (double)v[3]) + ((double)w[3] + (double)x[3]);
        }
        for (; r < rows; r += 8) {
            const f32x4 u = *(const f32x4*)(base + (size_t)r * 2 * C);
            a0 += (double)u[0]; a1 += (double)u[1]; a2 += (double)u[2]; a3 += (double)u[3];
        }
        part[rg][q * 4 + 0] = a0; part[rg][q * 4 + 1] = a1; part[rg][q * 4 + 2] = a2; part[rg][q * 4 + 3] = a3;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += part[k][threadIdx.x];
        tot[threadIdx.x >> 6][threadIdx.x & 63] = a;
    }
    __syncthreads();
    if (sp == 0 && threadIdx.x < 64 && dgamma) {
        const int c = chunk * 64 + threadIdx.x;
        dgamma[c] = (beta_acc != 0.f ? beta_acc * dgamma[c] : 0.f) + (float)tot[1][threadIdx.x];
        dbeta[c] = (beta_acc != 0.f ? beta_acc * dbeta[c] : 0.f) + (float)tot[0][threadIdx.x];
    }
    const int cg = threadIdx.x & 7, prow = threadIdx.x >> 3;
    const int c0 = chunk * 64 + cg * 8;
    const double count = (double)npix;
    float mu[8], is[8], ca[8], cb[8], cc[8], ga[8];
    load8<float>(mean + c0, mu);
    load8<float>(invstd + c0, is);
    load8<float>(gamma + c0, ga);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ca[e] = ga[e] * is[e];
        cb[e] = (float)(tot[0][cg * 8 + e] / count);
        cc[e] = (float)(tot[1][cg * 8 + e] / count);
    }
    const size_t p0 = (size_t)sp * P;
    size_t p1 = p0 + P;
    if (p1 > npix) p1 = npix;
    for (size_t p = p0 + prow; p < p1; p += 32) {
        const size_t off = p * C + c0;
        float d[8];
        load8<DZ>(g + off, d);
        const elem8 yy = *(const elem8*)(y + off);
        elem8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)yy[e] - mu[e]) * is[e];
            o[e] = (elem_t)(ca[e] * (d[e] - cb[e] - xh * cc[e]));
        }
        *(elem8*)(dy + off) = o;
    }
}

// ------------------------------------------------------------------ max pooling (NHWC bf16)
// 3x3 stride 2 pad 1 (ResNet stem).  Saves the winning tap (0..8, first max in (kh,kw) scan order like ATen) per element.
template <typename T>
__global__ void maxpool3x3s2_fwd_k(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx, int N, int H,
                                   int W, int C, int Ho, int Wo) {
    const int G = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int g = (int)(i % G);
        size_t r = i / G;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float best[8];
        unsigned char bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = ho * 2 - 1 + kh, w = wo * 2 - 1 + kw;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                    float v[8];
                    ld8<T>(x + (((size_t)n * H + h) * W + w) * C + g * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = v[e];
                        if (first || f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(kh * 3 + kw); }
                    }
                    first = false;
                }
            }
        st8<T>(y + i * 8, best);
        if (idx) {
            unsigned long long pk = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk |= (unsigned long long)bi[e] << (8 * e);
            *(unsigned long long*)(idx + i * 8) = pk;
        }
    }
}
// The stem's BatchNorm apply + ReLU + 3x3 stride-2 max-pool in ONE sweep: z = relu(y*scale + shift) is formed per tap in registers,
// rounded to the storage type (so values, ties and winning taps are exactly those of bn_apply_k followed by maxpool3x3s2_fwd_k) and
// never written: the backward recomputes the ReLU mask from y.  Saves one 67 MB write and one (1.5x amplified) read per pass.
__global__ void bn_relu_maxpool3x3s2_k(const elem_t* __restrict__ x, elem_t* __restrict__ y, unsigned char* __restrict__ idx, int N, int H, int W,
                                       int C, int Ho, int Wo, const float* __restrict__ scale, const float* __restrict__ shift) {
    const int G = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int g = (int)(i % G);
        size_t r = i / G;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float sc[8], sh[8];
        load8<float>(scale + g * 8, sc);
        load8<float>(shift + g * 8, sh);
        float best[8];
        unsigned char bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = ho * 2 - 1 + kh, w = wo * 2 - 1 + kw;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                    float v[8];
                    ld8<elem_t>(x + (((size_t)n * H + h) * W + w) * C + g * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float o = v[e] * sc[e] + sh[e];
                        o = o < 0.f ? 0.f : o;                       // (bn_apply_k's expression and ReLU)
                        const float f = (float)(elem_t)o;            // ... and its rounding to the storage type
                        if (first || f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(kh * 3 + kw); }
                    }
                    first = false;
                }
            }
        st8<elem_t>(y + i * 8, best);
        unsigned long long pk = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk |= (unsigned long long)bi[e] << (8 * e);
        *(unsigned long long*)(idx + i * 8) = pk;
    }
}
// gather-style backward (no atomics): each input pixel checks the <=4 windows that contain it
__global__ void maxpool3x3s2_bwd_k(const elem_t* __restrict__ dy, const unsigned char* __restrict__ idx, elem_t* __restrict__ dx, int N,
                                   int H, int W, int C, int Ho, int Wo) {
    const int G = C >> 3;
    const size_t total = (size_t)N * H * W * G;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int g = (int)(i % G);
        size_t r = i / G;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        const int ho0 = h >> 1, wo0 = w >> 1;     // windows with ho in {ho0, ho0+1 if h odd}: 2ho-1 <= h <= 2ho+1
        for (int ho = ho0; ho <= ((h + 1) >> 1); ++ho)
            for (int wo = wo0; wo <= ((w + 1) >> 1); ++wo) {
                if (ho >= Ho || wo >= Wo) continue;
                const int kh = h - (ho * 2 - 1), kw = w - (wo * 2 - 1);
                if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + g * 8;
                const unsigned long long pk = *(const unsigned long long*)(idx + o);
                const elem8 d = *(const elem8*)(dy + o);
                const unsigned tap = (unsigned)(kh * 3 + kw);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (((pk >> (8 * e)) & 0xFF) == tap) acc[e] += (float)d[e];
            }
        elem8 o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = (elem_t)acc[e];
        *(elem8*)(dx + i * 8) = o8;
    }
}
// 2x2 stride 2 ceil-mode (VGG encoder, inference only)
template <typename T>
__global__ void maxpool2x2_ceil_k(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
    const int G = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int g = (int)(i % G);
        size_t r = i / G;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
        for (int kh = 0; kh < 2; ++kh)
            for (int kw = 0; kw < 2; ++kw) {
                const int h = ho * 2 + kh, w = wo * 2 + kw;
                if (h < H && w < W) {
                    float v[8];
                    ld8<T>(x + (((size_t)n * H + h) * W + w) * C + g * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], v[e]);
                }
            }
        st8<T>(y + i * 8, best);
    }
}

// per-plane sums of an NCHW fp32 tensor over (n, hw): out[c] (beta*old +) = sum  (head bias gradient)
__global__ __launch_bounds__(1024) void plane_sum_k(const float* __restrict__ x, float* __restrict__ out, int N, int C, int HW, float beta) {
    const int c = blockIdx.x;
    double s = 0.0;
    if ((HW & 3) == 0) {
        const int hw4 = HW >> 2;
        for (int i = threadIdx.x; i < hw4; i += 1024) {
            int n = 0;
            for (; n + 4 <= N; n += 4) {          // four independent 16-byte loads in flight
                const f32x4 a = *(const f32x4*)(x + ((size_t)n * C + c) * HW + 4 * i);
                const f32x4 b = *(const f32x4*)(x + ((size_t)(n + 1) * C + c) * HW + 4 * i);
                const f32x4 d = *(const f32x4*)(x + ((size_t)(n + 2) * C + c) * HW + 4 * i);
                const f32x4 e = *(const f32x4*)(x + ((size_t)(n + 3) * C + c) * HW + 4 * i);
                s += (double)((a[0] + a[1]) + (a[2] + a[3])) + (double)((b[0] + b[1]) + (b[2] + b[3])) +
                     (double)((d[0] + d[1]) + (d[2] + d[3])) + (double)((e[0] + e[1]) + (e[2] + e[3]));
            }
            for (; n < N; ++n) {
                const f32x4 a = *(const f32x4*)(x + ((size_t)n * C + c) * HW + 4 * i);
                s += (double)((a[0] + a[1]) + (a[2] + a[3]));
            }
        }
    } else {
        for (int n = 0; n < N; ++n)
            for (int i = threadIdx.x; i < HW; i += 1024) s += (double)x[((size_t)n * C + c) * HW + i];
    }
    __shared__ double red[16];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[c] = (beta != 0.f ? beta * out[c] : 0.f) + (float)t;
    }
}

inline int grid_for(size_t items) {
    size_t b = (items + TPB - 1) / TPB;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// ------------------------------------------------------------------ host launchers (internal C++ API; C-ABI wrappers in capi.hip)
int pw_nchw_f32_to_nhwc_bf16(hipStream_t s, const float* src, elem_t* dst, int N, int C, int HW, int Cp) {
    if (Cp % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_k<elem_t>, dim3(grid_for((size_t)N * HW * (Cp / 8))), dim3(TPB), 0, s, src, dst, N, C, HW, Cp);
    return udapose_check_launch();
}
int pw_nchw_f32_to_nhwc_f32(hipStream_t s, const float* src, float* dst, int N, int C, int HW, int Cp) {
    if (Cp % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_k<float>, dim3(grid_for((size_t)N * HW * (Cp / 8))), dim3(TPB), 0, s, src, dst, N, C, HW, Cp);
    return udapose_check_launch();
}
int pw_nchw_f32_to_nhwc_split(hipStream_t s, const float* src, void* dst, int N, int C, int HW, int Cp) {
    if (Cp % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_k<sp32>, dim3(grid_for((size_t)N * HW * (Cp / 8))), dim3(TPB), 0, s, src, (sp32*)dst, N, C, HW, Cp);
    return udapose_check_launch();
}
// src_is_f32: 0 = element type, 1 = fp32, 2 = f16x2 split
int pw_nhwc_to_nchw_f32(hipStream_t s, const void* src, int src_is_f32, float* dst, int N, int C, int HW, int Cs, const float* lo, const float* hi) {
    if (src_is_f32 == 2)
        hipLaunchKernelGGL(nhwc_to_nchw_f32_k<sp32>, dim3(grid_for((size_t)N * HW)), dim3(TPB), 0, s, (const sp32*)src, dst, N, C, HW, Cs, lo, hi);
    else if (src_is_f32)
        hipLaunchKernelGGL(nhwc_to_nchw_f32_k<float>, dim3(grid_for((size_t)N * HW)), dim3(TPB), 0, s, (const float*)src, dst, N, C, HW, Cs, lo, hi);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_f32_k<elem_t>, dim3(grid_for((size_t)N * HW)), dim3(TPB), 0, s, (const elem_t*)src, dst, N, C, HW, Cs, lo, hi);
    return udapose_check_launch();
}
int pw_cast_f32_bf16(hipStream_t s, const float* src, elem_t* dst, size_t n) {
    if (n % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(cast_f32_bf16_k, dim3(grid_for(n / 8)), dim3(TPB), 0, s, src, dst, n / 8);
    return udapose_check_launch();
}
int pw_transpose_cast(hipStream_t s, const float* src, elem_t* dst, int A, int T, int B) {
    hipLaunchKernelGGL(transpose_cast_k<elem_t>, dim3((B + 31) / 32, (A + 31) / 32, T), dim3(TPB), 0, s, src, dst, A, T, B);
    return udapose_check_launch();
}
int pw_transpose_f32(hipStream_t s, const float* src, float* dst, int A, int T, int B) {
    hipLaunchKernelGGL(transpose_cast_k<float>, dim3((B + 31) / 32, (A + 31) / 32, T), dim3(TPB), 0, s, src, dst, A, T, B);
    return udapose_check_launch();
}
// fp32 -> f16x2 split, n % 8 == 0 (src == dst allowed: every thread reads its 32 bytes before it writes them)
__global__ void f32_to_split_k(const float* src, sp32* dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (size_t)gridDim.x * TPB) {
        float v[8];
        ld8<float>(src + i * 8, v);
        st8<sp32>(dst + i * 8, v);
    }
}
__global__ void split_to_f32_k(const sp32* src, float* dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (size_t)gridDim.x * TPB) {
        float v[8];
        ld8<sp32>(src + i * 8, v);
        st8<float>(dst + i * 8, v);
    }
}
int pw_f32_to_split(hipStream_t s, const float* src, void* dst, size_t n) {
    if (n % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(f32_to_split_k, dim3(grid_for(n / 8)), dim3(TPB), 0, s, src, (sp32*)dst, n / 8);
    return udapose_check_launch();
}
int pw_split_to_f32(hipStream_t s, const void* src, float* dst, size_t n) {
    if (n % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(split_to_f32_k, dim3(grid_for(n / 8)), dim3(TPB), 0, s, (const sp32*)src, dst, n / 8);
    return udapose_check_launch();
}
int pw_transpose_split(hipStream_t s, const float* src, void* dst, int A, int T, int B) {
    hipLaunchKernelGGL(transpose_cast_k<sp32>, dim3((B + 31) / 32, (A + 31) / 32, T), dim3(TPB), 0, s, src, (sp32*)dst, A, T, B);
    return udapose_check_launch();
}
int pw_pack_strided_split(hipStream_t s, const float* src, void* dst, int A, int KH, int KWp, int KW, int Bp, int B, long sa, long skh, long skw, long sb) {
    hipLaunchKernelGGL(pack_strided_k<sp32>, dim3(grid_for((size_t)A * KH * KWp * Bp)), dim3(TPB), 0, s, src, (sp32*)dst, A, KH, KWp, KW, Bp, B, sa, skh, skw, sb);
    return udapose_check_launch();
}
int pw_pack_multi(hipStream_t s, const void* jobs, const int* blk_job, const int* blk_sub, int nblocks) {
    if (nblocks <= 0) return UDAPOSE_OK;
    hipLaunchKernelGGL(pack_multi_k, dim3(nblocks), dim3(TPB), 0, s, (const PackJob*)jobs, blk_job, blk_sub);
    return udapose_check_launch();
}
int pw_pack_strided(hipStream_t s, const float* src, elem_t* dst, int A, int KH, int KWp, int KW, int Bp, int B, long sa, long skh, long skw, long sb) {
    hipLaunchKernelGGL(pack_strided_k<elem_t>, dim3(grid_for((size_t)A * KH * KWp * Bp)), dim3(TPB), 0, s, src, dst, A, KH, KWp, KW, Bp, B, sa, skh, skw, sb);
    return udapose_check_launch();
}
int pw_pack_strided_f32(hipStream_t s, const float* src, float* dst, int A, int KH, int KWp, int KW, int Bp, int B, long sa, long skh, long skw, long sb) {
    hipLaunchKernelGGL(pack_strided_k<float>, dim3(grid_for((size_t)A * KH * KWp * Bp)), dim3(TPB), 0, s, src, dst, A, KH, KWp, KW, Bp, B, sa, skh, skw, sb);
    return udapose_check_launch();
}
int pw_unpack_strided(hipStream_t s, const float* src, float* dst, int A, int KH, int KWp, int KW, int Bp, int B, long sa, long skh, long skw, long sb, float beta) {
    hipLaunchKernelGGL(unpack_strided_k, dim3(grid_for((size_t)A * KH * KW * B)), dim3(TPB), 0, s, src, dst, A, KH, KWp, KW, Bp, B, sa, skh, skw, sb, beta);
    return udapose_check_launch();
}
int pw_bn_finalize(hipStream_t s, const float* slab, int rows, int C, double count, const float* gamma, const float* beta, float* rm, float* rv,
                   long long* nbt, float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_invstd, const float* pre_bias) {
    hipLaunchKernelGGL(bn_finalize_k, dim3((C + FIN_C - 1) / FIN_C), dim3(FIN_T), 0, s, slab, rows, C, count, gamma, beta, rm, rv, nbt, momentum, eps, scale, shift,
                       save_mean, save_invstd, pre_bias);
    return udapose_check_launch();
}
// finalize + apply in one launch where the chunked form applies; returns 1 when it took the layer, 0 when the caller must use
// pw_bn_finalize + pw_bn_apply, < 0 on error
int pw_bn_train_fused(hipStream_t s, const elem_t* y, const elem_t* res, elem_t* z, size_t npix, int C, const float* slab, int rows,
                      const float* gamma, const float* beta, float* rm, float* rv, long long* nbt, float momentum, float eps, float* save,
                      int relu, int enabled, unsigned char* mask) {
    // layer3 / layer4 / the first deconv (<= 8 K pixels, <= 128 slab rows): measured -0.1 ms per step; with the 32 K-pixel
    // layers included the 128-byte row segments of the chunked layout cost what the saved launches gain
    // (enabled: Policy::bn_fwd_chunked)
    const int xcd = (enabled >> 30) & 1;                   // (policy bit 30: XCD-aligned pixel ranges)
    enabled &= ~(1 << 30);
    if (!enabled || C < 256 || C % 64 || npix > 8192 || npix < 1024 || rows > 128) return 0;
    const int chunks = C / 64;
    int S = (enabled > 1 ? enabled : 1024) / chunks;      // (policy value > 1: the target work-group count; tuning)
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    int P = (int)((npix + S - 1) / S);
    P = (P + 31) & ~31;
    S = (int)((npix + P - 1) / P);
    hipLaunchKernelGGL((bn_apply_chunk_k<elem_t, elem_t>), dim3(chunks, S), dim3(TPB), 0, s, y, res, z, npix, C, slab, rows, (double)npix, gamma, beta, eps,
                       momentum, rm, rv, nbt, save, relu, P, mask, xcd);
    return udapose_check_launch() == UDAPOSE_OK ? 1 : UDAPOSE_ERR_LAUNCH;
}
// the same for the f16x2 mode: y fp32 (the split convolutions' pre-BN output), res / z split tensors.  On the teacher's forward - the
// critical path of a step in the reference precision mix - this removes one launch (bn_finalize_k) per BatchNorm of layer3 / layer4 / deconv1.
int pw_bn_train_fused_split(hipStream_t s, const float* y, const void* res, void* z, size_t npix, int C, const float* slab, int rows,
                            const float* gamma, const float* beta, float* rm, float* rv, long long* nbt, float momentum, float eps, float* save,
                            int relu, int enabled) {
    const int xcd = (enabled >> 30) & 1;
    enabled &= ~(1 << 30);
    if (!enabled || C < 256 || C % 64 || npix > 8192 || npix < 1024 || rows > 128) return 0;
    const int chunks = C / 64;
    int S = (enabled > 1 ? enabled : 1024) / chunks;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    int P = (int)((npix + S - 1) / S);
    P = (P + 31) & ~31;
    S = (int)((npix + P - 1) / P);
    hipLaunchKernelGGL((bn_apply_chunk_k<float, sp32>), dim3(chunks, S), dim3(TPB), 0, s, y, (const sp32*)res, (sp32*)z, npix, C, slab, rows, (double)npix,
                       gamma, beta, eps, momentum, rm, rv, nbt, save, relu, P, (unsigned char*)nullptr, xcd);
    return udapose_check_launch() == UDAPOSE_OK ? 1 : UDAPOSE_ERR_LAUNCH;
}
// the same for every BN layer of a net in one launch: jobs[blockIdx.x], channels blockIdx.y*TPB..; save = act + save_off
__global__ void bn_running_update_multi_k(const BnRunJob* __restrict__ jobs, const char* __restrict__ act, float momentum) {
    const BnRunJob j = jobs[blockIdx.x];
    const int c = blockIdx.y * TPB + threadIdx.x;
    if (c == 0 && j.nbt) *j.nbt += 1;
    if (c >= j.C) return;
    const float* save = (const float*)(act + j.save_off);
    j.rm[c] = (1.f - momentum) * j.rm[c] + momentum * save[c];
    j.rv[c] = (1.f - momentum) * j.rv[c] + momentum * save[2 * j.C + c];
}
int pw_bn_running_update_multi(hipStream_t s, const BnRunJob* d_jobs, int njobs, int maxC, const void* act, float momentum) {
    hipLaunchKernelGGL(bn_running_update_multi_k, dim3(njobs, nblk(maxC)), dim3(TPB), 0, s, d_jobs, (const char*)act, momentum);
    return udapose_check_launch();
}
int pw_bn_running_update(hipStream_t s, const float* save, int C, float* rm, float* rv, long long* nbt, float momentum) {
    hipLaunchKernelGGL(bn_running_update_k, dim3(nblk(C)), dim3(TPB), 0, s, save, C, rm, rv, nbt, momentum);
    return udapose_check_launch();
}
// clears several ranges in one launch: grid (jobs, 32); replaces one hipMemsetAsync node per split weight gradient
__global__ void zero_multi_k(const ZeroJob* __restrict__ jobs, char* __restrict__ base) {
    const ZeroJob j = jobs[blockIdx.x];
    u32x4* p = (u32x4*)(base + j.off);
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (long long i = (long long)blockIdx.y * TPB + threadIdx.x; i < j.n16; i += (long long)gridDim.y * TPB) p[i] = z;
}
int pw_zero_multi(hipStream_t s, const ZeroJob* d_jobs, int njobs, void* base) {
    if (njobs <= 0) return UDAPOSE_OK;
    hipLaunchKernelGGL(zero_multi_k, dim3(njobs, 32), dim3(TPB), 0, s, d_jobs, (char*)base);
    return udapose_check_launch();
}
// one range cleared by a kernel (common.h: why not hipMemsetAsync): 16-byte stores over the aligned body, 4-byte stores over the ragged ends
__global__ void zero_k(unsigned int* __restrict__ p, size_t head, size_t n16, size_t tail) {
    const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x, step = (size_t)gridDim.x * TPB;
    if (t < head) p[t] = 0u;
    u32x4* b = (u32x4*)(p + head);
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (size_t i = t; i < n16; i += step) b[i] = z;
    if (t < tail) p[head + n16 * 4 + t] = 0u;
}
int pw_zero(hipStream_t s, void* p, size_t bytes) {
    if (!bytes) return UDAPOSE_OK;
    if (!p || (bytes & 3) || ((uintptr_t)p & 3)) return UDAPOSE_ERR_ARG;
    size_t words = bytes / 4, head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;
    if (head > words) head = words;
    const size_t n16 = (words - head) / 4, tail = (words - head) & 3;
    size_t g = (n16 + TPB - 1) / TPB;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(zero_k, dim3((unsigned)g), dim3(TPB), 0, s, (unsigned int*)p, head, n16, tail);
    return udapose_check_launch();
}
// Deterministic split reductions of the grouped weight-gradient launches (net.hip build_wg_group): split z of a layer stored its partial tile at
// part + z * stride; dst = beta * dst + ((p0 + p1) + p2) + ... in split order.  One block per UDAPOSE_SPLIT_SUM_CHUNK elements of a job (blk: (job, chunk) pairs).
// (pair form: blocks [nblk, 2 nblk) run the same jobs on the second pass's workspace and gradient base - the two passes of one plan in ONE launch)
__global__ __launch_bounds__(TPB) void split_sum_k(const SumJob* __restrict__ jobs, const int* __restrict__ blk, int nblk, char* __restrict__ ws, char* __restrict__ gbase,
                                                   char* __restrict__ ws2, char* __restrict__ gbase2) {
    int bi = blockIdx.x;
    if (bi >= nblk) { bi -= nblk; ws = ws2; gbase = gbase2; }
    const SumJob j = jobs[blk[2 * bi]];
    const unsigned c0 = (unsigned)blk[2 * bi + 1] * UDAPOSE_SPLIT_SUM_CHUNK;
    const unsigned c1 = c0 + UDAPOSE_SPLIT_SUM_CHUNK < j.n ? c0 + UDAPOSE_SPLIT_SUM_CHUNK : j.n;
    const float* part = (const float*)(ws + j.part_off);
    float* dst = (float*)((j.dst_ws ? ws : gbase) + j.dst_off);
    if (((((uintptr_t)part) | ((uintptr_t)dst)) & 15) == 0 && (j.stride & 3u) == 0 && (j.n & 3u) == 0) {
        for (unsigned e = c0 + threadIdx.x * 4u; e < c1; e += TPB * 4u) {
            f32x4 a = *(const f32x4*)(part + e);
#pragma unroll 8
            for (int z = 1; z < j.ks; ++z) a += *(const f32x4*)(part + (size_t)z * j.stride + e);        // (loads hoisted in groups of eight, adds in split order)
            if (j.beta != 0.f) a = *(const f32x4*)(dst + e) + a;
            *(f32x4*)(dst + e) = a;
        }
    } else {
        for (unsigned e = c0 + threadIdx.x; e < c1; e += TPB) {
            float a = part[e];
            for (int z = 1; z < j.ks; ++z) a += part[(size_t)z * j.stride + e];
            if (j.beta != 0.f) a = dst[e] + a;
            dst[e] = a;
        }
    }
}
int pw_split_sum(hipStream_t s, const SumJob* d_jobs, const int* d_blk, int nblk, void* ws, void* grad_base, void* ws2, void* grad_base2) {
    if (nblk <= 0) return UDAPOSE_OK;
    hipLaunchKernelGGL(split_sum_k, dim3(ws2 ? 2 * nblk : nblk), dim3(TPB), 0, s, d_jobs, d_blk, nblk, (char*)ws, (char*)grad_base, (char*)ws2, (char*)grad_base2);
    return udapose_check_launch();
}
int pw_axpy(hipStream_t s, float* y, const float* x, size_t n) {
    const size_t n4 = n / 4;
    if (n4) hipLaunchKernelGGL(axpy_k, dim3(grid_for(n4)), dim3(TPB), 0, s, y, x, n4);
    if (n % 4) hipLaunchKernelGGL(axpy_tail_k, dim3(1), dim3(64), 0, s, y + n4 * 4, x + n4 * 4, (int)(n % 4));
    return udapose_check_launch();
}
int pw_bn_eval_coeff(hipStream_t s, int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale, float* shift) {
    hipLaunchKernelGGL(bn_eval_coeff_k, dim3(nblk(C)), dim3(TPB), 0, s, C, gamma, beta, rm, rv, eps, scale, shift);
    return udapose_check_launch();
}
// grid of bn_apply_k: every thread must keep its channel group over the grid-stride loop -> (grid * TPB) % (C/8) == 0
static int bn_apply_grid(size_t n8, int C) {
    const int G = C / 8;
    int g = grid_for(n8);
    if (g > 2048) g = 2048;             // 8 work-groups per CU: threads of the large tensors loop, coefficients stay in registers
    if ((TPB % G) != 0) {               // C/8 not a divisor of the block size (e.g. C = 24): round the grid to a multiple of G
        g = (g / G) * G;
        if (g < G) g = G;
    }
    return g;
}
int pw_bn_apply(hipStream_t s, const elem_t* y, const elem_t* res, elem_t* z, size_t n, int C, const float* scale, const float* shift, int relu,
                unsigned char* mask, int xcd) {
    if (C % 8 || n % 8) return UDAPOSE_ERR_ARG;
    const int grid = bn_apply_grid(n / 8, C), G = C / 8;
    // (XCD-aligned rows: needs whole rows per block pass, a grid that is a multiple of the 8 XCDs and enough rows to give every XCD work)
    const int ok = xcd && (TPB % G) == 0 && (grid % 8) == 0 && (n / 8 / G) >= (size_t)8 * (TPB / G);
    hipLaunchKernelGGL(bn_apply_k<elem_t>, dim3(grid), dim3(TPB), 0, s, y, res, z, n / 8, C, scale, shift, relu, mask, ok);
    return udapose_check_launch();
}
// f16x2 mode: y fp32 (the conv epilogue's fp32 output), residual and z split
int pw_bn_apply_split(hipStream_t s, const float* y, const void* res, void* z, size_t n, int C, const float* scale, const float* shift, int relu) {
    if (C % 8 || n % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL((bn_apply_k<sp32, float>), dim3(bn_apply_grid(n / 8, C)), dim3(TPB), 0, s, y, (const sp32*)res, (sp32*)z, n / 8, C, scale, shift, relu,
                       (unsigned char*)nullptr);
    return udapose_check_launch();
}
int pw_bn_apply_f32(hipStream_t s, const float* y, const float* res, float* z, size_t n, int C, const float* scale, const float* shift, int relu) {
    if (C % 8 || n % 8) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(bn_apply_k<float>, dim3(bn_apply_grid(n / 8, C)), dim3(TPB), 0, s, y, res, z, n / 8, C, scale, shift, relu, (unsigned char*)nullptr);
    return udapose_check_launch();
}
int pw_bn_bwd_rows(size_t npix) {
    // upper bound used to size the scratch slab; the launch picks rows = min(1024, ceil(npix / pixels-per-iteration))
    return (int)(npix < 1024 ? (npix < 1 ? 1 : npix) : 1024);
}
int pw_bn_bwd(hipStream_t s, const void* dz, int dz_is_f32, const elem_t* z, const elem_t* y, elem_t* dy, elem_t* gout, size_t npix, int C,
              const float* gamma, const float* mean, const float* invstd, int relu, float* slab, float* coef, float* dgamma, float* dbeta,
              float beta_acc, const float* beta, int chunked) {
    // chunked (Policy::bn_bwd_chunked): 0 restores reduce / finalize / apply everywhere
    if (relu == 2 && !beta) return UDAPOSE_ERR_ARG;
    if (relu == 1 && !z) return UDAPOSE_ERR_ARG;
    const int G = C / 8;
    if (C % 8 || G > 256 || (G & (G - 1))) return UDAPOSE_ERR_UNSUPPORTED;
    if (chunked && C >= 256 && npix <= 32768 && npix >= 1024) {
        // channel-chunked form without a finalize launch (see bn_bwd_reduce_chunk_k)
        const int chunks = C / 64;
        int S = 1024 / chunks;
        if (S > 64) S = 64;
        if (S < 1) S = 1;
        int P = (int)((npix + S - 1) / S);
        P = (P + 31) & ~31;
        S = (int)((npix + P - 1) / P);
        const dim3 grid(chunks, S);
        if (dz_is_f32) {
            hipLaunchKernelGGL(bn_bwd_reduce_chunk_k<float>, grid, dim3(TPB), 0, s, (const float*)dz, z, y, npix, C, mean, invstd, relu, slab, P, gamma, beta);
            hipLaunchKernelGGL(bn_bwd_apply_chunk_k<float>, grid, dim3(TPB), 0, s, (const float*)dz, z, y, dy, gout, npix, C, mean, invstd, relu, gamma,
                               beta, slab, P, dgamma, dbeta, beta_acc);
        } else {
            hipLaunchKernelGGL(bn_bwd_reduce_chunk_k<elem_t>, grid, dim3(TPB), 0, s, (const elem_t*)dz, z, y, npix, C, mean, invstd, relu, slab, P, gamma, beta);
            hipLaunchKernelGGL(bn_bwd_apply_chunk_k<elem_t>, grid, dim3(TPB), 0, s, (const elem_t*)dz, z, y, dy, gout, npix, C, mean, invstd, relu, gamma,
                               beta, slab, P, dgamma, dbeta, beta_acc);
        }
        return udapose_check_launch();
    }
    const int pstep = TPB / G;                                  // pixels a block covers per iteration
    size_t want = (npix + pstep - 1) / pstep;
    if (want > 1024) want = 1024;
    const int rows = (int)(want < 1 ? 1 : want);
    const int ppb = (int)((npix + rows - 1) / rows);
    if (dz_is_f32)
        hipLaunchKernelGGL(bn_bwd_reduce_k<float>, dim3(rows), dim3(TPB), 0, s, (const float*)dz, z, y, npix, C, mean, invstd, relu, slab, ppb, gamma, beta);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_k<elem_t>, dim3(rows), dim3(TPB), 0, s, (const elem_t*)dz, z, y, npix, C, mean, invstd, relu, slab, ppb, gamma, beta);
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((C + FIN_C - 1) / FIN_C), dim3(FIN_T), 0, s, slab, rows, C, (double)npix, gamma, invstd, dgamma, dbeta, beta_acc, coef);
    if (dz_is_f32)
        hipLaunchKernelGGL(bn_bwd_apply_k<float>, dim3(grid_for(npix * G)), dim3(TPB), 0, s, (const float*)dz, z, y, dy, gout, npix * G, C, mean, invstd,
                           coef, relu, gamma, beta);
    else
        hipLaunchKernelGGL(bn_bwd_apply_k<elem_t>, dim3(grid_for(npix * G)), dim3(TPB), 0, s, (const elem_t*)dz, z, y, dy, gout, npix * G, C, mean,
                           invstd, coef, relu, gamma, beta);
    return udapose_check_launch();
}
// BN backward after a dgrad that already masked dz and reduced it (DgradBnStat): slab[rows][2][C] -> dgamma / dbeta and
// dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)).  One launch for the wide, small-spatial layers, finalize + apply otherwise.
int pw_bn_bwd_pre(hipStream_t s, const void* g, int g_is_f32, const elem_t* y, elem_t* dy, size_t npix, int C, const float* gamma, const float* mean,
                  const float* invstd, const float* slab, int rows, float* coef, float* dgamma, float* dbeta, float beta_acc, int chunked,
                  int legacy) {
    const int G = C / 8;
    if (C % 8 || G > 256 || (G & (G - 1)) || rows < 1) return UDAPOSE_ERR_UNSUPPORTED;
    const int xcd = (chunked >> 30) & 1, xcd_stream = (chunked >> 29) & 1;     // (bit 30: XCD-aligned pixel ranges in the chunked form; bit 29: in the streaming form too)
    chunked &= ~(3 << 29);
    const int skip_finalize = 0;
    if (chunked && C >= 256 && npix <= 32768 && npix >= 1024 && rows <= 128) {
        const int chunks = C / 64;
        int S = (chunked > 1 ? chunked : 1024) / chunks;
        if (S > 64) S = 64;
        if (S < 1) S = 1;
        int P = (int)((npix + S - 1) / S);
        P = (P + 31) & ~31;
        S = (int)((npix + P - 1) / P);
        const dim3 grid(chunks, S);
        if (g_is_f32)
            hipLaunchKernelGGL(bn_bwd_apply_pre_chunk_k<float>, grid, dim3(TPB), 0, s, (const float*)g, y, dy, npix, C, mean, invstd, gamma, slab, rows, P,
                               dgamma, dbeta, beta_acc, xcd);
        else
            hipLaunchKernelGGL(bn_bwd_apply_pre_chunk_k<elem_t>, grid, dim3(TPB), 0, s, (const elem_t*)g, y, dy, npix, C, mean, invstd, gamma, slab, rows, P,
                               dgamma, dbeta, beta_acc, xcd);
        return udapose_check_launch();
    }
    if (!skip_finalize)
        hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((C + FIN_C - 1) / FIN_C), dim3(FIN_T), 0, s, slab, rows, C, (double)npix, gamma, invstd, dgamma, dbeta, beta_acc, coef);
    if (legacy) {
        if (g_is_f32)
            hipLaunchKernelGGL(bn_bwd_apply_k<float>, dim3(grid_for(npix * G)), dim3(TPB), 0, s, (const float*)g, (const elem_t*)nullptr, y, dy, (elem_t*)nullptr,
                               npix * G, C, mean, invstd, coef, 0, gamma, (const float*)nullptr);
        else
            hipLaunchKernelGGL(bn_bwd_apply_k<elem_t>, dim3(grid_for(npix * G)), dim3(TPB), 0, s, (const elem_t*)g, (const elem_t*)nullptr, y, dy,
                               (elem_t*)nullptr, npix * G, C, mean, invstd, coef, 0, gamma, (const float*)nullptr);
        return udapose_check_launch();
    }
    const int grid = bn_apply_grid(npix * G, C);
    const int ok = xcd_stream && (TPB % G) == 0 && (grid % 8) == 0 && npix >= (size_t)8 * (TPB / G);
    if (g_is_f32)
        hipLaunchKernelGGL(bn_bwd_apply_pre_k<float>, dim3(grid), dim3(TPB), 0, s, (const float*)g, y, dy, npix * G, C, mean, invstd, coef, ok);
    else
        hipLaunchKernelGGL(bn_bwd_apply_pre_k<elem_t>, dim3(grid), dim3(TPB), 0, s, (const elem_t*)g, y, dy, npix * G, C, mean, invstd, coef, ok);
    return udapose_check_launch();
}
int pw_maxpool3x3s2_fwd(hipStream_t s, const elem_t* x, elem_t* y, unsigned char* idx, int N, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_fwd_k<elem_t>, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(TPB), 0, s, x, y, idx, N, H, W, C, Ho, Wo);
    return udapose_check_launch();
}
int pw_bn_relu_maxpool3x3s2(hipStream_t s, const elem_t* x, elem_t* y, unsigned char* idx, int N, int H, int W, int C, const float* scale,
                            const float* shift) {
    if (C % 8 || !idx) return UDAPOSE_ERR_ARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(bn_relu_maxpool3x3s2_k, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(TPB), 0, s, x, y, idx, N, H, W, C, Ho, Wo, scale, shift);
    return udapose_check_launch();
}
// BN backward (relu mask recomputed from y) whose incoming gradient is the max-pool backward of (pool_dy, pool_idx), gathered on the
// fly: reduce -> finalize -> apply; H x W = the pool's input size, npix = N*H*W
int pw_bn_bwd_pooled(hipStream_t s, const elem_t* pool_dy, const unsigned char* pool_idx, int H, int W, const elem_t* y, elem_t* dy, size_t npix, int C,
                     const float* gamma, const float* mean, const float* invstd, float* slab, float* coef, float* dgamma, float* dbeta, float beta_acc,
                     const float* beta) {
    const int G = C / 8;
    if (C % 8 || G > 256 || (G & (G - 1)) || !beta || npix % ((size_t)H * W)) return UDAPOSE_ERR_UNSUPPORTED;
    const PoolSrc ps{pool_dy, pool_idx, H, W};
    const int pstep = TPB / G;
    size_t want = (npix + pstep - 1) / pstep;
    if (want > 1024) want = 1024;
    const int rows = (int)(want < 1 ? 1 : want);
    const int ppb = (int)((npix + rows - 1) / rows);
    hipLaunchKernelGGL((bn_bwd_reduce_k<elem_t, true>), dim3(rows), dim3(TPB), 0, s, (const elem_t*)nullptr, (const elem_t*)nullptr, y, npix, C, mean, invstd,
                       2, slab, ppb, gamma, beta, ps);
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((C + FIN_C - 1) / FIN_C), dim3(FIN_T), 0, s, slab, rows, C, (double)npix, gamma, invstd, dgamma, dbeta, beta_acc, coef);
    hipLaunchKernelGGL((bn_bwd_apply_k<elem_t, true>), dim3(grid_for(npix * G)), dim3(TPB), 0, s, (const elem_t*)nullptr, (const elem_t*)nullptr, y, dy,
                       (elem_t*)nullptr, npix * G, C, mean, invstd, coef, 2, gamma, beta, ps);
    return udapose_check_launch();
}
int pw_maxpool3x3s2_fwd_f32(hipStream_t s, const float* x, float* y, unsigned char* idx, int N, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_fwd_k<float>, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(TPB), 0, s, x, y, idx, N, H, W, C, Ho, Wo);
    return udapose_check_launch();
}
int pw_maxpool3x3s2_fwd_split(hipStream_t s, const void* x, void* y, unsigned char* idx, int N, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_fwd_k<sp32>, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(TPB), 0, s, (const sp32*)x, (sp32*)y, idx, N, H, W, C, Ho, Wo);
    return udapose_check_launch();
}
int pw_maxpool2x2_ceil_split(hipStream_t s, const void* x, void* y, int N, int H, int W, int C) {
    if (C % 8) return UDAPOSE_ERR_ARG;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(maxpool2x2_ceil_k<sp32>, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(TPB), 0, s, (const sp32*)x, (sp32*)y, N, H, W, C, Ho, Wo);
    return udapose_check_launch();
}
int pw_maxpool3x3s2_bwd(hipStream_t s, const elem_t* dy, const unsigned char* idx, elem_t* dx, int N, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_k, dim3(grid_for((size_t)N * H * W * (C / 8))), dim3(TPB), 0, s, dy, idx, dx, N, H, W, C, Ho, Wo);
    return udapose_check_launch();
}
int pw_maxpool2x2_ceil(hipStream_t s, const elem_t* x, elem_t* y, int N, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(maxpool2x2_ceil_k<elem_t>, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(TPB), 0, s, x, y, N, H, W, C, Ho, Wo);
    return udapose_check_launch();
}
int pw_maxpool2x2_ceil_f32(hipStream_t s, const float* x, float* y, int N, int H, int W, int C) {
    if (C % 8) return UDAPOSE_ERR_ARG;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(maxpool2x2_ceil_k<float>, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(TPB), 0, s, x, y, N, H, W, C, Ho, Wo);
    return udapose_check_launch();
}
int pw_plane_sum(hipStream_t s, const float* x, float* out, int N, int C, int HW, float beta) {
    hipLaunchKernelGGL(plane_sum_k, dim3(C), dim3(1024), 0, s, x, out, N, C, HW, beta);
    return udapose_check_launch();
}

UDAPOSE_SP_SAT_READER(sp_sat_read_pointwise)
