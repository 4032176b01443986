// Heat-map kernels (fp32 NCHW [B,K,H*W] rows): JointsMSE / consistency losses (fwd+bwd), arg-max decode, Gaussian
// "rectify" stamping, k-th value confidence mask, PCK.  Pure HBM sweeps: one pass over each operand, one block per
// (b,k) row where a per-row result is needed.  Deterministic (no atomics).
#include "common.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < TPB / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}

// rows[r] = 0.5 * w[r] * sum((p-g)^2) / HW           (JointsMSE 'none' rows; mean of rows = 'mean' loss)
// rows[r] = m[r] * sum((s-t)^2) / HW                  (ConsLoss: mode 1)
// `valid` (optional, ConsLoss(valid_mask=), loss.py:129-130): per-PIXEL selection [B][HW] shared by the Kc rows of an image;
// unselected pixels contribute nothing and the caller divides by the selected count instead of B*HW.
__global__ void sqdiff_rows_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                              const unsigned char* __restrict__ mask, int HW, float half, float* __restrict__ rows,
                              const unsigned char* __restrict__ valid, int Kc) {
    __shared__ double red[TPB / 64];
    const size_t r = blockIdx.x;
    const float* pa = a + r * HW;
    const float* pb = b + r * HW;
    double s = 0.0;
    if (valid) {
        const unsigned char* pv = valid + (r / Kc) * HW;
        for (int i = threadIdx.x; i < HW; i += TPB) { const float d = pv[i] ? pa[i] - pb[i] : 0.f; s += (double)(d * d); }
    } else if ((HW & 3) == 0) {
        for (int i = threadIdx.x * 4; i < HW; i += TPB * 4) {
            const f32x4 x = *(const f32x4*)(pa + i), y = *(const f32x4*)(pb + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = x[e] - y[e]; s += (double)(d * d); }
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += TPB) { const float d = pa[i] - pb[i]; s += (double)(d * d); }
    }
    const double t = block_sum_d(s, red);
    if (threadIdx.x == 0) {
        float f = half;
        if (w) f *= w[r];
        if (mask) f *= mask[r] ? 1.f : 0.f;
        rows[r] = (float)(t / HW) * f;
    }
}
// out = sum(rows) / R, or - with a pixel selection - sum(rows) * HW / (Kc * count): rows carry sum/HW, the mean runs over
// the `count` selected (b, h, w) positions of loss_map = mean over the Kc channels
__global__ void mean_rows_k(const float* __restrict__ rows, int R, float* __restrict__ out, const float* __restrict__ count, int HW, int Kc) {
    __shared__ double red[TPB / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < R; i += TPB) s += (double)rows[i];
    const double t = block_sum_d(s, red);
    if (threadIdx.x == 0) out[0] = count ? (float)(t * HW / ((double)Kc * (double)count[0])) : (float)(t / R);
}
// count[0] = number of non-zero bytes of m[0..n)
__global__ void mask_count_k(const unsigned char* __restrict__ m, size_t n, float* __restrict__ count) {
    __shared__ double red[TPB / 64];
    double s = 0.0;
    for (size_t i = threadIdx.x; i < n; i += TPB) s += m[i] ? 1.0 : 0.0;
    const double t = block_sum_d(s, red);
    if (threadIdx.x == 0) count[0] = (float)t;
}
// da = gscale[0] * coef * f[r] * (a-b), f = w or mask (or 1)
__global__ void sqdiff_bwd_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                             const unsigned char* __restrict__ mask, const float* __restrict__ gscale, float coef, int HW,
                             size_t total, float* __restrict__ da, const unsigned char* __restrict__ valid, const float* __restrict__ count, int Kc) {
    float gs = gscale ? gscale[0] : 1.f;
    if (count) gs *= (float)(((double)(total / HW) / Kc) * HW / (double)count[0]);     // B*HW / count: the mean runs over the selected pixels
    for (size_t i = ((size_t)blockIdx.x * TPB + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * TPB * 4) {
        const size_t r = i / HW;     // HW % 4 == 0 enforced by the launcher
        float f = gs * coef;
        if (w) f *= w[r];
        if (mask) f *= mask[r] ? 1.f : 0.f;
        const f32x4 x = *(const f32x4*)(a + i), y = *(const f32x4*)(b + i);
        f32x4 o = (f32x4){f * (x[0] - y[0]), f * (x[1] - y[1]), f * (x[2] - y[2]), f * (x[3] - y[3])};
        if (valid) {
            const unsigned char* pv = valid + (r / Kc) * HW + (i - r * HW);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!pv[e]) o[e] = 0.f;
        }
        *(f32x4*)(da + i) = o;
    }
}

// the same element by element, for maps whose pixel count is not a multiple of 4 (rows are then not 16-byte aligned)
__global__ void sqdiff_bwd_scalar_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                                    const unsigned char* __restrict__ mask, const float* __restrict__ gscale, float coef, int HW,
                                    size_t total, float* __restrict__ da, const unsigned char* __restrict__ valid, const float* __restrict__ count, int Kc) {
    float gs = gscale ? gscale[0] : 1.f;
    if (count) gs *= (float)(((double)(total / HW) / Kc) * HW / (double)count[0]);
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const size_t r = i / HW;
        float f = gs * coef;
        if (w) f *= w[r];
        if (mask) f *= mask[r] ? 1.f : 0.f;
        float o = f * (a[i] - b[i]);
        if (valid && !valid[(r / Kc) * HW + (i - r * HW)]) o = 0.f;
        da[i] = o;
    }
}

// torch's total order on floats for max / kthvalue: NaN is the LARGEST value (and all NaNs are equal)
__device__ __forceinline__ bool hm_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    if (vn) return i < bi;
    return v > bv || (v == bv && i < bi);
}
// order-preserving float -> uint32 key (NaN -> 0xffffffff, the largest: torch.kthvalue's order)
__device__ __forceinline__ unsigned int hm_key(float v) {
    if (v != v) return 0xffffffffu;
    const unsigned int b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float hm_unkey(unsigned int k) {
    if (k == 0xffffffffu) return __uint_as_float(0x7fc00000u);
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// per row: max value, first flat arg-max, (x,y) zeroed when max <= 0  (utils.py:54-75 / keypoint_detection.py:9-37);
// optionally stamps the rectify patch (utils.py:77-109) into out (full row written: zeros elsewhere).
__global__ void argmax_rectify_k(const float* __restrict__ hm, int H, int W, float* __restrict__ maxv, int* __restrict__ idx_out,
                                 float* __restrict__ preds, float* __restrict__ rect, const float* __restrict__ patch, int rad) {
    __shared__ float sv[TPB / 64];
    __shared__ int si[TPB / 64];
    const size_t r = blockIdx.x;
    const int HW = H * W;
    const float* p = hm + r * HW;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < HW; i += TPB) {
        const float v = p[i];
        if (hm_better(v, i, bv, bi)) { bv = v; bi = i; }      // (NaN counts as the maximum, first index on ties: torch / numpy)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (hm_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    bv = sv[0]; bi = si[0];
    for (int k = 1; k < TPB / 64; ++k)
        if (hm_better(sv[k], si[k], bv, bi)) { bv = sv[k]; bi = si[k]; }
    const bool pos = bv > 0.f;
    const int mx = pos ? bi % W : 0, my = pos ? bi / W : 0;
    if (threadIdx.x == 0) {
        if (maxv) maxv[r] = bv;
        if (idx_out) idx_out[r] = bi;
        if (preds) { preds[r * 2] = (float)mx; preds[r * 2 + 1] = (float)my; }
    }
    if (rect) {
        // reference quirk kept: bounds test compares x with H and y with W (utils.py:89)
        const bool skip = (mx >= H) || (my >= W);
        const int size = 2 * rad + 1;
        float* o = rect + r * HW;
        for (int i = threadIdx.x; i < HW; i += TPB) {
            const int y = i / W, x = i % W;
            const int gx = x - (mx - rad), gy = y - (my - rad);
            float v = 0.f;
            // image range: x in [max(0,ulx), min(brx,H)), y in [max(0,uly), min(bry,W))  (reference's h/w usage)
            if (!skip && gx >= 0 && gx < size && gy >= 0 && gy < size && x < H && y < W) v = patch[gy * size + gx];
            o[i] = v;
        }
    }
}

// k-th smallest (1-indexed) of n values, then mask[i] = (tm[i]*act[i]) > thr  (train_human.py:429-430).
// MSB-first radix select in ONE work-group: 4 passes of 8 bits, each a 256-bin LDS histogram of the keys that still match
// the prefix found so far, then a scan of the bins for the one that holds rank k.  O(n) per pass (the previous rank-count
// form was O(n^2): 23 us at n = 512, ~0.7 ms at the n = 4096 of an 8-rank global batch); n = 4096 costs 16 key loads per
// thread in all.  Values are read from L2 each pass (n <= 65536 floats = 256 KiB).  NaN orders as the largest value and a
// NaN threshold gives an all-false mask, exactly like torch.kthvalue + `>`.
__global__ void kth_mask_k(const float* __restrict__ act, const float* __restrict__ tm, int n, int k, float* __restrict__ thr_out,
                           unsigned char* __restrict__ mask, const float* __restrict__ act_local, int n_local) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sel_bin, sel_rank;
    unsigned int prefix = 0, pmask = 0, rank = (unsigned int)(k - 1);     // 0-based rank inside the surviving set
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned int key = hm_key(act[i]);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // wave 0: each lane owns 4 consecutive bins; exclusive prefix over lanes, then the owner of the rank reports
            const int b0 = threadIdx.x * 4;
            const unsigned int c0 = hist[b0], c1 = hist[b0 + 1], c2 = hist[b0 + 2], c3 = hist[b0 + 3];
            const unsigned int tot = c0 + c1 + c2 + c3;
            unsigned int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int up = __shfl_up(incl, o, 64);
                if ((int)threadIdx.x >= o) incl += up;
            }
            const unsigned int excl = incl - tot;
            if (rank >= excl && rank < incl) {
                unsigned int r = rank - excl;
                int b = b0;
                if (r >= c0) { r -= c0; ++b; if (r >= c1) { r -= c1; ++b; if (r >= c2) { r -= c2; ++b; } } }
                sel_bin = (unsigned int)b;
                sel_rank = r;
            }
        }
        __syncthreads();
        prefix |= sel_bin << shift;
        pmask |= 255u << shift;
        rank = sel_rank;
        __syncthreads();
    }
    const float thr = hm_unkey(prefix);
    if (threadIdx.x == 0 && thr_out) thr_out[0] = thr;
    const float* al = act_local ? act_local : act;
    const int nl = act_local ? n_local : n;
    for (int i = threadIdx.x; i < nl; i += blockDim.x) mask[i] = ((tm ? tm[i] : 1.f) * al[i]) > thr ? 1 : 0;
}

// PCK from decoded coordinates (keypoint_detection.py:40-94): per key-point hits / counts over the batch.
__global__ void pck_k(const float* __restrict__ pred, const float* __restrict__ gt, int B, int K, float nh, float nw, float thr,
                      float* __restrict__ acc, float* __restrict__ avg_cnt) {
    // single block; thread per key-point
    __shared__ float s_acc[256];
    const int c = threadIdx.x;
    float a = -1.f;
    if (c < K) {
        int hits = 0, n = 0;
        for (int b = 0; b < B; ++b) {
            const float tx = gt[(b * K + c) * 2], ty = gt[(b * K + c) * 2 + 1];
            if (tx > 1.f && ty > 1.f) {
                const float dx = pred[(b * K + c) * 2] / nh - tx / nh, dy = pred[(b * K + c) * 2 + 1] / nw - ty / nw;
                ++n;
                hits += sqrtf(dx * dx + dy * dy) < thr;
            }
        }
        a = n ? (float)hits / n : -1.f;
        acc[c] = a;
    }
    s_acc[c] = a;
    __syncthreads();
    if (c == 0) {
        float t = 0.f; int cnt = 0;
        for (int i = 0; i < K; ++i) if (s_acc[i] >= 0.f) { t += s_acc[i]; ++cnt; }
        avg_cnt[0] = cnt ? t / cnt : 0.f;
        avg_cnt[1] = (float)cnt;
    }
}
}  // namespace

// valid / count / Kc: optional per-pixel selection [R/Kc][HW] of ConsLoss(valid_mask=) and its device-resident count (hm_mask_count)
int hm_sqdiff_rows(hipStream_t s, const float* a, const float* b, const float* w, const unsigned char* mask, int R, int HW, float half,
                   float* rows, float* mean_out, const unsigned char* valid, const float* count, int Kc) {
    if (valid && (!count || Kc < 1 || R % Kc)) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(sqdiff_rows_k, dim3(R), dim3(TPB), 0, s, a, b, w, mask, HW, half, rows, valid, Kc);
    if (mean_out) hipLaunchKernelGGL(mean_rows_k, dim3(1), dim3(TPB), 0, s, rows, R, mean_out, valid ? count : nullptr, HW, Kc);
    return udapose_check_launch();
}
int hm_sqdiff_bwd(hipStream_t s, const float* a, const float* b, const float* w, const unsigned char* mask, const float* gscale, float coef,
                  int R, int HW, float* da, const unsigned char* valid, const float* count, int Kc) {
    if (valid && (!count || Kc < 1 || R % Kc)) return UDAPOSE_ERR_ARG;
    const size_t total = (size_t)R * HW;
    if (HW % 4) {
        size_t blocks = (total + TPB - 1) / TPB;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(sqdiff_bwd_scalar_k, dim3((int)blocks), dim3(TPB), 0, s, a, b, w, mask, gscale, coef, HW, total, da, valid,
                           valid ? count : nullptr, Kc);
        return udapose_check_launch();
    }
    size_t blocks = (total / 4 + TPB - 1) / TPB;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sqdiff_bwd_k, dim3((int)blocks), dim3(TPB), 0, s, a, b, w, mask, gscale, coef, HW, total, da, valid, valid ? count : nullptr,
                       Kc);
    return udapose_check_launch();
}
int hm_mask_count(hipStream_t s, const unsigned char* m, size_t n, float* count) {
    hipLaunchKernelGGL(mask_count_k, dim3(1), dim3(TPB), 0, s, m, n, count);
    return udapose_check_launch();
}
int hm_argmax_rectify(hipStream_t s, const float* hm, int R, int H, int W, float* maxv, int* idx, float* preds, float* rect, const float* patch,
                      int rad) {
    hipLaunchKernelGGL(argmax_rectify_k, dim3(R), dim3(TPB), 0, s, hm, H, W, maxv, idx, preds, rect, patch, rad);
    return udapose_check_launch();
}
int hm_kth_mask(hipStream_t s, const float* act, const float* tm, int n, int k, float* thr_out, unsigned char* mask, const float* act_local,
                int n_local) {
    if (k < 1 || k > n || n > (1 << 22)) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(kth_mask_k, dim3(1), dim3(1024), 0, s, act, tm, n, k, thr_out, mask, act_local, n_local);
    return udapose_check_launch();
}
int hm_pck(hipStream_t s, const float* pred, const float* gt, int B, int K, float nh, float nw, float thr, float* acc, float* avg_cnt) {
    if (K > 256) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(pck_k, dim3(1), dim3(256), 0, s, pred, gt, B, K, nh, nw, thr, acc, avg_cnt);
    return udapose_check_launch();
}
