// Device-side data pipeline of the mean-teacher target views: what the reference's DataLoader workers
// do per sample with PIL / torchvision (lib/transforms/keypoint_detection.py:137-167,365-453 RandomAffineRotation -> F.affine
// on a PIL image, ColorJitter = PIL.ImageEnhance brightness / contrast / saturation in a random order, ToTensor, Normalize;
// lib/datasets/util.py:12-70 generate_target) for a whole batch on the GPU, bit-exact with PIL's own integer arithmetic:
//   * Image.transform(AFFINE, NEAREST) steps 16.16 fixed-point source coordinates (libImaging affine_fixed): reproduced with
//     the same FIX()ed coefficients (computed on the host in double, like PIL) and arithmetic shifts;
//   * ImageEnhance = Image.blend(degenerate, image, factor): float32  d + f * (v - d), clipped to [0, 255], TRUNCATED to uint8;
//     degenerate = black (brightness), the rounded mean of the L image (contrast), the L image (saturation);
//     L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16.
// All kernels are byte / float sweeps (HBM-bound, 196 KB per 256x256 image); images are uint8 NHWC as PIL arrays are.
#include "common.h"

// no FMA contraction: PIL's blend rounds the product and the sum separately (d + f * (v - d) in C float); a fused
// multiply-add lands on the other side of an integer for ~1 % of the bytes and the truncation to uint8 then differs
#pragma clang fp contract(off)

namespace {
constexpr int TPB = 256;

// dst[n][y][x][:] = src[n][yin][xin][:] with (xin, yin) = ((a2 + x*a0 + y*a1) >> 16, (a5 + x*a3 + y*a4) >> 16), zero outside
__global__ void aug_affine_u8_k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const long long* __restrict__ coef, int H, int W) {
    const int n = blockIdx.y;
    const long long* c = coef + (size_t)n * 6;
    const long long a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3], a4 = c[4], a5 = c[5];
    const unsigned char* s = src + (size_t)n * H * W * 3;
    unsigned char* d = dst + (size_t)n * H * W * 3;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < H * W; i += gridDim.x * TPB) {
        const int y = i / W, x = i - y * W;
        const long long xin = (a2 + (long long)x * a0 + (long long)y * a1) >> 16;
        const long long yin = (a5 + (long long)x * a3 + (long long)y * a4) >> 16;
        unsigned char r = 0, g = 0, b = 0;
        if (xin >= 0 && xin < W && yin >= 0 && yin < H) {
            const unsigned char* p = s + ((size_t)yin * W + xin) * 3;
            r = p[0]; g = p[1]; b = p[2];
        }
        d[(size_t)i * 3] = r; d[(size_t)i * 3 + 1] = g; d[(size_t)i * 3 + 2] = b;
    }
}

__device__ __forceinline__ int lum(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// mean[n] = int(sum(L) / HW + 0.5) of image n (ImageStat.Stat(img.convert("L")).mean[0] rounded as ImageEnhance.Contrast does);
// computed only where this launch's op for the image is "contrast" (op[n] == 2): one block per image
__global__ void aug_gray_mean_k(const unsigned char* __restrict__ img, const int* __restrict__ op, int HW, int* __restrict__ mean) {
    __shared__ unsigned long long red[TPB / 64];
    const int n = blockIdx.x;
    if (op[n] != 2) return;
    const unsigned char* p = img + (size_t)n * HW * 3;
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < HW; i += TPB) s += (unsigned long long)lum(p[(size_t)i * 3], p[(size_t)i * 3 + 1], p[(size_t)i * 3 + 2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < TPB / 64; ++i) t += red[i];
        // int(t / HW + 0.5) with t / HW in double exactly as ImageStat computes it (sum / count), then truncation
        mean[n] = (int)((double)t / (double)HW + 0.5);
    }
}

__device__ __forceinline__ unsigned char blend1(float d, float v, float f) {
    float prod = f * (v - d);
    asm volatile("" : "+v"(prod));        // (belt and braces: the product is pinned in a register, no contraction can cross this)
    const float t = d + prod;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)t);
}

// one ImageEnhance step per image, in place: op 0 none, 1 brightness, 2 contrast, 3 saturation (Color)
__global__ void aug_color_op_k(unsigned char* __restrict__ img, const int* __restrict__ op, const float* __restrict__ factor,
                               const int* __restrict__ mean, int HW) {
    const int n = blockIdx.y, o = op[n];
    if (o == 0) return;
    const float f = factor[n];
    const float dm = o == 2 ? (float)mean[n] : 0.f;
    unsigned char* p = img + (size_t)n * HW * 3;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
        const int r = p[(size_t)i * 3], g = p[(size_t)i * 3 + 1], b = p[(size_t)i * 3 + 2];
        const float d = o == 3 ? (float)lum(r, g, b) : dm;
        p[(size_t)i * 3] = blend1(d, (float)r, f);
        p[(size_t)i * 3 + 1] = blend1(d, (float)g, f);
        p[(size_t)i * 3 + 2] = blend1(d, (float)b, f);
    }
}

// ToTensor + Normalize: out[n][c][i] = (img[n][i][c] / 255 - mean[c]) / std[c], float32 operations in torch's order
__global__ void aug_to_tensor_k(const unsigned char* __restrict__ img, float* __restrict__ out, int HW, const float* __restrict__ mean3,
                                const float* __restrict__ std3) {
    const int n = blockIdx.y;
    const unsigned char* p = img + (size_t)n * HW * 3;
    float* o = out + (size_t)n * 3 * HW;
    const float m0 = mean3[0], m1 = mean3[1], m2 = mean3[2], s0 = std3[0], s1 = std3[1], s2 = std3[2];
    for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
        o[i] = ((float)p[(size_t)i * 3] / 255.f - m0) / s0;
        o[HW + i] = ((float)p[(size_t)i * 3 + 1] / 255.f - m1) / s1;
        o[2 * HW + i] = ((float)p[(size_t)i * 3 + 2] / 255.f - m2) / s2;
    }
}

// generate_target (lib/datasets/util.py:12-70): one block per (sample, joint) row; kp [R][2] double (x, y) in image pixels,
// vis [R]; centre mu = int(kp / stride + 0.5) (truncation, like Python's int()); weight = vis, 0 when the centre is outside the
// map; the (2*rad+1)^2 patch (built on the host exactly as the reference builds it) is copied where weight > 0.5.
__global__ void gaussian_labels_k(const double* __restrict__ kp, const float* __restrict__ vis, float* __restrict__ target, float* __restrict__ weight,
                                  int Hh, int Wh, double stride_x, double stride_y, const float* __restrict__ patch, int rad) {
    const size_t r = blockIdx.x;
    const int mx = (int)(kp[r * 2] / stride_x + 0.5), my = (int)(kp[r * 2 + 1] / stride_y + 0.5);
    float w = vis[r];
    const bool outside = mx >= Wh || my >= Hh || mx < 0 || my < 0;
    if (outside) w = 0.f;
    if (threadIdx.x == 0) weight[r] = w;
    const bool draw = !outside && w > 0.5f;
    const int size = 2 * rad + 1;
    float* t = target + r * (size_t)Hh * Wh;
    for (int i = threadIdx.x; i < Hh * Wh; i += TPB) {
        const int y = i / Wh, x = i - y * Wh;
        const int gx = x - (mx - rad), gy = y - (my - rad);
        t[i] = (draw && gx >= 0 && gx < size && gy >= 0 && gy < size) ? patch[gy * size + gx] : 0.f;
    }
}
// draw_labelmap_ori (lib/datasets/util.py:326-363), the label generator of the animal `_mt` datasets (call loop:
// lib/datasets/real_animal_all_mt.py:274-283, animal_pose_mt.py:169-177,200-205): row r's centre is pt[r] truncated to int32
// (`pt.to(torch.int32)`), the stamp's corners are int(centre -+ r3 (+ 1)) with the sums in float32 (an int32 tensor and a Python
// float), and the stamp is drawn - weight kept - only when ALL of it lies inside the map (any part outside: zeros, weight 0).
// gate[r] = the caller's `tpts[i, 1] > 0` test of the un-transformed key point: closed rows keep their visibility weight and an
// empty map.  patch = the reference's (6 sigma + 1)^2 float64 stamp (Gaussian or Cauchy) rounded to float32 by the caller.
__global__ void draw_labelmap_ori_k(const float* __restrict__ pt, const float* __restrict__ vis, const unsigned char* __restrict__ gate,
                                    float* __restrict__ target, float* __restrict__ weight, int Hh, int Wh, float r3,
                                    const float* __restrict__ patch, int psize) {
    const size_t r = blockIdx.x;
    const int cx = (int)pt[r * 2], cy = (int)pt[r * 2 + 1];
    const int ulx = (int)((float)cx - r3), uly = (int)((float)cy - r3);
    const int brx = (int)(((float)cx + r3) + 1.0f), bry = (int)(((float)cy + r3) + 1.0f);
    const bool inside = !(brx >= Wh || bry >= Hh || ulx < 0 || uly < 0);
    const bool open = gate[r] != 0;
    if (threadIdx.x == 0) weight[r] = open ? vis[r] * (inside ? 1.0f : 0.0f) : vis[r];
    const bool draw = open && inside;
    const int nx = min(brx - ulx, psize), ny = min(bry - uly, psize);
    float* t = target + r * (size_t)Hh * Wh;
    for (int i = threadIdx.x; i < Hh * Wh; i += TPB) {
        const int y = i / Wh, x = i - y * Wh;
        const int gx = x - ulx, gy = y - uly;
        t[i] = (draw && gx >= 0 && gx < nx && gy >= 0 && gy < ny) ? patch[gy * psize + gx] : 0.f;
    }
}
// PIL's ImageFilter.GaussianBlur (lib/transforms/keypoint_detection.py:216-225; libImaging BoxBlur.c): three passes of a box blur
// per direction, every pass in 8.24 fixed point with replicated edges:
//   out[x] = (sum_{k=-r..r} in[x+k] * ww + (in[x-r-1] + in[x+r+1]) * fw + 2^23) >> 24     (32-bit unsigned arithmetic)
// with r = int(box radius), ww = uint32(2^24 / (2 * box radius + 1)) in float32, fw = (2^24 - (2r + 1) * ww) / 2; the box radius
// follows from the Gaussian radius on the host (data_gpu.pil_box_blur_params restates `_gaussian_blur_radius`).  prm[n] = (r, ww, fw),
// r < 0: the sample is copied (radius 0: PIL returns a copy).  One launch = one pass over [N][H][W][3] uint8 along x or along y.
__global__ void box_blur_u8_k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const unsigned int* __restrict__ prm,
                              int H, int W, int vertical) {
    const int n = blockIdx.y;
    const int r = (int)prm[n * 3];
    const unsigned ww = prm[n * 3 + 1], fw = prm[n * 3 + 2];
    const unsigned char* si = src + (size_t)n * H * W * 3;
    unsigned char* di = dst + (size_t)n * H * W * 3;
    const int L = vertical ? H : W;                 // length of a blur line
    const int step = vertical ? W * 3 : 3;          // bytes between neighbours of a line
    for (int i = blockIdx.x * TPB + threadIdx.x; i < H * W; i += gridDim.x * TPB) {
        const int y = i / W, x = i - y * W;
        const unsigned char* px = si + (size_t)i * 3;
        if (r < 0) { di[(size_t)i * 3] = px[0]; di[(size_t)i * 3 + 1] = px[1]; di[(size_t)i * 3 + 2] = px[2]; continue; }
        const int pos = vertical ? y : x;
        const unsigned char* line = px - (size_t)pos * step;       // element 0 of this pixel's line
        unsigned a0 = 0, a1 = 0, a2 = 0;
        for (int k = -r; k <= r; ++k) {
            int q = pos + k;
            q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
            const unsigned char* t = line + (size_t)q * step;
            a0 += t[0]; a1 += t[1]; a2 += t[2];
        }
        int ql = pos - r - 1, qr = pos + r + 1;
        ql = ql < 0 ? 0 : (ql > L - 1 ? L - 1 : ql);
        qr = qr < 0 ? 0 : (qr > L - 1 ? L - 1 : qr);
        const unsigned char* tl = line + (size_t)ql * step;
        const unsigned char* tr = line + (size_t)qr * step;
        const unsigned b0 = a0 * ww + (unsigned)(tl[0] + tr[0]) * fw, b1 = a1 * ww + (unsigned)(tl[1] + tr[1]) * fw,
                       b2 = a2 * ww + (unsigned)(tl[2] + tr[2]) * fw;
        di[(size_t)i * 3] = (unsigned char)((b0 + (1u << 23)) >> 24);
        di[(size_t)i * 3 + 1] = (unsigned char)((b1 + (1u << 23)) >> 24);
        di[(size_t)i * 3 + 2] = (unsigned char)((b2 + (1u << 23)) >> 24);
    }
}

// PIL's Image.resize(BILINEAR) of a crop (T.RandomResizedCrop, lib/transforms/keypoint_detection.py:456-521 -> resized_crop :66-88 ->
// F.crop + F.resize; libImaging Resample.c): a separable convolution whose support grows with the reduction factor, in two passes -
// horizontal into a uint8 intermediate, then vertical - each  out = clip8((2^21 + sum_k in[min + k] * coef[k]) >> 22)  with the
// coefficients normalised in double and rounded to 22 fractional bits on the host (data_gpu.pil_resample_coeffs restates
// `precompute_coeffs` / `normalize_coeffs_8bpc`).  bounds[n][axis][o] = (first source index, tap count), coef[n][axis][o][ksize].
__device__ __forceinline__ unsigned char clip8_22(int v) {
    v >>= 22;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// pass 0: tmp[n][r][x] (r < box h, x < S) from src[n][top + r][left + ...];  pass 1: dst[n][y][x] (y, x < S) from tmp[n][...][x]
__global__ void resample_u8_k(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, const int* __restrict__ box,
                              const int* __restrict__ bounds, const int* __restrict__ coef, int in_h, int in_w, int out_hmax, int S, int ksize,
                              int vertical) {
    const int n = blockIdx.y;
    const int top = box[n * 4], left = box[n * 4 + 1], bh = box[n * 4 + 2];
    const int rows = vertical ? S : bh;
    const unsigned char* src = in + (size_t)n * in_h * in_w * 3;
    unsigned char* dst = out + (size_t)n * out_hmax * S * 3;
    const int* bnd = bounds + (size_t)(n * 2 + vertical) * S * 2;
    const int* kk = coef + (size_t)(n * 2 + vertical) * S * ksize;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < rows * S; i += gridDim.x * TPB) {
        const int y = i / S, x = i - y * S;
        const int o = vertical ? y : x;
        const int first = bnd[o * 2], cnt = bnd[o * 2 + 1];
        const int* k = kk + (size_t)o * ksize;
        // horizontal: walk along the source row (top + y) from column left + first; vertical: walk down column x of the intermediate
        const unsigned char* p = vertical ? src + ((size_t)first * in_w + x) * 3 : src + ((size_t)(top + y) * in_w + left + first) * 3;
        const size_t step = vertical ? (size_t)in_w * 3 : 3;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int t = 0; t < cnt; ++t, p += step) {
            const int c = k[t];
            s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
        }
        unsigned char* q = dst + (size_t)i * 3;
        q[0] = clip8_22(s0); q[1] = clip8_22(s1); q[2] = clip8_22(s2);
    }
}
}  // namespace

// src [N][Hs][Ws][3] uint8 -> dst [N][S][S][3] uint8: crop box[n] = (top, left, h, w) resized to S x S; tmp [N][Hs][S][3] uint8
int aug_resized_crop_u8(hipStream_t s, const unsigned char* src, unsigned char* dst, unsigned char* tmp, const int* box, const int* bounds,
                        const int* coef, int N, int Hs, int Ws, int S, int ksize) {
    if (N <= 0 || Hs <= 0 || Ws <= 0 || S <= 0 || ksize <= 0) return UDAPOSE_ERR_ARG;
    int gx = (Hs * S + TPB - 1) / TPB;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(resample_u8_k, dim3(gx, N), dim3(TPB), 0, s, src, tmp, box, bounds, coef, Hs, Ws, Hs, S, ksize, 0);
    gx = (S * S + TPB - 1) / TPB;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(resample_u8_k, dim3(gx, N), dim3(TPB), 0, s, (const unsigned char*)tmp, dst, box, bounds, coef, Hs, S, S, S, ksize, 1);
    return udapose_check_launch();
}

// img [N][H][W][3] uint8 is blurred IN PLACE through the scratch buffer tmp (same size): 3 passes along x, 3 along y
int aug_gaussian_blur_u8(hipStream_t s, unsigned char* img, unsigned char* tmp, const unsigned int* prm, int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return UDAPOSE_ERR_ARG;
    int gx = (H * W + TPB - 1) / TPB;
    if (gx > 256) gx = 256;
    unsigned char* a = img;
    unsigned char* b = tmp;
    for (int pass = 0; pass < 6; ++pass) {
        hipLaunchKernelGGL(box_blur_u8_k, dim3(gx, N), dim3(TPB), 0, s, a, b, prm, H, W, pass >= 3 ? 1 : 0);
        unsigned char* t = a; a = b; b = t;
    }
    return udapose_check_launch();      // (6 passes: the result is back in img)
}

int aug_affine_u8(hipStream_t s, const unsigned char* src, unsigned char* dst, const long long* coef, int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return UDAPOSE_ERR_ARG;
    int gx = (H * W + TPB - 1) / TPB;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(aug_affine_u8_k, dim3(gx, N), dim3(TPB), 0, s, src, dst, coef, H, W);
    return udapose_check_launch();
}
int aug_color_op(hipStream_t s, unsigned char* img, const int* op, const float* factor, int* mean_scratch, int N, int HW) {
    if (N <= 0 || HW <= 0) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(aug_gray_mean_k, dim3(N), dim3(TPB), 0, s, img, op, HW, mean_scratch);
    int gx = (HW + TPB - 1) / TPB;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(aug_color_op_k, dim3(gx, N), dim3(TPB), 0, s, img, op, factor, mean_scratch, HW);
    return udapose_check_launch();
}
int aug_to_tensor(hipStream_t s, const unsigned char* img, float* out, int N, int HW, const float* mean3, const float* std3) {
    if (N <= 0 || HW <= 0) return UDAPOSE_ERR_ARG;
    int gx = (HW + TPB - 1) / TPB;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(aug_to_tensor_k, dim3(gx, N), dim3(TPB), 0, s, img, out, HW, mean3, std3);
    return udapose_check_launch();
}
int aug_gaussian_labels(hipStream_t s, const double* kp, const float* vis, float* target, float* weight, int R, int Hh, int Wh, double stride_x,
                        double stride_y, const float* patch, int rad) {
    if (R <= 0 || Hh <= 0 || Wh <= 0 || rad < 0 || !patch) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(gaussian_labels_k, dim3(R), dim3(TPB), 0, s, kp, vis, target, weight, Hh, Wh, stride_x, stride_y, patch, rad);
    return udapose_check_launch();
}
int aug_draw_labelmap_ori(hipStream_t s, const float* pt, const float* vis, const unsigned char* gate, float* target, float* weight, int R, int Hh,
                          int Wh, float r3, const float* patch, int psize) {
    if (R <= 0 || Hh <= 0 || Wh <= 0 || psize <= 0 || !patch || !pt || !vis || !gate) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(draw_labelmap_ori_k, dim3(R), dim3(TPB), 0, s, pt, vis, gate, target, weight, Hh, Wh, r3, patch, psize);
    return udapose_check_launch();
}
