// Batched heat-map / image re-warp: the three sequential nearest-neighbour inverse-affine resamplings the reference
// applies per sample with torchvision.transforms.functional.affine (train_human.py:366-368, 421-423), as ONE gather
// kernel driven by a [N][3][6] matrix tensor.  Because every stage is a nearest resample, the composition is evaluated
// as a chain of index maps p3 -> p2 -> p1 -> p0 (NOT as one composed matrix: that would change results).
// Arithmetic follows torchvision/_gen_affine_grid + ATen grid_sampler(nearest, zeros, align_corners=False) in fp32:
// theta / (0.5*[W,H]); base grid at half-integers; ix = ((x+1)*W-1)/2; nearbyint.
#include "conv_plan.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ bool step(const float* __restrict__ th, int W, int H, int& px, int& py) {
    const float bx = (float)px - 0.5f * (float)W + 0.5f, by = (float)py - 0.5f * (float)H + 0.5f;
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const float gx = bx * (th[0] / hw) + by * (th[1] / hw) + th[2] / hw;
    const float gy = bx * (th[3] / hh) + by * (th[4] / hh) + th[5] / hh;
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    if (!(rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1))) return false;
    px = (int)rx; py = (int)ry;
    return true;
}

// nstage sequential warps (1..3); theta: [N][nstage][6] fp32 in application order (stage 0 applied first)
template <bool BWD>
__global__ void warp_chain_k(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ theta, int N, int C, int H, int W,
                             int nstage) {
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (size_t)gridDim.x * TPB) {
        const int n = (int)(i / ((size_t)H * W));
        const int rem = (int)(i % ((size_t)H * W));
        int py = rem / W, px = rem % W;
        bool ok = true;
        for (int s = nstage - 1; s >= 0 && ok; --s) ok = step(theta + ((size_t)n * nstage + s) * 6, W, H, px, py);
        const size_t o = (size_t)n * C * H * W;
        // (blockIdx.y strides the channels: the index map is cheap to recompute and the launch sits, alone on the device, between the
        // forwards and the backward of a step - 16 channel groups took the backward of [32,16,64,64] from 58 us to what one atomic costs)
        if (!BWD) {
            for (int c = blockIdx.y; c < C; c += gridDim.y) dst[o + (size_t)c * H * W + rem] = ok ? src[o + (size_t)c * H * W + py * W + px] : 0.f;
        } else if (ok) {
            // src = d(out), dst = d(in) (pre-zeroed): several outputs may read the same input pixel
            for (int c = blockIdx.y; c < C; c += gridDim.y) atomicAdd(dst + o + (size_t)c * H * W + py * W + px, src[o + (size_t)c * H * W + rem]);
        }
    }
}
// Backward of the chain, DETERMINISTIC form (round 6).  d(in)[p] = sum of d(out)[i] over the output pixels i whose chain ends at p; an up-scaling
// chain sends several outputs to one input pixel, and fp32 atomics add them in arrival order (rounds 1-5: the one place outside the weight
// gradients where two runs of a step could differ in the last bit).  Here one work-group owns one (sample, channel) plane: it computes the
// plane's index map into LDS, RANKS the outputs that share a target by ascending output index (round r: every unranked output proposes itself
// with an LDS atomicMin on its target's slot - an integer minimum is order-independent - and the winner takes rank r), and then adds
// rank 0, rank 1, ... into an LDS accumulator with one barrier per rank: every input pixel receives its contributions in ascending output
// index, whatever the scheduling.  The plane is stored with plain coalesced stores (no clear of dst beforehand).
// LDS: (int target + int slot + float acc) per pixel + 1 byte rank = 13 bytes per pixel (53 KB for 64x64, 120 KB for 96x96).
__global__ __launch_bounds__(TPB) void warp_chain_bwd_det_k(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ theta,
                                                            int C, int H, int W, int nstage) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = H * W;
    int* tgt = (int*)smem;
    int* slot = tgt + HW;
    float* acc = (float*)(slot + HW);
    unsigned char* rank = (unsigned char*)(acc + HW);
    const int n = blockIdx.x / C, c = blockIdx.x % C;
    for (int i = threadIdx.x; i < HW; i += TPB) {
        int py = i / W, px = i % W;
        bool ok = true;
        for (int s = nstage - 1; s >= 0 && ok; --s) ok = step(theta + ((size_t)n * nstage + s) * 6, W, H, px, py);
        tgt[i] = ok ? py * W + px : -1;
        rank[i] = ok ? 255 : 254;          // 255: not ranked yet; 254: contributes nothing
        acc[i] = 0.f;
    }
    __syncthreads();
    int nrounds = 0;
    for (int r = 0; r < 254; ++r) {
        for (int i = threadIdx.x; i < HW; i += TPB) slot[i] = 0x7fffffff;
        __syncthreads();
        int pending = 0;
        for (int i = threadIdx.x; i < HW; i += TPB)
            if (rank[i] == 255) { atomicMin(&slot[tgt[i]], i); pending = 1; }
        if (!__syncthreads_or(pending)) break;
        for (int i = threadIdx.x; i < HW; i += TPB)
            if (rank[i] == 255 && slot[tgt[i]] == i) rank[i] = (unsigned char)r;
        nrounds = r + 1;
        __syncthreads();
    }
    const float* sp = src + ((size_t)n * C + c) * HW;
    for (int r = 0; r < nrounds; ++r) {
        for (int i = threadIdx.x; i < HW; i += TPB)
            if (rank[i] == r) acc[tgt[i]] += sp[i];        // (one writer per target and round)
        __syncthreads();
    }
    float* dp = dst + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += TPB) dp[i] = acc[i];
}
// Occlusion paste (train_human.py:409): img[:, r0:r1, c0:c1] = img[:, rs:rs+(r1-r0), cs:cs+(c1-c0)] for each listed image,
// reading the whole source patch before writing (one block per image; patches are at most 20x20x3 = 1200 values).
__global__ void patch_paste_k(float* __restrict__ img, const int* __restrict__ boxes, int C, int H, int W) {
    __shared__ float buf[4096];
    const int* b = boxes + blockIdx.x * 6;          // r0, r1, c0, c1, rs, cs
    const int r0 = b[0], r1 = b[1], c0 = b[2], c1 = b[3], rs = b[4], cs = b[5];
    const int ph = r1 - r0, pw = c1 - c0, n = C * ph * pw;
    float* base = img + (size_t)blockIdx.x * C * H * W;
    for (int i = threadIdx.x; i < n; i += TPB) {
        const int c = i / (ph * pw), r = (i / pw) % ph, q = i % pw;
        buf[i] = base[((size_t)c * H + rs + r) * W + cs + q];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += TPB) {
        const int c = i / (ph * pw), r = (i / pw) % ph, q = i % pw;
        base[((size_t)c * H + r0 + r) * W + c0 + q] = buf[i];
    }
}
// The occlusion DECISIONS of train_human.py:374-410 on the device, one thread per sample: which samples (any confidence >=
// thresh and a uniform draw <= rate), which confident key point, the box around it in image pixels (rows from y, columns from
// x: the reference's index math) and where the replacement patch comes from.  u[n][0..3] are uniform [0,1) draws supplied by
// the caller: u0 = the rate test, u1 -> np.random.choice among the confident key points, u2 / u3 -> np.random.randint of the
// patch origin.  (The reference consumes its draws only for qualifying samples, from an unseeded global generator; drawing
// all four for every sample gives the same distribution without the read-back that made this step un-capturable.)
// boxes[n] = {r0, r1, c0, c1, rs, cs} (a zero-area box when the sample is not selected), apply[n] = 0 / 1.
__global__ void occlusion_pick_k(const float* __restrict__ conf, const int* __restrict__ idx, const float* __restrict__ u, int N, int K, int w,
                                 double ratio, int image_size, float rate, float thresh, int occ, int* __restrict__ boxes,
                                 unsigned char* __restrict__ apply) {
    const int n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    int count = 0;
    for (int k = 0; k < K; ++k) count += conf[n * K + k] >= thresh;
    int* b = boxes + n * 6;
    if (count == 0 || !(u[n * 4] <= rate)) {
        for (int i = 0; i < 6; ++i) b[i] = 0;
        apply[n] = 0;
        return;
    }
    int j = (int)(u[n * 4 + 1] * (float)count);
    if (j > count - 1) j = count - 1;
    int c = 0;
    for (int k = 0; k < K; ++k)
        if (conf[n * K + k] >= thresh) { if (j == 0) { c = k; break; } --j; }
    const int flat = idx[n * K + c];
    const int px = (int)((double)(flat % w) * ratio), py = (int)((double)(flat / w) * ratio);      // (pred_position * ratio).astype(int)
    const int r0 = max(py - occ, 0), r1 = min(py + occ, image_size);
    const int c0 = max(px - occ, 0), c1 = min(px + occ, image_size);
    const int mr = image_size - (r1 - r0) + 1, mc = image_size - (c1 - c0) + 1;
    int rs = (int)(u[n * 4 + 2] * (float)mr), cs = (int)(u[n * 4 + 3] * (float)mc);
    if (rs > mr - 1) rs = mr - 1;
    if (cs > mc - 1) cs = mc - 1;
    b[0] = r0; b[1] = r1; b[2] = c0; b[3] = c1; b[4] = rs; b[5] = cs;
    apply[n] = 1;
}

// dst[n] = flag[n] ? a[n] : b[n] over rows of `row` floats (16-byte vectors): only the selected samples take the occluded image
__global__ void select_rows_k(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b, const unsigned char* __restrict__ flag,
                              size_t row4, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total4; i += (size_t)gridDim.x * TPB) {
        const size_t n = i / row4;
        ((f32x4*)dst)[i] = flag[n] ? ((const f32x4*)a)[i] : ((const f32x4*)b)[i];
    }
}
}  // namespace

int occlusion_pick(hipStream_t s, const float* conf, const int* idx, const float* u, int N, int K, int w, double ratio, int image_size, float rate,
                   float thresh, int occ, int* boxes, unsigned char* apply) {
    if (N <= 0 || K <= 0 || w <= 0 || occ < 0) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(occlusion_pick_k, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, s, conf, idx, u, N, K, w, ratio, image_size, rate, thresh, occ, boxes,
                       apply);
    return udapose_check_launch();
}
int select_rows(hipStream_t s, float* dst, const float* a, const float* b, const unsigned char* flag, int N, size_t row) {
    if (N <= 0 || row % 4) return UDAPOSE_ERR_ARG;
    const size_t total4 = (size_t)N * row / 4;
    size_t blocks = (total4 + TPB - 1) / TPB;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(select_rows_k, dim3((int)blocks), dim3(TPB), 0, s, dst, a, b, flag, row / 4, total4);
    return udapose_check_launch();
}

int patch_paste(hipStream_t s, float* img, const int* boxes, int n, int C, int H, int W, int max_patch_elems) {
    if (n <= 0) return UDAPOSE_OK;
    if (max_patch_elems > 4096) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(patch_paste_k, dim3(n), dim3(TPB), 0, s, img, boxes, C, H, W);
    return udapose_check_launch();
}

// The loop's inverse affine matrices on the device, in double precision (torchvision's _get_inverse_affine_matrix with centre (0,0),
// as warp.inverse_affine_matrix states it): params[n] = (angle, tx, ty, shear_x, shear_y, scale) of the collated aug_param
// (lib/transforms/keypoint_detection.py:139), angles in degrees.
//   fwd[n][3][6]: translate by (tx, ty) / ratio | rotate by angle and scale | shear      (train_human.py:366-368, 421-423)
//   back[n][1][6]: the occlusion path's single warp back (-angle, (-tx, -ty) / ratio, 1 / scale, -shear)   (train_human.py:412)
__device__ __forceinline__ void inv_affine(double angle, double tx, double ty, double scale, double shx, double shy, float* m) {
    const double d2r = 0.017453292519943295;
    const double rot = angle * d2r, sx = shx * d2r, sy = shy * d2r;
    const double a = cos(rot - sy) / cos(sy);
    const double b = -cos(rot - sy) * tan(sx) / cos(sy) - sin(rot);
    const double c = sin(rot - sy) / cos(sy);
    const double d = -sin(rot - sy) * tan(sx) / cos(sy) + cos(rot);
    const double m0 = d / scale, m1 = -b / scale, m3 = -c / scale, m4 = a / scale;
    m[0] = (float)m0; m[1] = (float)m1; m[2] = (float)(m0 * (-tx) + m1 * (-ty));
    m[3] = (float)m3; m[4] = (float)m4; m[5] = (float)(m3 * (-tx) + m4 * (-ty));
}
__global__ void recon_thetas_k(const double* __restrict__ params, int N, double ratio, float* __restrict__ fwd, float* __restrict__ back) {
    const int n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    const double* q = params + (size_t)n * 6;
    const double angle = q[0], tx = q[1], ty = q[2], shx = q[3], shy = q[4], scale = q[5];
    if (fwd) {
        float* f = fwd + (size_t)n * 18;
        inv_affine(0.0, tx / ratio, ty / ratio, 1.0, 0.0, 0.0, f);
        inv_affine(angle, 0.0, 0.0, scale, 0.0, 0.0, f + 6);
        inv_affine(0.0, 0.0, 0.0, 1.0, shx, shy, f + 12);
    }
    if (back) inv_affine(-angle, -tx / ratio, -ty / ratio, 1.0 / scale, -shx, -shy, back + (size_t)n * 6);
}
int affine_recon_thetas(hipStream_t s, const double* params, int N, double ratio, float* fwd, float* back) {
    if (N <= 0 || ratio == 0.0) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(recon_thetas_k, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, s, params, N, ratio, fwd, back);
    return udapose_check_launch();
}

int affine_warp_chain(hipStream_t s, const float* src, float* dst, const float* theta, int N, int C, int H, int W, int nstage, int backward) {
    if (nstage < 1 || nstage > 8) return UDAPOSE_ERR_ARG;
    const size_t total = (size_t)N * H * W;
    int blocks = (int)((total + TPB - 1) / TPB);
    if (blocks > 4096) blocks = 4096;
    const int cgroups = C >= 16 ? 16 : (C >= 4 ? 4 : 1);
    if (backward) {
        // deterministic form (one work-group per (sample, channel) plane, ranks in LDS) wherever the plane fits; more than 253 outputs on one
        // input pixel cannot happen with the loop's scales (<= 1 / 0.6 per axis), and a plane beyond the LDS budget takes the atomic form
        const size_t lds = (size_t)H * W * 13 + 16;
        if (lds <= 150 * 1024 && (long long)N * C < (1ll << 31)) {
            static std::atomic<unsigned long long> attr_done{0};
            static std::mutex attr_mu;
            once_per_device(attr_done, attr_mu, [] {
                (void)hipFuncSetAttribute((const void*)warp_chain_bwd_det_k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            });
            hipLaunchKernelGGL(warp_chain_bwd_det_k, dim3(N * C), dim3(TPB), (unsigned)((lds + 15) & ~(size_t)15), s, src, dst, theta, C, H, W, nstage);
            return udapose_check_launch();
        }
        if (pw_zero(s, dst, (size_t)N * C * H * W * sizeof(float)) != UDAPOSE_OK) return UDAPOSE_ERR_LAUNCH;
        hipLaunchKernelGGL(warp_chain_k<true>, dim3(blocks, cgroups), dim3(TPB), 0, s, src, dst, theta, N, C, H, W, nstage);
    } else {
        hipLaunchKernelGGL(warp_chain_k<false>, dim3(blocks, cgroups), dim3(TPB), 0, s, src, dst, theta, N, C, H, W, nstage);
    }
    return udapose_check_launch();
}

// mean over k re-warped teacher views (train_human.py:361-372 with --k > 1: `torch.mean(recons, dim=0)`): the k values of an element
// are added in view order in fp32 and divided by k, as ATen's mean over the leading dimension does for a handful of rows
struct ViewPtrs { const float* p[8]; };
__global__ void mean_views_k(ViewPtrs v, int k, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float a = v.p[0][i];
        for (int j = 1; j < k; ++j) a += v.p[j][i];
        dst[i] = a / (float)k;
    }
}
int affine_mean_views(hipStream_t s, const float* const* srcs, int k, float* dst, size_t n) {
    if (k < 1 || k > 8 || !srcs || !dst) return UDAPOSE_ERR_ARG;
    ViewPtrs v;
    for (int j = 0; j < 8; ++j) v.p[j] = j < k ? srcs[j] : nullptr;
    for (int j = 0; j < k; ++j) if (!v.p[j]) return UDAPOSE_ERR_ARG;
    size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(mean_views_k, dim3((unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g))), dim3(256), 0, s, v, k, dst, n);
    return udapose_check_launch();
}
