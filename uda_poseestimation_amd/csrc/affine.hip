// Batched heat-map / image re-warp: the three sequential nearest-neighbour inverse-affine resamplings the reference
// applies per sample with torchvision.transforms.functional.affine (train_human.py:366-368, 421-423), as ONE gather
// kernel driven by a [N][3][6] matrix tensor.  Because every stage is a nearest resample, the composition is evaluated
// as a chain of index maps p3 -> p2 -> p1 -> p0 (NOT as one composed matrix: that would change results).
// Arithmetic follows torchvision/_gen_affine_grid + ATen grid_sampler(nearest, zeros, align_corners=False) in fp32:
// theta / (0.5*[W,H]); base grid at half-integers; ix = ((x+1)*W-1)/2; nearbyint.
#include "common.h"

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
        if (!BWD) {
            for (int c = 0; c < C; ++c) dst[o + (size_t)c * H * W + rem] = ok ? src[o + (size_t)c * H * W + py * W + px] : 0.f;
        } else if (ok) {
            // src = d(out), dst = d(in) (pre-zeroed): several outputs may read the same input pixel
            for (int c = 0; c < C; ++c) atomicAdd(dst + o + (size_t)c * H * W + py * W + px, src[o + (size_t)c * H * W + rem]);
        }
    }
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
}  // namespace

int patch_paste(hipStream_t s, float* img, const int* boxes, int n, int C, int H, int W, int max_patch_elems) {
    if (n <= 0) return UDAPOSE_OK;
    if (max_patch_elems > 4096) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(patch_paste_k, dim3(n), dim3(TPB), 0, s, img, boxes, C, H, W);
    return udapose_check_launch();
}

int affine_warp_chain(hipStream_t s, const float* src, float* dst, const float* theta, int N, int C, int H, int W, int nstage, int backward) {
    if (nstage < 1 || nstage > 8) return UDAPOSE_ERR_ARG;
    const size_t total = (size_t)N * H * W;
    int blocks = (int)((total + TPB - 1) / TPB);
    if (blocks > 4096) blocks = 4096;
    if (backward) {
        if (hipMemsetAsync(dst, 0, (size_t)N * C * H * W * sizeof(float), s) != hipSuccess) return UDAPOSE_ERR_LAUNCH;
        hipLaunchKernelGGL(warp_chain_k<true>, dim3(blocks), dim3(TPB), 0, s, src, dst, theta, N, C, H, W, nstage);
    } else {
        hipLaunchKernelGGL(warp_chain_k<false>, dim3(blocks), dim3(TPB), 0, s, src, dst, theta, N, C, H, W, nstage);
    }
    return udapose_check_launch();
}
