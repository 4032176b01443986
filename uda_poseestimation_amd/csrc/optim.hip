// Multi-tensor optimizer kernels: EMA teacher update, Adam, SGD(nesterov).  One launch sweeps every parameter tensor
// through a block -> (tensor, offset) table; 16-byte accesses; HBM-roofline bound (EMA 12 B/param, Adam 28 B/param).
#include "common.h"

namespace {
constexpr int TPB = 256;
constexpr int CHUNK = 4096;   // elements per block

struct MtTable {
    const long long* a;        // device array of tensor base addresses (operand 0)
    const long long* b;
    const long long* c;
    const long long* d;
    const long long* sizes;    // elements per tensor
    const int* blk_tensor;     // block -> tensor index
    const long long* blk_off;  // block -> element offset inside the tensor
};

// teacher = fl(fl(teacher*alpha) + fl(student*(1-alpha)))  -- two roundings, exactly utils.py:21-25 (no FMA contraction)
__global__ void ema_k(MtTable t, float alpha, float one_minus_alpha) {
    const int ti = t.blk_tensor[blockIdx.x];
    const long long off = t.blk_off[blockIdx.x];
    float* tp = (float*)t.a[ti];
    const float* sp = (const float*)t.b[ti];
    const long long n = t.sizes[ti];
    const long long end = off + CHUNK < n ? off + CHUNK : n;
    const bool vec = ((((uintptr_t)tp) | ((uintptr_t)sp)) & 15) == 0;
    if (vec) {
        for (long long i = off + threadIdx.x * 4; i + 3 < end; i += TPB * 4) {
            f32x4 p = *(f32x4*)(tp + i);
            const f32x4 s = *(const f32x4*)(sp + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = __fadd_rn(__fmul_rn(p[e], alpha), __fmul_rn(s[e], one_minus_alpha));
            *(f32x4*)(tp + i) = p;
        }
        const long long tail = off + ((end - off) & ~3LL);
        for (long long i = tail + threadIdx.x; i < end; i += TPB)
            tp[i] = __fadd_rn(__fmul_rn(tp[i], alpha), __fmul_rn(sp[i], one_minus_alpha));
    } else {
        for (long long i = off + threadIdx.x; i < end; i += TPB)
            tp[i] = __fadd_rn(__fmul_rn(tp[i], alpha), __fmul_rn(sp[i], one_minus_alpha));
    }
}

// torch.optim.Adam (no amsgrad, no weight decay unless wd != 0): operands a=param b=grad c=exp_avg d=exp_avg_sq
__global__ void adam_k(MtTable t, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const int ti = t.blk_tensor[blockIdx.x];
    const long long off = t.blk_off[blockIdx.x];
    float* p = (float*)t.a[ti];
    const float* g = (const float*)t.b[ti];
    float* m = (float*)t.c[ti];
    float* v = (float*)t.d[ti];
    const long long n = t.sizes[ti];
    const long long end = off + CHUNK < n ? off + CHUNK : n;
    const float step = lr / bc1;
    for (long long i = off + threadIdx.x; i < end; i += TPB) {
        float gr = g[i] * gscale;
        const float pv = p[i];
        if (wd != 0.f) gr += wd * pv;
        const float mi = m[i] * beta1 + (1.f - beta1) * gr;
        const float vi = v[i] * beta2 + (1.f - beta2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        p[i] = pv - step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

// torch.optim.SGD(momentum, nesterov, weight_decay): a=param b=grad c=momentum buffer
__global__ void sgd_k(MtTable t, float lr, float momentum, float wd, int nesterov, int first_step, float gscale) {
    const int ti = t.blk_tensor[blockIdx.x];
    const long long off = t.blk_off[blockIdx.x];
    float* p = (float*)t.a[ti];
    const float* g = (const float*)t.b[ti];
    float* buf = (float*)t.c[ti];
    const long long n = t.sizes[ti];
    const long long end = off + CHUNK < n ? off + CHUNK : n;
    for (long long i = off + threadIdx.x; i < end; i += TPB) {
        float gr = g[i] * gscale + wd * p[i];
        const float b = first_step ? gr : buf[i] * momentum + gr;
        buf[i] = b;
        gr = nesterov ? gr + momentum * b : b;
        p[i] -= lr * gr;
    }
}
}  // namespace

int opt_chunk() { return CHUNK; }

int opt_ema(hipStream_t s, const long long* tgt, const long long* src, const long long* sizes, const int* blk_tensor, const long long* blk_off,
            int nblocks, float alpha, float one_minus_alpha) {
    MtTable t{tgt, src, nullptr, nullptr, sizes, blk_tensor, blk_off};
    if (nblocks <= 0) return UDAPOSE_OK;
    hipLaunchKernelGGL(ema_k, dim3(nblocks), dim3(TPB), 0, s, t, alpha, one_minus_alpha);
    return udapose_check_launch();
}
int opt_adam(hipStream_t s, const long long* p, const long long* g, const long long* m, const long long* v, const long long* sizes,
             const int* blk_tensor, const long long* blk_off, int nblocks, float lr, float beta1, float beta2, float eps, float wd, int step,
             float gscale) {
    MtTable t{p, g, m, v, sizes, blk_tensor, blk_off};
    if (nblocks <= 0) return UDAPOSE_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_k, dim3(nblocks), dim3(TPB), 0, s, t, lr, beta1, beta2, eps, wd, (float)bc1, (float)sqrt(bc2), gscale);
    return udapose_check_launch();
}
int opt_sgd(hipStream_t s, const long long* p, const long long* g, const long long* buf, const long long* sizes, const int* blk_tensor,
            const long long* blk_off, int nblocks, float lr, float momentum, float wd, int nesterov, int first_step, float gscale) {
    MtTable t{p, g, buf, nullptr, sizes, blk_tensor, blk_off};
    if (nblocks <= 0) return UDAPOSE_OK;
    hipLaunchKernelGGL(sgd_k, dim3(nblocks), dim3(TPB), 0, s, t, lr, momentum, wd, nesterov, first_step, gscale);
    return udapose_check_launch();
}
