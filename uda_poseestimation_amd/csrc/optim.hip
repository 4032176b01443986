// Multi-tensor optimizer kernels: EMA teacher update, Adam, SGD(nesterov).  One launch sweeps every parameter tensor
// through a block -> (tensor, offset) table; 16-byte accesses; HBM-roofline bound (EMA 12 B/param, Adam 28 B/param).
#include "common.h"

// no FMA contraction in this file: the EMA must round p*alpha and src*(1-alpha) separately to be bit-identical with the
// reference's two-step update (hipcc contracts a*b+c by default, and __fmul_rn/__fadd_rn do not prevent it)
#pragma clang fp contract(off)
// (belt and braces: the products are additionally pinned in registers by an empty asm, which no contraction can cross)
__device__ __forceinline__ float ema1(float p, float s, float a, float b) {
    float t1 = p * a, t2 = s * b;
    asm volatile("" : "+v"(t1), "+v"(t2));
    return t1 + t2;
}

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
            for (int e = 0; e < 4; ++e) p[e] = ema1(p[e], s[e], alpha, one_minus_alpha);
            *(f32x4*)(tp + i) = p;
        }
        const long long tail = off + ((end - off) & ~3LL);
        for (long long i = tail + threadIdx.x; i < end; i += TPB)
            tp[i] = ema1(tp[i], sp[i], alpha, one_minus_alpha);
    } else {
        for (long long i = off + threadIdx.x; i < end; i += TPB)
            tp[i] = ema1(tp[i], sp[i], alpha, one_minus_alpha);
    }
}

// device-resident optimizer state (graph-replay safe): state = [step, 1-beta1^step, sqrt(1-beta2^step), lr, grad_scale, -, -, -].
// The step counter ticks on the device; lr and grad_scale are READ from it by the sweep, so that an lr scheduler
// (MultiStepLR in the reference, train_human.py:143,202) or a loss-scale change reaches a captured step through a
// 8-byte host-to-device copy instead of being frozen into the graph as kernel arguments.
// state[5] = found_inf (set by grad_check_k: a non-finite gradient under loss scaling): the step is skipped whole - no
// tick, no update - exactly as GradScaler.step() skips optimizer.step() (train_human.py:286,437); state[6] = loss scale S,
// state[7] = growth tracker, state[4] = grad_scale = 1/S (scaler_update_k).
__global__ void adam_tick_k(float* __restrict__ state, float beta1, float beta2) {
    if (state[5] != 0.f) return;
    const double t = (double)state[0] + 1.0;
    state[0] = (float)t;
    state[1] = (float)(1.0 - pow((double)beta1, t));
    state[2] = (float)sqrt(1.0 - pow((double)beta2, t));
}

// torch.optim.Adam (no amsgrad, no weight decay unless wd != 0): operands a=param b=grad c=exp_avg d=exp_avg_sq
__global__ void adam_k(MtTable t, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                       const float* __restrict__ dev_state) {
    if (dev_state) {
        if (dev_state[5] != 0.f) return;            // inf / nan gradients: skip the step (GradScaler semantics)
        bc1 = dev_state[1]; bc2_sqrt = dev_state[2]; lr = dev_state[3]; gscale = dev_state[4];
    }
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
__global__ void sgd_tick_k(float* __restrict__ state) { if (state[5] == 0.f) state[0] += 1.f; }
__global__ void sgd_k(MtTable t, float lr, float momentum, float wd, int nesterov, int first_step, float gscale,
                      const float* __restrict__ dev_state) {
    // (first_step: torch initialises the momentum buffer with the first gradient)
    if (dev_state) {
        if (dev_state[5] != 0.f) return;
        first_step = dev_state[0] == 1.f; lr = dev_state[3]; gscale = dev_state[4];
    }
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
// ---- dynamic loss scaling (torch.cuda.amp.GradScaler, train_human.py:260,285-287,324,436-440) on the device
// found_inf: operand b = the gradient tensors; any non-finite value raises state[5] (every writer stores the same 1.0f)
// g2 (round 4): byte distance to a second per-pass gradient buffer whose sum with the first is still pending (the fused tail adds the two
// itself): the check then looks at g + g2, the value the optimizer sweep will form - no separate 0.64 GB axpy in front of the check
__global__ void grad_check_k(MtTable t, float* __restrict__ state, long long g2) {
    const int ti = t.blk_tensor[blockIdx.x];
    const long long off = t.blk_off[blockIdx.x];
    const float* g = (const float*)t.b[ti];
    const long long n = t.sizes[ti];
    const long long end = off + CHUNK < n ? off + CHUNK : n;
    bool bad = false;
    long long i0 = off;
    // 16-byte loads where the chunk allows it (a pure streaming read of 4 - 8 bytes per parameter, 0.44 GB for PoseResNet-101's two buffers: with
    // 4-byte loads it ran at about half the rate); the scalar loop takes what is left (tensor tails, unaligned small tensors)
    if (((uintptr_t)(g + off) & 15) == 0 && (g2 & 15) == 0) {
        const long long n4 = (end - off) >> 2;
        for (long long q = threadIdx.x; q < n4; q += TPB) {
            f32x4 v = *(const f32x4*)(g + off + q * 4);
            if (g2) {
                const f32x4 w = *(const f32x4*)((const char*)(g + off + q * 4) + g2);
                v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
            }
            bad = bad || !(fabsf(v[0]) <= 3.4028234e38f) || !(fabsf(v[1]) <= 3.4028234e38f) || !(fabsf(v[2]) <= 3.4028234e38f) || !(fabsf(v[3]) <= 3.4028234e38f);
        }
        i0 = off + n4 * 4;
    }
    for (long long i = i0 + threadIdx.x; i < end; i += TPB) {
        float v = g[i];
        if (g2) v += *(const float*)((const char*)(g + i) + g2);
        bad = bad || !(fabsf(v) <= 3.4028234e38f);        // inf or nan
    }
    if (bad) state[5] = 1.f;
}
// GradScaler.update(): found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and scale *= growth every `interval`
// clean steps.  Re-arms found_inf and refreshes grad_scale = 1 / scale for the next step's sweep.
__global__ void scaler_update_k(float* __restrict__ state, float growth, float backoff, float interval) {
    float scale = state[6], tr = state[7];
    if (state[5] != 0.f) { scale *= backoff; tr = 0.f; }
    else { tr += 1.f; if (tr >= interval) { scale *= growth; tr = 0.f; } }
    state[6] = scale; state[7] = tr; state[5] = 0.f;
    state[4] = 1.f / scale;
}
// ---- fused optimizer tail -------------------------------------------------------------------------------------------------
// After the backward a step has nothing left but sweeps over the 55 M parameters, one after the other on one stream: Adam
// (28 B / parameter), the EMA teacher update (12 B), and - at the start of the next step - the element-type weight packs of
// both networks, which re-read the fp32 masters (3 x 6 B).  One kernel does all of it while every value is in registers:
// read p, g, m, v and the teacher; write p, m, v, the teacher and the packs: 46 B / parameter instead of 58.
// The arithmetic is adam_k's and ema_k's, expression for expression (this file is compiled without FMA contraction), so the
// parameters, moments and teacher come out bit-identical to the separate launches and the packs are casts of those values.
// Linear jobs (BatchNorm vectors, biases, stem / head / fc weights): 4096-element chunks, optional same-layout packs.
// Tiled jobs (conv / deconv weights [A][T][B], A and B multiples of 64): one 64 x 64 tile of one tap per block; the
// same-layout pack is written from registers, the transposed pack [B][T][A] goes through an LDS tile.
struct TailJob {
    float* p; const float* g; float* m; float* v; float* t;   // student parameter, gradient, moments, teacher parameter
    elem_t* sd; elem_t* td; elem_t* sx; elem_t* tx;             // packs: student / teacher same-layout, student / teacher transposed
    int A, T, B, adam;                                          // A == 0: linear job; adam == 0: no gradient (EMA + packs only)
    long long n;
};

struct TailHyper { float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, gscale, alpha, oma; int do_adam; long long g2; };   // g2: byte distance to a second gradient buffer (0: none)

__device__ __forceinline__ void tail1(const TailHyper& h, bool adam, float& pv, float gr, float& mi, float& vi, float& tv) {
    if (adam) {
        gr *= h.gscale;
        if (h.wd != 0.f) gr += h.wd * pv;
        mi = mi * h.beta1 + (1.f - h.beta1) * gr;
        vi = vi * h.beta2 + (1.f - h.beta2) * gr * gr;
        pv = pv - (h.lr / h.bc1) * (mi / (sqrtf(vi) / h.bc2_sqrt + h.eps));
    }
    tv = ema1(tv, pv, h.alpha, h.oma);
}

__global__ __launch_bounds__(TPB) void opt_tail_k(const TailJob* __restrict__ jobs, const int* __restrict__ blk_job, const int* __restrict__ blk_sub,
                                                  TailHyper h, const float* __restrict__ dev_state) {
    __shared__ elem_t ts[64][66], tt[64][66];
    if (dev_state) {
        h.bc1 = dev_state[1]; h.bc2_sqrt = dev_state[2]; h.lr = dev_state[3]; h.gscale = dev_state[4];
        if (dev_state[5] != 0.f) h.do_adam = 0;          // inf / nan gradients: the optimizer step is skipped, the EMA is not
    }
    const TailJob j = jobs[blk_job[blockIdx.x]];
    const int sub = blk_sub[blockIdx.x];
    const bool adam = j.adam && h.do_adam;
    if (j.A == 0) {
        const long long off = (long long)sub * CHUNK;
        const long long end = off + CHUNK < j.n ? off + CHUNK : j.n;
        for (long long i = off + threadIdx.x; i < end; i += TPB) {
            float pv = j.p[i], tv = j.t[i], mi = 0.f, vi = 0.f;
            if (adam) { mi = j.m[i]; vi = j.v[i]; }
            float gr = adam ? j.g[i] : 0.f;
            if (adam && h.g2) gr += *(const float*)((const char*)(j.g + i) + h.g2);      // (the two passes' gradients: g1 + g2 as axpy would)
            tail1(h, adam, pv, gr, mi, vi, tv);
            if (adam) { j.p[i] = pv; j.m[i] = mi; j.v[i] = vi; }
            j.t[i] = tv;
            if (j.sd) j.sd[i] = (elem_t)pv;
            if (j.td) j.td[i] = (elem_t)tv;
        }
        return;
    }
    const int tb = j.B >> 6;
    const int t = sub % j.T, tile = sub / j.T;
    const int a0 = (tile / tb) << 6, b0 = (tile % tb) << 6;
    const int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) << 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = r + 16 * i;
        const size_t idx = ((size_t)(a0 + a) * j.T + t) * j.B + b0 + c4;
        f32x4 pv = *(const f32x4*)(j.p + idx), tv = *(const f32x4*)(j.t + idx);
        f32x4 mi = {0.f, 0.f, 0.f, 0.f}, vi = mi, gr = mi;
        if (adam) {
            mi = *(const f32x4*)(j.m + idx); vi = *(const f32x4*)(j.v + idx); gr = *(const f32x4*)(j.g + idx);
            if (h.g2) { const f32x4 g2 = *(const f32x4*)((const char*)(j.g + idx) + h.g2); gr[0] += g2[0]; gr[1] += g2[1]; gr[2] += g2[2]; gr[3] += g2[3]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float p1 = pv[e], m1 = mi[e], v1 = vi[e], t1 = tv[e];
            tail1(h, adam, p1, gr[e], m1, v1, t1);
            pv[e] = p1; mi[e] = m1; vi[e] = v1; tv[e] = t1;
        }
        if (adam) { *(f32x4*)(j.p + idx) = pv; *(f32x4*)(j.m + idx) = mi; *(f32x4*)(j.v + idx) = vi; }
        *(f32x4*)(j.t + idx) = tv;
        const elem4 ps = {(elem_t)pv[0], (elem_t)pv[1], (elem_t)pv[2], (elem_t)pv[3]};
        const elem4 pt = {(elem_t)tv[0], (elem_t)tv[1], (elem_t)tv[2], (elem_t)tv[3]};
        if (j.sd) *(elem4*)(j.sd + idx) = ps;
        if (j.td) *(elem4*)(j.td + idx) = pt;
        if (j.sx) { ts[a][c4] = ps[0]; ts[a][c4 + 1] = ps[1]; ts[a][c4 + 2] = ps[2]; ts[a][c4 + 3] = ps[3]; }
        if (j.tx) { tt[a][c4] = pt[0]; tt[a][c4 + 1] = pt[1]; tt[a][c4 + 2] = pt[2]; tt[a][c4 + 3] = pt[3]; }
    }
    if (!j.sx && !j.tx) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = r + 16 * i;
        const size_t idx = ((size_t)(b0 + b) * j.T + t) * j.A + a0 + c4;       // transposed pack [B][T][A]
        if (j.sx) { const elem4 o = {ts[c4][b], ts[c4 + 1][b], ts[c4 + 2][b], ts[c4 + 3][b]}; *(elem4*)(j.sx + idx) = o; }
        if (j.tx) { const elem4 o = {tt[c4][b], tt[c4 + 1][b], tt[c4 + 2][b], tt[c4 + 3][b]}; *(elem4*)(j.tx + idx) = o; }
    }
}
}  // namespace

int opt_chunk() { return CHUNK; }
size_t opt_tail_job_bytes() { return sizeof(TailJob); }
// host-side filler of one table entry (net.hip builds the table: it knows the pack offsets)
void opt_tail_job_fill(void* dst, float* p, const float* g, float* m, float* v, float* t, void* sd, void* td, void* sx, void* tx, int A, int T, int B,
                       int adam, long long n) {
    TailJob j{p, g, m, v, t, (elem_t*)sd, (elem_t*)td, (elem_t*)sx, (elem_t*)tx, A, T, B, adam, n};
    *(TailJob*)dst = j;
}
// tick: advance the device-side step counter / bias corrections in front of the sweep (once per optimizer step: a step whose sweep is
// issued in two parts ticks with the first)
int opt_tail(hipStream_t s, const void* d_jobs, const int* blk_job, const int* blk_sub, int nblocks, float lr, float beta1, float beta2, float eps,
             float wd, int step, float gscale, float* dev_state, float alpha, float oma, int do_adam, long long grad2_delta, int tick) {
    if (grad2_delta % 16) return UDAPOSE_ERR_ARG;
    double bc1 = 1.0, bc2 = 1.0;
    if (dev_state) { if (do_adam && tick) hipLaunchKernelGGL(adam_tick_k, dim3(1), dim3(1), 0, s, dev_state, beta1, beta2); }
    else { bc1 = 1.0 - pow((double)beta1, (double)step); bc2 = 1.0 - pow((double)beta2, (double)step); }
    if (nblocks <= 0) return UDAPOSE_OK;
    TailHyper h{lr, beta1, beta2, eps, wd, (float)bc1, (float)sqrt(bc2), gscale, alpha, oma, do_adam, grad2_delta};
    hipLaunchKernelGGL(opt_tail_k, dim3(nblocks), dim3(TPB), 0, s, (const TailJob*)d_jobs, blk_job, blk_sub, h, dev_state);
    return udapose_check_launch();
}
int opt_grad_check(hipStream_t s, const long long* g, const long long* sizes, const int* blk_tensor, const long long* blk_off, int nblocks,
                   float* dev_state, long long grad2_delta) {
    if (!dev_state || grad2_delta % 4) return UDAPOSE_ERR_ARG;
    MtTable t{nullptr, g, nullptr, nullptr, sizes, blk_tensor, blk_off};
    if (nblocks <= 0) return UDAPOSE_OK;
    hipLaunchKernelGGL(grad_check_k, dim3(nblocks), dim3(TPB), 0, s, t, dev_state, grad2_delta);
    return udapose_check_launch();
}
int opt_scaler_update(hipStream_t s, float* dev_state, float growth, float backoff, int interval) {
    if (!dev_state || interval < 1) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(scaler_update_k, dim3(1), dim3(1), 0, s, dev_state, growth, backoff, (float)interval);
    return udapose_check_launch();
}

int opt_ema(hipStream_t s, const long long* tgt, const long long* src, const long long* sizes, const int* blk_tensor, const long long* blk_off,
            int nblocks, float alpha, float one_minus_alpha) {
    MtTable t{tgt, src, nullptr, nullptr, sizes, blk_tensor, blk_off};
    if (nblocks <= 0) return UDAPOSE_OK;
    hipLaunchKernelGGL(ema_k, dim3(nblocks), dim3(TPB), 0, s, t, alpha, one_minus_alpha);
    return udapose_check_launch();
}
int opt_adam(hipStream_t s, const long long* p, const long long* g, const long long* m, const long long* v, const long long* sizes,
             const int* blk_tensor, const long long* blk_off, int nblocks, float lr, float beta1, float beta2, float eps, float wd, int step,
             float gscale, float* dev_state) {
    MtTable t{p, g, m, v, sizes, blk_tensor, blk_off};
    if (nblocks <= 0) return UDAPOSE_OK;
    // dev_state != NULL: the step counter and bias corrections live on the device (incremented here, so that a captured
    // graph replays correctly); otherwise `step` is the host's 1-based step
    double bc1 = 1.0, bc2 = 1.0;
    if (dev_state) hipLaunchKernelGGL(adam_tick_k, dim3(1), dim3(1), 0, s, dev_state, beta1, beta2);
    else { bc1 = 1.0 - pow((double)beta1, (double)step); bc2 = 1.0 - pow((double)beta2, (double)step); }
    hipLaunchKernelGGL(adam_k, dim3(nblocks), dim3(TPB), 0, s, t, lr, beta1, beta2, eps, wd, (float)bc1, (float)sqrt(bc2), gscale, dev_state);
    return udapose_check_launch();
}
int opt_sgd(hipStream_t s, const long long* p, const long long* g, const long long* buf, const long long* sizes, const int* blk_tensor,
            const long long* blk_off, int nblocks, float lr, float momentum, float wd, int nesterov, int first_step, float gscale,
            float* dev_state) {
    MtTable t{p, g, buf, nullptr, sizes, blk_tensor, blk_off};
    if (nblocks <= 0) return UDAPOSE_OK;
    if (dev_state) hipLaunchKernelGGL(sgd_tick_k, dim3(1), dim3(1), 0, s, dev_state);
    hipLaunchKernelGGL(sgd_k, dim3(nblocks), dim3(TPB), 0, s, t, lr, momentum, wd, nesterov, first_step, gscale, dev_state);
    return udapose_check_launch();
}

// ---- gradient communication in bf16 (engine.GradSync(comm_dtype='bf16')): the fp32 bucket is rounded ONCE to bf16 for the wire,
// every rank receives its 1/W shard of all W ranks (all-to-all), adds them in fp32, and the averaged shard travels back in bf16
// (all-gather): half the bytes of an fp32 all-reduce on every xGMI link, one bf16 rounding per contribution, fp32 accumulation.
namespace {
__global__ void comm_pack_bf16_k(const float* __restrict__ src, long long n, __bf16* __restrict__ dst, long long npad) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npad; i += (long long)gridDim.x * 256) dst[i] = (__bf16)(i < n ? src[i] : 0.f);
}
__global__ void comm_shard_mean_k(const __bf16* __restrict__ in, int W, long long m, float inv_w, __bf16* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256) {
        float a = 0.f;
        for (int w = 0; w < W; ++w) a += (float)in[(long long)w * m + i];
        out[i] = (__bf16)(a * inv_w);
    }
}
__global__ void comm_unpack_bf16_k(const __bf16* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (float)src[i];
}
inline int comm_grid(long long n) { long long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }
}  // namespace
int comm_pack_bf16(hipStream_t s, const float* src, long long n, void* dst, long long npad) {
    if (n < 0 || npad < n) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(comm_pack_bf16_k, dim3(comm_grid(npad)), dim3(256), 0, s, src, n, (__bf16*)dst, npad);
    return udapose_check_launch();
}
int comm_shard_mean(hipStream_t s, const void* in, int W, long long m, void* out) {
    if (W < 1 || m < 0) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(comm_shard_mean_k, dim3(comm_grid(m)), dim3(256), 0, s, (const __bf16*)in, W, m, 1.f / (float)W, (__bf16*)out);
    return udapose_check_launch();
}
int comm_unpack_bf16(hipStream_t s, const void* src, float* dst, long long n) {
    if (n < 0) return UDAPOSE_ERR_ARG;
    hipLaunchKernelGGL(comm_unpack_bf16_k, dim3(comm_grid(n)), dim3(256), 0, s, (const __bf16*)src, dst, n);
    return udapose_check_launch();
}
