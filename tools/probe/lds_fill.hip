// Dev probe (round 4): what the L2 / Infinity Cache / HBM -> LDS operand fill path gives a CU, as a function of how many work-groups
// share the CU, how many 16 KiB stages each keeps in flight, and where the bytes come from.  This is the resource DESIGN.md section 3
// names as the bound of the implicit-GEMM and weight-gradient kernels: each work-group here does what their K loops do to the memory
// system - 256 threads issue global_load_lds_dwordx4 pieces (1 KiB per wave instruction) into a ring of LDS stages, wait with a
// counted vmcnt for the oldest stage, pass a barrier, and re-issue - with no MFMA and no LDS read in between (a ceiling, not a kernel).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_fill tools/probe/lds_fill.hip
//   run:   tools/probe/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NS>   // stages in flight (ring depth); a stage = 16 KiB = 4 pieces per wave
__global__ __launch_bounds__(256) void fill_k(const char* __restrict__ src, size_t region_bytes, size_t stride_bytes, int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * stride_bytes;
    const size_t nst = region_bytes / 16384;                 // stages in this work-group's region (it cycles through them)
    auto issue = [&](size_t st, int buf) {
        const char* s = base + (st % nst) * 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = i * 4 + wid;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + buf * 16384 + piece * 1024), 16, 0, 0);
        }
    };
    for (int b = 0; b < NS - 1; ++b) issue(b, b);
    for (int it = 0; it < iters; ++it) {
        // the oldest stage has landed when at most (NS - 2) younger stages (4 pieces each) are still in flight
        if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (NS == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue((size_t)it + NS - 1, (it + NS - 1) % NS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && out) out[blockIdx.x] = ((unsigned long long*)smem)[0];     // keep the fills alive
}

template <int NS>
double run(const char* d, size_t footprint, int wgs, int lds_bytes, int iters, unsigned long long* o, bool shared) {
    hipFuncSetAttribute((const void*)fill_k<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const size_t region = shared ? footprint : (footprint / wgs) / 16384 * 16384;
    const size_t stride = shared ? 0 : region;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(fill_k<NS>, dim3(wgs), dim3(256), lds_bytes, 0, d, region, stride, iters, o);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(fill_k<NS>, dim3(wgs), dim3(256), lds_bytes, 0, d, region, stride, iters, o);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)wgs * iters * 16384.0 / (ms * 1e-3) / 1e9;      // GB/s chip-wide
}

int main() {
    const size_t big = (size_t)2 << 30;
    char* d; unsigned long long* o;
    if (hipMalloc((void**)&d, big) != hipSuccess || hipMalloc((void**)&o, 8192 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, big);
    hipDeviceSynchronize();
    printf("# LDS-DMA fill ceiling, 256 CUs: source | work-groups per CU | stages (16 KiB) in flight per work-group | TB/s chip-wide | GB/s per CU\n");
    struct Src { const char* name; size_t bytes; bool shared; } srcs[] = {
        {"one 1 MiB panel read by every work-group (L2 hits: a weight tile)", (size_t)1 << 20, true},
        {"16 MiB cycled, a private slice per work-group (fits the 8 L2s)", (size_t)16 << 20, false},
        {"128 MiB cycled, private slices (Infinity Cache)", (size_t)128 << 20, false},
        {"2 GiB streamed once, private slices (HBM)", big, false}};
    for (const Src& s : srcs)
        for (int per_cu : {1, 2, 3, 4}) {
            const int wgs = 256 * per_cu;
            // LDS request sized so that exactly per_cu work-groups fit a CU (160 KiB): the ring itself needs NS * 16 KiB
            const int lds_occ = (160 * 1024 / per_cu) / 1024 * 1024 - 1024;
            const size_t per_wg = s.shared ? s.bytes : s.bytes / wgs;
            const int iters = s.bytes == big ? (int)(per_wg / 16384) - 8 : 2048;
            double r2 = run<2>(d, s.bytes, wgs, lds_occ, iters, o, s.shared);
            double r3 = lds_occ >= 3 * 16384 ? run<3>(d, s.bytes, wgs, lds_occ, iters, o, s.shared) : 0.0;     // (the ring must fit the request)
            double r4 = lds_occ >= 4 * 16384 ? run<4>(d, s.bytes, wgs, lds_occ, iters, o, s.shared) : 0.0;
            double r6 = lds_occ >= 6 * 16384 ? run<6>(d, s.bytes, wgs, lds_occ, iters, o, s.shared) : 0.0;
            printf("%-66s | %d | 1: %6.2f TB/s %6.1f GB/s/CU | 2: %6.2f %6.1f | 3: %6.2f %6.1f | 5: %6.2f %6.1f\n", s.name, per_cu, r2 / 1e3, r2 / 256,
                   r3 / 1e3, r3 / 256, r4 / 1e3, r4 / 256, r6 / 1e3, r6 / 256);
            fflush(stdout);
        }
    return 0;
}
