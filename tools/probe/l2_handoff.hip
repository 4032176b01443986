// Dev probe (round 4): does a consumer kernel find, in ITS XCD's L2, the bytes a producer kernel's work-groups on the same XCD
// wrote just before (two launches on one stream)?  Producer: work-group b (256 threads) writes slice b of a buffer with plain 16-byte
// stores.  Consumer: work-group b streams slice (b + shift) of that buffer into LDS (global_load_lds_dwordx4, 3 stages of 16 KiB in
// flight).  shift = 0: same work-group index = same XCD under the hardware's round-robin dealing (b mod 8); shift = 1, 3: a
// neighbour's slice, written on another XCD.  Duration = max(end) - min(start) of in-kernel s_memrealtime stamps (100 MHz).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/l2_handoff tools/probe/l2_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void produce_k(char* dst, size_t slice, unsigned val) {
    char* p = dst + (size_t)blockIdx.x * slice;
    const uint4 v = {val, val + 1u, val + 2u, val + 3u};
    for (size_t o = (size_t)threadIdx.x * 16; o < slice; o += 256 * 16) *(uint4*)(p + o) = v;
}
__global__ __launch_bounds__(256) void consume_k(const char* src, size_t slice, int shift, unsigned long long* stamps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const char* base = src + (size_t)((blockIdx.x + shift) % gridDim.x) * slice;
    const int nst = (int)(slice / 16384);
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    auto issue = [&](int st, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = i * 4 + wid;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)st * 16384 + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + buf * 16384 + piece * 1024), 16, 0, 0);
        }
    };
    issue(0, 0);
    if (nst > 1) issue(1, 1);
    for (int st = 0; st < nst; ++st) {
        if (st + 2 <= nst - 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 2 < nst) issue(st + 2, (st + 2) % 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; sink[blockIdx.x] = ((unsigned*)smem)[0]; }
}
int main() {
    const int wgs = 512;                               // two per CU
    char* d; unsigned long long* st; unsigned* sink;
    hipMalloc((void**)&d, (size_t)256 << 20); hipMalloc((void**)&st, wgs * 16); hipMalloc((void**)&sink, wgs * 4);
    hipFuncSetAttribute((const void*)consume_k, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    printf("# producer launch -> consumer launch, 512 work-groups; total MiB | consumer reads slice b+shift | us | TB/s (LDS fill)\n");
    for (size_t total_mib : {8, 16, 24, 32, 64, 128}) {
        const size_t slice = ((size_t)total_mib << 20) / wgs;
        for (int shift : {0, 8, 1, 3}) {
            double best = 1e30;
            for (int rep = 0; rep < 6; ++rep) {
                hipLaunchKernelGGL(produce_k, dim3(wgs), dim3(256), 0, 0, d, slice, (unsigned)rep);
                hipLaunchKernelGGL(consume_k, dim3(wgs), dim3(256), 49152, 0, d, slice, shift, st, sink);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(2 * wgs);
                hipMemcpy(h.data(), st, wgs * 16, hipMemcpyDeviceToHost);
                unsigned long long a = ~0ull, b = 0;
                for (int i = 0; i < wgs; ++i) { a = std::min(a, h[2 * i]); b = std::max(b, h[2 * i + 1]); }
                if (rep >= 2) best = std::min(best, (double)(b - a) * 10e-9);
            }
            printf("%4zu MiB | shift %d%s | %7.1f us | %6.2f TB/s\n", total_mib, shift, shift % 8 == 0 ? " (same XCD)" : " (another XCD)", best * 1e6,
                   (double)(total_mib << 20) / best / 1e12);
        }
    }
    return 0;
}
