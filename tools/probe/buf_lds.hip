// Dev probe: does buffer_load_dwordx4 ... lds write ZEROS for out-of-range lanes, and is soffset part of the range check?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* src, int nbytes, int soff, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((float*)smem)[i] = -7.f;      // stale pattern
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    int voff = threadIdx.x * 16;
    if ((threadIdx.x & 3) == 1) voff = 0x7fffffff;           // explicit out-of-range marker
    if ((threadIdx.x & 3) == 2) voff = nbytes - 16 + 16;     // just past the end (before soffset)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
    const int n = 64 * 16 * 2;     // 2 KiB
    std::vector<float> h(n / 4);
    for (int i = 0; i < n / 4; ++i) h[i] = 1.f + i;
    char* d; float* o;
    hipMalloc((void**)&d, n); hipMalloc((void**)&o, 256 * 4);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    for (int soff : {0, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, n, soff, o);
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        printf("soffset %d: lane0 %.0f %.0f | lane1(marker) %.0f %.0f | lane2(past end) %.0f %.0f | lane3 %.0f %.0f | lane 35 (voff 560, +soff in range?) %.0f | lane 63 %.0f\n", soff,
               r[0], r[3], r[4], r[7], r[8], r[11], r[12], r[15], r[35 * 4], r[63 * 4]);
    }
    return 0;
}
