// Dev probe: cycles per ds_read_b128 / ds_write_b128 of one wave for the lane -> address patterns of the MFMA fragment reads (which lanes share
// a bank pass on gfx950?).  One wave, 512 dependent-free reads per pattern, s_memtime around them.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const int* __restrict__ addr, int npat, unsigned long long* out, int wr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 64) ((float*)smem)[i] = (float)i;
    __syncthreads();
    for (int p = 0; p < npat; ++p) {
        const int a = addr[p * 64 + threadIdx.x];
        u32x4 acc = {0, 0, 0, 0};
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 32; ++it) {
            if (wr) {
#pragma unroll
                for (int u = 0; u < 16; ++u) *(u32x4*)(smem + a) = acc;
            } else {
                u32x4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *(const volatile u32x4*)(smem + a);       // 16 reads in flight
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u];
            }
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[p] = t1 - t0;
        if (acc[0] == 0x12345678u) out[npat] = acc[1];
    }
}
int main() {
    struct Pat { const char* name; int (*f)(int); };
    static Pat pats[] = {
        {"linear lane*16", [](int l) { return l * 16; }},
        {"rows l&15 x128, chunk l>>4, no swizzle", [](int l) { return (l & 15) * 128 + (l >> 4) * 16; }},
        {"rows l&15 x128, chunk (l>>4) ^ (row>>1)&7 (igemm)", [](int l) { int r = l & 15; return r * 128 + (((l >> 4) ^ ((r >> 1) & 7)) << 4); }},
        {"permuted rows qq*8+r, (row>>1)&7 (first patch kernel)", [](int l) { int i = l & 15, r = (i >> 2) * 8 + (i & 3); return r * 128 + (((l >> 4) ^ ((r >> 1) & 7)) << 4); }},
        {"permuted rows qq*8+r, bit1 | qq<<1 (fixed)", [](int l) { int i = l & 15, r = (i >> 2) * 8 + (i & 3); int f = ((r >> 1) & 1) | (((r / 8) & 3) << 1); return r * 128 + (((l >> 4) ^ f) << 4); }},
        {"16-byte pixels contiguous (ci8 patch)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1056; }},
        {"rows x128, chunk q ^ (l15>>1) (column swizzle, even start)", [](int l) { int p = l & 15; return p * 128 + (((l >> 4) ^ ((p >> 1) & 7)) << 4); }},
        {"rows x128 start odd pixel, column swizzle", [](int l) { int p = (l & 15) + 1; return p * 128 + (((l >> 4) ^ ((p >> 1) & 7)) << 4); }},
        {"rows x128, lanes' q on different taps (row offsets 0, 34, 68, 35 px)", [](int l) { static const int off[4] = {0, 34, 68, 35}; int p = (l & 15) + off[l >> 4]; return p * 128 + (((l >> 4) ^ ((p >> 1) & 7)) << 4); }},
        {"rows x144 (padded), chunk q", [](int l) { return (l & 15) * 144 + (l >> 4) * 16; }},
        {"all lanes same address", [](int l) { return 0; }},
        {"group l%16, unique rows (free if a pass = 16 consecutive lanes)", [](int l) { return (l % 16) * 16 + l * 256; }},
        {"group l%8 + 8*(l>>5&1) (free if a pass = lanes 0-7 + 32-39)", [](int l) { return ((l % 8) + 8 * ((l >> 5) & 1)) * 16 + l * 256; }},
        {"group l%8 + 8*(l>>4&1) (free if a pass = lanes 0-7 + 16-23)", [](int l) { return ((l % 8) + 8 * ((l >> 4) & 1)) * 16 + l * 256; }},
        {"group (l%4) + 4*(l>>4) (free if a pass = lanes {0-3,16-19,32-35,48-51})", [](int l) { return ((l % 4) + 4 * (l >> 4)) * 16 + l * 256; }},
        {"group l%8, unique rows (free if a pass = 8 consecutive lanes)", [](int l) { return (l % 8) * 16 + l * 256; }},
        {"group l%4, unique rows (free if a pass = 4 consecutive lanes)", [](int l) { return (l % 4) * 16 + l * 256; }},
        {"group 0, unique rows (64-way)", [](int l) { return l * 256; }},
        {"store pattern e>>3 rows, (e&7)^swz", [](int l) { int r = l >> 3; return r * 128 + (((l & 7) ^ ((r >> 1) & 7)) << 4); }},
    };
    const int np = sizeof(pats) / sizeof(pats[0]);
    std::vector<int> h(np * 64);
    for (int p = 0; p < np; ++p) for (int l = 0; l < 64; ++l) h[p * 64 + l] = pats[p].f(l);
    int* d; unsigned long long* o;
    hipMalloc((void**)&d, h.size() * 4); hipMalloc((void**)&o, (np + 1) * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wr = 0; wr < 2; ++wr) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, d, np, o, wr);
        std::vector<unsigned long long> r(np + 1);
        hipMemcpy(r.data(), o, (np + 1) * 8, hipMemcpyDeviceToHost);
        for (int p = 0; p < np; ++p) printf("%s %-70s %6.1f memtime ticks per instruction\n", wr ? "write" : "read ", pats[p].name, r[p] / 512.0);
    }
    return 0;
}
