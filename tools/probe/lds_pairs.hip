// Dev probe: which lane pairs of a wave conflict on ds_read_b128?  Base pattern: lane l reads bank group (l % 16) of its own 256-byte row (conflict-free:
// tools/probe/lds_conflicts).  Variant (a, b): lane b is moved to lane a's group (its own row).  Prints the penalty matrix for lanes 0..31 and a few 32+ pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const int* __restrict__ addr, int npat, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 64) ((float*)smem)[i] = (float)i;
    __syncthreads();
    for (int p = 0; p < npat; ++p) {
        const int a = addr[p * 64 + threadIdx.x];
        u32x4 acc = {0, 0, 0, 0};
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 16; ++it) {
            u32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const volatile u32x4*)(smem + a);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[p] = t1 - t0;
        if (acc[0] == 0x12345678u) out[npat] = acc[1];
    }
}
int main() {
    std::vector<int> h;
    std::vector<std::pair<int, int>> pr;
    auto base = [](int l) { return (l % 16) * 16 + l * 256; };
    auto push = [&](int a, int b) {
        for (int l = 0; l < 64; ++l) h.push_back(l == b && a >= 0 ? (a % 16) * 16 + b * 256 : base(l));
        pr.push_back({a, b});
    };
    push(-1, -1);
    for (int a = 0; a < 32; ++a) for (int b = a + 1; b < 32; ++b) if ((a % 16) != (b % 16)) push(a, b);
    for (int b = 32; b < 64; b += 3) push(1, b);
    const int np = (int)pr.size();
    int* d; unsigned long long* o;
    (void)hipMalloc((void**)&d, h.size() * 4); (void)hipMalloc((void**)&o, (np + 1) * 8);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, d, np, o);
    std::vector<unsigned long long> r(np + 1);
    (void)hipMemcpy(r.data(), o, (np + 1) * 8, hipMemcpyDeviceToHost);
    const double b0 = r[0] / 256.0;
    printf("base %.2f ticks per read\n", b0);
    std::vector<std::vector<double>> m(32, std::vector<double>(64, -1));
    for (int p = 1; p < np; ++p) m[pr[p].first][pr[p].second] = r[p] / 256.0 - b0;
    printf("penalty of moving lane b (column) to the bank group of lane a (row); '.' = same group anyway\n     ");
    for (int b = 0; b < 32; ++b) printf("%2d ", b);
    printf("\n");
    for (int a = 0; a < 32; ++a) {
        printf("a=%2d ", a);
        for (int b = 0; b < 32; ++b) { if (m[a][b] < -0.5) printf(" . "); else printf("%2.0f ", m[a][b]); }
        printf("\n");
    }
    printf("a=1 vs b>=32:");
    for (int b = 32; b < 64; b += 3) printf(" b%d:%.0f", b, m[1][b]);
    printf("\n");
    return 0;
}
