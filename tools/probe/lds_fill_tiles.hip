// Dev probe (round 5): does the SHAPE of what a work-group streams from HBM matter to the LDS-DMA fill rate?  The weight-gradient launch reads
// [pixels x C] bf16 tensors as [pixel range x 128-channel] tiles - 256 contiguous bytes per pixel row, the row's other tiles read by neighbouring
// work-groups of the same XCD at about the same time - and runs at 3.4 TB/s, against 5.6-6.0 TB/s for work-groups that stream private CONTIGUOUS
// slices (lds_fill.hip).  Here: a 2 GiB [P x 1024] bf16 matrix (2 KiB rows), work-groups as (pixel range, channel tile of TW bytes) units, four per
// CU, 16 KiB stages, ring of 2 (the weight gradients' residency); the C/TW tiles of one pixel range are blocks b, b + 8, b + 16 .. (one XCD).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_fill_tiles tools/probe/lds_fill_tiles.hip        run: tools/probe/lds_fill_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int TW>   // bytes of one pixel row a work-group reads (its channel tile): 128 .. 2048
__global__ __launch_bounds__(256) void tile_fill_k(const char* __restrict__ src, int row_bytes, int rows_per_unit, int tiles, int lockstep, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // unit = (pixel range, tile): blocks of one XCD (b % 8 equal) walk the tiles of a range first
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = local % tiles, range = (local / tiles) * 8 + xcd;
    const char* base = src + (size_t)range * rows_per_unit * row_bytes + (size_t)tile * TW;
    constexpr int RPI = 1024 / TW > 0 ? 1024 / TW : 1;      // rows per wave instruction (1 KiB)
    constexpr int LPR = TW / 16;                              // lanes per row
    constexpr int ROWS_PER_STAGE = 16384 / TW;
    const int nst = rows_per_unit / ROWS_PER_STAGE;
    auto issue = [&](int st, int buf) {
        const char* s = base + (size_t)st * ROWS_PER_STAGE * row_bytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = i * 4 + wid;                    // 16 pieces of 1 KiB per stage
            size_t off;
            if constexpr (TW >= 1024) off = (size_t)(piece / (TW / 1024)) * row_bytes + (size_t)(piece % (TW / 1024)) * 1024 + lane * 16;
            else off = (size_t)(piece * RPI + lane / LPR) * row_bytes + (size_t)(lane % LPR) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + off),
                                             (__attribute__((address_space(3))) void*)(smem + buf * 16384 + piece * 1024), 16, 0, 0);
        }
    };
    issue(0, 0);
    for (int it = 0; it + 1 < nst; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(it + 1, (it + 1) & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && out) out[blockIdx.x] = ((unsigned long long*)smem)[0] + (unsigned long long)lockstep;
}

template <int TW>
double run(const char* d, size_t total_bytes, int row_bytes, int rows_per_unit, unsigned long long* o) {
    const int tiles = row_bytes / TW;
    const size_t rows = total_bytes / row_bytes;
    const int ranges = (int)(rows / rows_per_unit) / 8 * 8;
    const int wgs = ranges * tiles;
    const int lds = 36 * 1024;                                 // four work-groups per CU
    hipFuncSetAttribute((const void*)tile_fill_k<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(tile_fill_k<TW>, dim3(wgs), dim3(256), lds, 0, d, row_bytes, rows_per_unit, tiles, 0, o);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(tile_fill_k<TW>, dim3(wgs), dim3(256), lds, 0, d, row_bytes, rows_per_unit, tiles, 0, o);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)ranges * rows_per_unit * row_bytes / (ms * 1e-3) / 1e12;      // TB/s
}

int main() {
    const size_t big = (size_t)2 << 30;
    char* d; unsigned long long* o;
    if (hipMalloc((void**)&d, big) != hipSuccess || hipMalloc((void**)&o, (size_t)1 << 22) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, big);
    hipDeviceSynchronize();
    printf("# [P x C] bf16 matrix streamed once from HBM through LDS-DMA as (pixel range, channel tile) units, 4 work-groups per CU, ring of two 16 KiB stages\n");
    for (int row_bytes : {2048, 512}) {
        for (int rpu : {4096, 1024}) {
            printf("row %4d B (C = %4d), %4d pixel rows per unit | tile 128 B: %5.2f TB/s | 256 B: %5.2f | 512 B: %5.2f", row_bytes, row_bytes / 2, rpu,
                   run<128>(d, big, row_bytes, rpu, o), run<256>(d, big, row_bytes, rpu, o), run<512>(d, big, row_bytes, rpu, o));
            if (row_bytes >= 2048) printf(" | 1024 B: %5.2f | 2048 B (whole rows): %5.2f", run<1024>(d, big, row_bytes, rpu, o), run<2048>(d, big, row_bytes, rpu, o));
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
