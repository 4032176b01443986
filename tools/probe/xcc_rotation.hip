// Dev probe (round 4): is "work-group b runs on XCD (b + r) mod 8" with the SAME rotation r for every launch - also when several streams
// launch kernels concurrently?  Every launch records, for each of its work-groups, HW_REG_XCC_ID; the host prints the histogram of
// r = (xcc - blockIdx.x) mod 8 per launch (a single r per launch = perfect round-robin from a rotated start) and how r varies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned char* out, int spin) {
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned char)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);
    // a little work so that launches of different streams overlap
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out[0] = 0;
}
int main() {
    const int L = 48, NB[3] = {256, 512, 1024};
    hipStream_t st[3];
    for (auto& s : st) hipStreamCreate(&s);
    unsigned char* d;
    hipMalloc((void**)&d, (size_t)3 * L * 1024);
    for (int mode = 0; mode < 2; ++mode) {         // 0: one stream, back to back; 1: three streams concurrently, different grid sizes
        for (int l = 0; l < L; ++l)
            for (int q = 0; q < (mode ? 3 : 1); ++q)
                hipLaunchKernelGGL(k, dim3(NB[q]), dim3(256), 0, st[mode ? q : 0], d + ((size_t)q * L + l) * 1024, 20000);
        hipDeviceSynchronize();
        std::vector<unsigned char> h((size_t)3 * L * 1024);
        hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost);
        for (int q = 0; q < (mode ? 3 : 1); ++q) {
            int rot_hist[8] = {0}, clean = 0;
            for (int l = 0; l < L; ++l) {
                int hist[8] = {0};
                for (int b = 0; b < NB[q]; ++b) hist[(h[((size_t)q * L + l) * 1024 + b] - b) & 7]++;
                int best = 0;
                for (int r = 1; r < 8; ++r) if (hist[r] > hist[best]) best = r;
                rot_hist[best]++;
                if (hist[best] == NB[q]) ++clean;
            }
            printf("%s, grid %4d: launches with ONE rotation for all work-groups %d / %d; rotation r histogram over launches:", mode ? "3 streams" : "1 stream ", NB[q], clean, L);
            for (int r = 0; r < 8; ++r) printf(" %d", rot_hist[r]);
            printf("\n");
        }
    }
    return 0;
}
